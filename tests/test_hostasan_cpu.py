"""CPU: the host-only translation units of the product library (text / binary PartitionMat emission with its manual pointer
arithmetic, frame tiling, weight packing) under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5,
"Race detection / sanitizers").  The sanitizer build is CPU only - GPU sanitizers are not available on the pool."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "pmp_vvc_tip2023_amd", "csrc")
ASAN_RT = "/opt/rocm/lib/llvm/lib/clang"


def _asan_runtime():
    for d, _, files in os.walk(ASAN_RT):
        if "libclang_rt.asan-x86_64.so" in files:
            return os.path.join(d, "libclang_rt.asan-x86_64.so")
    return None


def test_host_code_under_asan_ubsan():
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("no shared ASan runtime in this toolchain")
    subprocess.check_call(["make", "-s", "-C", CSRC, "hostasan"])
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hostasan_checks.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "hostasan checks passed" in r.stdout
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
