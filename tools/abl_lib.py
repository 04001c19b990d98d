"""Where the measurement library lives: tools/abl/libpmp_hip_abl.so (`make -C tools/abl`) - the product's host code with the notebook
forms of the convolution kernels, their timing-only builds and the experiments that never shipped.  Not part of the product package,
git-ignored AND gpurun-ignored: a tool that needs it on the GPU box builds it there (ensure(), ~2 min of hipcc)."""
import os
import subprocess

ABL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "abl")
ABL_LIB_PATH = os.path.join(ABL_DIR, "libpmp_hip_abl.so")


def ensure():
    subprocess.check_call(["make", "-s", "-j", str(min(16, os.cpu_count() or 2)), "-C", ABL_DIR])
    return ABL_LIB_PATH
