"""CPU: SURVEY.md 8(f) rows N1 / N2 / N4 against the REAL consumer - the reference's patched VTM-10.0, built on Linux from its
sources where they lie (tools/vtm_build/CMakeLists.txt; nothing is copied into this repository).

  N1  EncoderApp parses a PartitionMat pair written by the product (pmp_write_partition_file) with its own
      EncAppCfg::parsePartitionMatrix (EncAppCfg.cpp:4234-4404), encodes, and DecoderApp decodes the stream with matching MD5s;
      different maps give a different stream (the maps steer the encoder); a missing file ends the encoder as the reference
      does (:4252-4263).
  N2  EncoderAppHook (the same encoder with tools/vtm_build/pmp_hook.cpp in place of the text parser) mmaps the binary side
      channel written by pmp_write_partition_binary and produces the IDENTICAL bitstream.
  N4  without any file the hook goes in-process through libpmp_hip.so; on this GPU-less box that must end with the library's
      "no CPU fallback" error (the GPU leg is recorded in profiles/r02_n4_inprocess_hook.txt).

Skipped where /root/reference or cmake is absent (the GPU box).  The first run builds VTM (about 3 minutes on 8 cores)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT

VTM_SRC = "/root/reference/codec/vtm10.0-source-with-pmp-fast-alg"
ENC_CFG = "/root/reference/codec/demo/cfg/encoder_intra_vtm.cfg"
BUILD = os.path.join(ROOT, "tools", "vtm_build", "_build")
W, H, F, QP = 128, 128, 2, 32
SEQ = "Tiny_%dx%d_30" % (W, H)


@pytest.fixture(scope="module")
def vtm():
    if not os.path.isdir(VTM_SRC) or not os.path.isfile(ENC_CFG):
        pytest.skip("reference VTM sources not present")
    if shutil.which("cmake") is None or shutil.which("ninja") is None:
        pytest.skip("cmake / ninja not available")
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "pmp_vvc_tip2023_amd", "csrc")])   # the hook links it
    bins = {n: os.path.join(BUILD, n) for n in ("EncoderApp", "DecoderApp", "EncoderAppHook")}
    if not all(os.path.isfile(p) for p in bins.values()):
        subprocess.check_call(["cmake", "-S", os.path.join(ROOT, "tools", "vtm_build"), "-B", BUILD, "-G", "Ninja", "-DCMAKE_BUILD_TYPE=Release"],
                              stdout=subprocess.DEVNULL)
    r = subprocess.run(["cmake", "--build", BUILD, "-j8"], capture_output=True, text=True, timeout=3000)   # no-op when up to date
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return bins


def _write_maps(dirpath, seed, binary=False, text=True, flat=False):
    """PartitionMat pair for the tiny sequence through the PRODUCT's writers; flags from the oracle's post-processing of random
    valid partitions (the nets need a GPU; what is under test here is the file contract and its consumer)."""
    from oracle import postproc as P
    from pmp_vvc_tip2023_amd import engine, synth
    P.build()
    os.makedirs(dirpath, exist_ok=True)
    n = F * (H // 64) * (W // 64)
    for cf, comp in ((1, "Luma"), (2, "Chroma")):
        qt, bt, dire = synth.random_partition_batch(n, seed + cf, cf, 0.15)
        if flat:                                            # no split anywhere below the 64x64 block
            qt = np.zeros_like(qt); bt = np.zeros_like(bt); dire = np.zeros_like(dire)
        hor, ver, q8, d8 = P.seq_post_process(qt, bt, dire, comp, F, W, H, None)
        stem = os.path.join(dirpath, "%s_%s_QP%d_PartitionMat" % (SEQ, comp, QP))
        if text:
            engine.write_partition_file(stem + ".txt", F, H, W, hor, ver, q8.astype(np.uint8), d8)
        if binary:
            engine.write_partition_binary(stem + ".pmpb", F, H, W, hor, ver, q8.astype(np.uint8), d8)


def _workdir(tmp_path, name):
    from pmp_vvc_tip2023_amd import synth
    d = tmp_path / name
    d.mkdir()
    y, u, v = synth.recipe_r_frames(F, H, W, 5)
    with open(d / (SEQ + ".yuv"), "wb") as f:
        for i in range(F):
            f.write(y[i].tobytes()); f.write(u[i].tobytes()); f.write(v[i].tobytes())
    (d / "seq.cfg").write_text("InputFile : %s.yuv\nInputBitDepth : 8\nFrameRate : 30\nFrameSkip : 0\nSourceWidth : %d\n"
                               "SourceHeight : %d\nFramesToBeEncoded : %d\nLevel : 4\n" % (SEQ, W, H, F))
    return d


def _encode(binary, cwd, env=None):
    return subprocess.run([binary, "-c", "seq.cfg", "-c", ENC_CFG, "-f", str(F), "-ts", "1", "-q", str(QP), "-b", "out.bin", "-o", "rec.yuv",
                           "--SEIDecodedPictureHash=1"], cwd=str(cwd), capture_output=True, text=True, timeout=900, env=env)


def test_vtm_parses_product_files_encodes_and_decodes(vtm, tmp_path):
    d = _workdir(tmp_path, "text")
    _write_maps(str(d / "PartitionMat"), 77)
    r = _encode(vtm["EncoderApp"], d)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "memory request finished" in r.stdout and "parse finished" in r.stdout      # EncAppCfg.cpp:4298, :4400
    assert r.stdout.count("I-SLICE") == F
    dec = subprocess.run([vtm["DecoderApp"], "-b", "out.bin", "-o", "dec.yuv"], cwd=str(d), capture_output=True, text=True, timeout=300)
    assert dec.returncode == 0 and dec.stdout.count("(OK)") == F, dec.stdout[-2000:]
    assert open(d / "rec.yuv", "rb").read() == open(d / "dec.yuv", "rb").read()
    # other maps -> another stream: the encoder follows what the files say
    d2 = _workdir(tmp_path, "flat")
    _write_maps(str(d2 / "PartitionMat"), 77, flat=True)
    r2 = _encode(vtm["EncoderApp"], d2)
    assert r2.returncode == 0
    assert open(d / "out.bin", "rb").read() != open(d2 / "out.bin", "rb").read()
    # N2: the hooked encoder on the binary side channel alone -> the identical stream
    d3 = _workdir(tmp_path, "pmpb")
    _write_maps(str(d3 / "PartitionMat"), 77, binary=True, text=False)
    r3 = _encode(vtm["EncoderAppHook"], d3)
    assert r3.returncode == 0, r3.stdout[-2000:] + r3.stderr[-2000:]
    assert "pmp_hook: partition maps mmap'ed" in r3.stdout
    assert open(d / "out.bin", "rb").read() == open(d3 / "out.bin", "rb").read()


def test_vtm_missing_files_end_the_encoder(vtm, tmp_path):
    d = _workdir(tmp_path, "none")
    r = _encode(vtm["EncoderApp"], d)
    assert r.returncode == 1 and "cannot open partitionMat file" in r.stderr            # EncAppCfg.cpp:4252-4263
    # N4 wiring: no files at all -> the hook calls into libpmp_hip.so; without a GPU the library refuses (no CPU fallback)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the in-process leg would run")
    r = _encode(vtm["EncoderAppHook"], d)
    assert r.returncode == 1 and "pmp_hook: pmp_create" in r.stderr and "no CPU fallback" in r.stderr
    # a .pmpb of another geometry is refused, not misread
    from pmp_vvc_tip2023_amd import engine
    os.makedirs(d / "PartitionMat")
    z = np.zeros((1, 16, 16), np.uint8)
    for comp in ("Luma", "Chroma"):
        engine.write_partition_binary(str(d / "PartitionMat" / ("%s_%s_QP%d_PartitionMat.pmpb" % (SEQ, comp, QP))), 1, 64, 64, z, z,
                                      np.zeros((1, 8, 8), np.uint8), np.zeros((1, 3, 16, 16), np.int8))
    r = _encode(vtm["EncoderAppHook"], d)
    assert r.returncode == 1 and "pmp_hook:" in r.stderr and ("shorter" in r.stderr or "geometry" in r.stderr)
