"""CPU: host logic of the driver (the counterpart of Inference_QBD.py) - table/cfg parsing, YUV import, CLI flags."""
import numpy as np

from conftest import golden, golden_path
from pmp_vvc_tip2023_amd import inference_qbd as D


def test_sequence_table_matches_reference_arithmetic():
    """Data file shipped by the reference (VVC_Test_Sequences.txt); sub-frame and block counts as
    Inference_QBD.py:64-74 computes them."""
    names, files, w, h, frames, sub, blocks = D.load_sequences_info(golden_path("VVC_Test_Sequences.txt"), 30)
    assert len(names) == 26 and names[0] == "Tango2" and files[0] == "Tango2_3840x2160_60fps_10bit_420.yuv"
    assert (w[0], h[0], frames[0]) == (3840, 2160, 294) and sub[0] == 10 and blocks[0] == 60 * 33 * 10
    i = names.index("RaceHorses")
    assert (w[i], h[i], frames[i]) == (416, 240, 300)
    assert D.load_sequences_info(golden_path("VVC_Test_Sequences.txt"), 8)[5][i] == 38   # demo fixtures hold 38 frames
    assert D.strip_yuv_suffix(files[i]) == "RaceHorses_416x240_30"
    # identical to the reference's rstrip(".yuv") for every shipped name
    assert all(D.strip_yuv_suffix(f) == f.rstrip(".yuv") for f in files)


def test_per_sequence_cfg():
    path, is10 = D.parse_seq_cfg(golden_path("RaceHorses_416x240_30.cfg"))
    assert path == "RaceHorses_416x240_30.yuv" and is10 is False


def test_import_yuv420_subsampling(tmp_path):
    g = golden("g6_cut.npz")
    for bd in (8, 10):
        y, u, v = g["y%d" % bd], g["u%d" % bd], g["v%d" % bd]
        p = tmp_path / ("f%d.yuv" % bd)
        with open(p, "wb") as f:
            for i in range(y.shape[0]):
                f.write(y[i].tobytes()); f.write(u[i].tobytes()); f.write(v[i].tobytes())
        for ratio in (1, 2):
            yy, uu, vv = D.import_yuv420(str(p), 136, 72, 3, ratio, bd == 10)
            assert np.array_equal(yy, y[::ratio]) and np.array_equal(uu, u[::ratio]) and np.array_equal(vv, v[::ratio])


def test_cli_flags_match_reference_defaults():
    a = D.build_parser().parse_args([])
    assert (a.jobID, a.inputDir, a.outDir, a.batchSize, a.startSeqID, a.seqNum) == ("0000", "/input/", "/output/", 200, 0, 22)
    assert a.ssRatio == 30
