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


def test_cli_extension_flags_default_to_reference_behaviour():
    """Flags this driver adds are off by default; with them the output contract is unchanged (tests/test_gpu_parity.py)."""
    a = D.build_parser().parse_args([])
    assert a.binary is False and a.hostBlocks is False and a.strictBatch is False and a.device is None and a.gpus == 1
    assert a.qps == "22,27,32,37" and a.comps == "Luma,Chroma"
    assert a.overlap is False                                     # round 6: opt-in again - the job-level A/B of round 5 showed no gain
    assert D.build_parser().parse_args(["--overlap"]).overlap is True
    b = D.build_parser().parse_args(["--hostBlocks", "--strictBatch", "--binary", "--batchSize", "7", "--qps", "22", "--comps", "Chroma"])
    assert b.hostBlocks and b.strictBatch and b.binary and b.batchSize == 7 and b.qps == "22" and b.comps == "Chroma"


def test_tile_partition_maps_matches_the_text_layout_and_rejects_bad_arguments():
    """pmp_tile_partition_maps (host code): frame matrices [F][R][C] in the order the VTM parser reads the text file."""
    import numpy as np
    from pmp_vvc_tip2023_amd import engine, _lib
    rng = np.random.default_rng(5)
    F, H, W = 2, 136, 200                                  # cropped to 128 x 192: 2 x 3 blocks per frame
    n = F * (H // 64) * (W // 64)
    hor = rng.integers(0, 2, (n, 16, 16), dtype=np.uint8); ver = rng.integers(0, 2, (n, 16, 16), dtype=np.uint8)
    qt = rng.integers(0, 4, (n, 8, 8), dtype=np.uint8); dire = rng.integers(-1, 2, (n, 3, 16, 16)).astype(np.int8)
    oh, ov, oq, od = engine.tile_partition_maps(F, H, W, hor, ver, qt, dire)
    assert oh.shape == (F, 32, 48) and oq.shape == (F, 16, 24) and od.shape == (F, 3, 32, 48)
    for f in range(F):
        for br in range(2):
            for bc in range(3):
                b = (f * 2 + br) * 3 + bc
                assert np.array_equal(oh[f, br * 16:(br + 1) * 16, bc * 16:(bc + 1) * 16], hor[b])
                assert np.array_equal(ov[f, br * 16:(br + 1) * 16, bc * 16:(bc + 1) * 16], ver[b])
                assert np.array_equal(oq[f, br * 8:(br + 1) * 8, bc * 8:(bc + 1) * 8], qt[b])
                assert np.array_equal(od[f, :, br * 16:(br + 1) * 16, bc * 16:(bc + 1) * 16], dire[b])
    lib = _lib.load()
    assert lib.pmp_tile_partition_maps(F, H, W, None, None, None, None, None, None, None, None) == -1
    assert b"pmp_tile_partition_maps" in lib.pmp_last_error(None)


def test_missing_mtt_weights_are_an_error_unless_asked_for(tmp_path):
    """Inference_QBD.py:219-222: the reference dies when a model file is missing.  The MTT-net files (*_BD_*) are absent from
    the reference checkout, so the loader refuses them by default and only the explicit opt-in falls back to the documented
    synthetic generator; a mistyped --modelDir is an error, not a silent fallback to the packaged weights."""
    import pytest
    from pmp_vvc_tip2023_amd import weights as W
    assert D.build_parser().parse_args([]).allowSyntheticMTT is False
    assert D.build_parser().parse_args(["--allowSyntheticMTT"]).allowSyntheticMTT is True
    w, src = W.load_net_weights("Luma_Q", 22)                       # the QT nets are real files
    assert src.endswith("Luma_Q_22.pmpw") and "conv_q1.weight" in w
    with pytest.raises(FileNotFoundError) as e:
        W.load_net_weights("Luma_MSBD", 22)
    assert "Luma_BD_22" in str(e.value) and "allowSyntheticMTT" in str(e.value)
    w, src = W.load_net_weights("Luma_MSBD", 22, allow_synthetic=True)
    assert src == "synthetic(seed=22)" and len(w) == 72
    with pytest.raises(FileNotFoundError):
        W.load_net_weights("Luma_Q", 22, weight_dir=str(tmp_path), allow_synthetic=True)   # never synthetic QT weights
    # --modelDir: the untouched default may be absent (packaged weights/ are used), an explicit one must exist
    assert D.resolve_model_dir(D.DEFAULT_MODEL_DIR) in (None, D.DEFAULT_MODEL_DIR)
    assert D.resolve_model_dir(str(tmp_path)) == str(tmp_path)
    with pytest.raises(FileNotFoundError):
        D.resolve_model_dir(str(tmp_path / "CTU_Modles"))


def test_time_sta_columns_follow_the_component_not_the_flag_order():
    assert D.COMP_COLUMN == {"Luma": 0, "Chroma": 1}


def test_rank_reads_only_its_frames(tmp_path, monkeypatch):
    """host I/O is the expected 8-GPU limiter, so a rank must read only the frames that hold its block range.
    The plane reader is wrapped to record every read; two ranks over a 5-frame sequence touch disjoint frame sets (one shared
    frame where the block boundary falls inside it) and together every frame."""
    from pmp_vvc_tip2023_amd import parallel
    w, h, fr = 192, 128, 5                                     # 3 x 2 blocks per frame
    rng = np.random.default_rng(9)
    y = rng.integers(0, 256, (fr, h, w), dtype=np.uint8); u = rng.integers(0, 256, (fr, h // 2, w // 2), dtype=np.uint8)
    v = rng.integers(0, 256, (fr, h // 2, w // 2), dtype=np.uint8)
    p = tmp_path / "s.yuv"
    with open(p, "wb") as f:
        for i in range(fr):
            f.write(y[i].tobytes()); f.write(u[i].tobytes()); f.write(v[i].tobytes())
    reads = []
    real = D._read_plane

    def spy(fp, a):
        reads.append((fp.tell(), a.size))
        return real(fp, a)
    monkeypatch.setattr(D, "_read_plane", spy)
    per_frame, n_total = 6, 6 * fr
    frame_bytes = w * h * 3 // 2
    touched = []
    for rank in range(2):
        reads.clear()
        lo, hi = parallel.shard_bounds(n_total, rank, 2)
        f0, f1 = D.shard_frames(lo, hi, per_frame)
        yy, uu, vv = D.import_yuv420(str(p), w, h, fr, 1, False, frames=(f0, f1))
        assert np.array_equal(yy, y[f0:f1]) and np.array_equal(uu, u[f0:f1]) and np.array_equal(vv, v[f0:f1])
        frames_read = sorted({off // frame_bytes for off, _ in reads})
        assert frames_read == list(range(f0, f1)) and len(reads) == 3 * (f1 - f0)
        touched.append(set(frames_read))
    assert touched[0] == {0, 1, 2} and touched[1] == {2, 3, 4}   # 15 blocks each: the boundary falls inside frame 2
    # temporal sub-sampling: sub-frame k is file frame k * ratio
    reads.clear()
    yy, _, _ = D.import_yuv420(str(p), w, h, fr, 2, False, frames=(1, 3))
    assert np.array_equal(yy, y[[2, 4]]) and sorted({off // frame_bytes for off, _ in reads}) == [2, 4]
    # out-of-range requests are clipped, an empty range reads nothing
    reads.clear()
    yy, _, _ = D.import_yuv420(str(p), w, h, fr, 1, False, frames=(4, 9))
    assert yy.shape[0] == 1 and np.array_equal(yy[0], y[4])
    reads.clear()
    assert D.import_yuv420(str(p), w, h, fr, 1, False, frames=(3, 3))[0].shape[0] == 0 and not reads
