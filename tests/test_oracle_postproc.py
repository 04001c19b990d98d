"""CPU: the oracle's C restatement of the integer side (cutter, eli fix, Map_to_Partition, writer) against
golden vectors produced by the imported reference (tools/gen_golden.py) — bit-exact."""
import hashlib
import os

import numpy as np
import pytest

from conftest import g3_sets, g3b_sets, golden, golden_path


def test_map_to_partition_bit_exact(oracle_lib):
    total = 0
    for tag, cf, qt, bt, dire, hor, ver, dout, leaves in g3_sets():
        h, v, d, lv = oracle_lib.map_to_partition(qt, bt, dire, cf)
        assert np.array_equal(h, hor), (tag, cf)
        assert np.array_equal(v, ver), (tag, cf)
        assert np.array_equal(d, dout), (tag, cf)
        assert np.array_equal(lv, leaves), (tag, cf)
        total += len(qt)
    assert total >= 2000          # >= 2000 reference-generated triples (2204), large trees for both chroma factors


def test_value_range_set_bit_exact(oracle_lib, tmp_path):
    """G3b: the reference's eli_structual_error + map_to_parititon on depth / direction logits of 50 ... 3e38, QT logits of +-1e4, and
    +-inf / NaN in every input - NaN depths included, which the reference carries through max-pool, round and clamp."""
    n_nan = 0
    for cf, qt, bt, dire, tags, fixed, q8, hor, ver, dout, leaves in g3b_sets():
        assert np.abs(bt[np.isfinite(bt)]).max() >= 2.9e38 and np.isnan(bt).any() and np.isinf(dire).any() and np.isnan(qt).any()
        of = oracle_lib.eli_structural_error(qt).reshape(-1, 8, 8)
        assert np.array_equal(of, fixed, equal_nan=True), cf
        h, v, d, lv = oracle_lib.map_to_partition(of, bt, dire, cf)
        for t in np.unique(tags):
            m = tags == t
            assert np.array_equal(h[m], hor[m]) and np.array_equal(v[m], ver[m]) and np.array_equal(d[m], dout[m]), (cf, t)
        assert np.array_equal(lv, leaves), cf
        n_nan += int(np.isnan(fixed).any(axis=(1, 2)).sum())
        # the writer's cast of a NaN depth (Map2Partition.py:403) = the 0 the fixture holds
        k = np.flatnonzero(np.isnan(fixed).any(axis=(1, 2)))[:2]
        p = str(tmp_path / ("nan%d.txt" % cf))
        oracle_lib.write_partition_file(p, 1, 64, 128, hor[k], ver[k], fixed[k], dout[k])
        vals = np.array(open(p).read().split(), np.int64)
        qsec = vals[2 * 16 * 32:2 * 16 * 32 + 8 * 16].reshape(8, 16)
        assert np.array_equal(qsec, np.concatenate([q8[k[0]], q8[k[1]]], axis=1))
    assert n_nan >= 50


def test_eli_structural_error_bit_exact(oracle_lib):
    g = golden("g4_eli.npz")
    out = oracle_lib.eli_structural_error(g["qt"])
    assert np.array_equal(out, g["out"].astype(np.float32))
    # invariants of Metrics.py:612-628: 2x2-constant, values 0..3
    o = out.reshape(-1, 8, 8)
    assert np.array_equal(o[:, ::2, ::2], o[:, 1::2, 1::2]) and o.min() >= 0 and o.max() <= 3


@pytest.mark.parametrize("comp", ["Luma", "Chroma"])
def test_sequence_file_bytes(oracle_lib, comp, tmp_path):
    g = golden("g5_seq_%s.npz" % comp)
    p = str(tmp_path / "o.txt")
    oracle_lib.seq_post_process(g["qt"], g["bt"], g["dire"], comp, int(g["F"]), int(g["W"]), int(g["H"]), p)
    data = open(p, "rb").read()
    assert hashlib.sha256(data).hexdigest() == str(g["sha256"])
    assert data == open(golden_path("g5_partitionmat_%s.txt" % comp), "rb").read()


@pytest.mark.parametrize("bd", [8, 10])
def test_block_cutter(oracle_lib, bd):
    g = golden("g6_cut.npz")
    by, bu, bv = oracle_lib.cut_blocks(g["y%d" % bd], g["u%d" % bd], g["v%d" % bd], bd)
    assert np.array_equal(by, g["by%d" % bd]) and np.array_equal(bu, g["bu%d" % bd]) and np.array_equal(bv, g["bv%d" % bd])
    assert by.shape == (3 * 1 * 2, 68, 68)      # 136x72 -> 2x1 blocks per frame, remainder dropped


def test_real_partitionmat_fixture_invariants():
    """G7: first frame of the reference's own demo output (RaceHorses 416x240 Luma QP22): format invariants
     that the product's writer/post-processing must also satisfy."""
    vals = np.array(open(golden_path("g7_racehorses_luma_qp22_frame0.txt")).read().split(), dtype=np.int64)
    R, C = 48, 96
    assert vals.size == 5 * R * C + R * C // 4
    hor = vals[:R * C].reshape(R, C); ver = vals[R * C:2 * R * C].reshape(R, C)
    qt = vals[2 * R * C:2 * R * C + R * C // 4].reshape(R // 2, C // 2)
    dire = vals[2 * R * C + R * C // 4:].reshape(3, R, C)
    assert set(np.unique(hor)) <= {0, 1} and set(np.unique(ver)) <= {0, 1}
    assert qt.min() >= 0 and qt.max() <= 3 and dire.min() >= -1 and dire.max() <= 1
    assert hor[::16].all() and ver[:, ::16].all()                  # block top rows / left columns are edges
    assert np.array_equal(qt[::2, ::2], qt[1::2, 1::2])            # QT section is 2x2-constant


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="live reference only exists in the build container")
def test_oracle_against_live_reference(oracle_lib):
    """Fresh random maps through the imported reference (not a committed fixture)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ref_harness as R
    from pmp_vvc_tip2023_amd import synth
    _, _, M2P, _ = R.load()
    seed = int.from_bytes(os.urandom(2), "little")
    for cf in (1, 2):
        qt, bt, dire = synth.random_partition_batch(40, seed + cf, cf, 0.25)
        h, v, d, _ = oracle_lib.map_to_partition(qt, bt, dire, cf)
        for i in range(len(qt)):
            rh, rv, rd = M2P.map_to_parititon(qt[i], bt[i], dire[i], cf)
            assert np.array_equal(rh, h[i]) and np.array_equal(rv, v[i]) and np.array_equal(rd, d[i]), (seed, cf, i)
    # ... and on raw random bit patterns (round 6: the kernel is tested against the oracle on these): the QT fix included
    import warnings
    import torch
    _, Met, _, _ = R.load()
    rng = np.random.default_rng(seed)
    for cf in (1, 2):
        for k in range(12):
            if k % 2:
                ql, b, d = (rng.integers(0, 2 ** 32, size=sh, dtype=np.uint64).astype(np.uint32).view(np.float32) for sh in ((8, 8), (3, 16, 16), (3, 16, 16)))
            else:
                q, b, d = synth.random_partition_maps(rng, cf)
                ql = (q + rng.normal(0, 0.2, q.shape)).astype(np.float32)
                for arr in (ql, b, d):
                    m = rng.random(arr.shape) < 0.15
                    arr[m] = rng.integers(0, 2 ** 32, size=int(m.sum()), dtype=np.uint64).astype(np.uint32).view(np.float32)
            with warnings.catch_warnings(), torch.no_grad():
                warnings.simplefilter("ignore")
                fixed = Met.eli_structual_error(torch.from_numpy(ql[None, None].copy())).numpy()[0, 0]
                rh, rv, rd = M2P.map_to_parititon(fixed, b, d, cf)
            of = oracle_lib.eli_structural_error(ql[None]).reshape(8, 8)
            oh, ov, od, _ = oracle_lib.map_to_partition(of[None], b[None], d[None], cf)
            assert np.array_equal(of, fixed, equal_nan=True), (seed, cf, k)
            assert np.array_equal(rh, oh[0]) and np.array_equal(rv, ov[0]) and np.array_equal(rd, od[0]), (seed, cf, k)
