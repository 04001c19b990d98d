"""Generate the golden fixtures under tests/golden/ from the IMPORTED REFERENCE.

Run in the build container only (needs /root/reference):   python tools/gen_golden.py
Every output is produced by calling the reference's own functions:
  G1  Model_QBD.{Luma,Chroma}_Q_Net          real weights, 4 QPs           -> g1_qt.npz
  G2  Model_QBD.{Luma,Chroma}_MSBD_Net       synthetic weights (synth.py)  -> g2_msbd.npz
  G2b Model_QBD.{Luma,Chroma}_MSBD_Net       TRAINED-LIKE weights (trained_like.msbd_weights: bootstrapped from the real QT
                                             tensors, trunks at 1e3, gated products to 9e3), 4 QPs -> g2b_msbd_trained_like.npz
  G3  Map2Partition.map_to_parititon         random/adversarial maps       -> g3_m2p.npz
  G3b Metrics.eli_structual_error + Map2Partition.map_to_parititon on the value RANGE the nets can hand over: |bt|, |dire| up to
      50, 99.5, 100, 100.5, 300, 1e4, 3e38 (np.round has no clamp), QT logits of +-1e4, +-inf and NaN everywhere -> g3b_m2p_range.npz
  G4  Metrics.eli_structual_error            random logits                 -> g4_eli.npz
  G5  Map2Partition.get_sequence_partition_for_VTM (text bytes)            -> g5_seq.npz + g5_partitionmat.txt
  G6  Inference_QBD.output_block_yuv         8-bit and 10-bit frames       -> g6_cut.npz
  G7  first frame of the reference's own demo PartitionMat file            -> g7_racehorses_luma_qp22_frame0.txt
The script also cross-checks the oracle restatements against the reference while it runs.
"""
import hashlib
import io
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import numpy as np
import torch

import ref_harness as R
import trained_like
from pmp_vvc_tip2023_amd import synth
from oracle import nets_torch as O
from oracle import postproc as P

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
M, Met, M2P, Inf = R.load()
torch.set_num_threads(8)
META = "torch %s (CPU, oneDNN), numpy %s, reference AolinFeng/PMP-VVC-TIP2023 @ /root/reference" % (
    torch.__version__, np.__version__)


def ref_weights(comp, kind, qp):
    return {k: v.numpy() for k, v in
            R.load_state_dict("/root/reference/trained_models/%s_%s_%d.pkl" % (comp, kind, qp)).items()}


# ------------------------------------------------------------------------------------------- G1 / G2
def gen_g1_g2():
    y, u, v = synth.recipe_r_blocks(16, 1)
    g1 = {"block_y": y, "block_u": u, "block_v": v, "meta": np.array(META)}
    g2 = {"meta": np.array(META + "; MSBD weights = synth.synth_msbd_weights(comp, seed=qp); inputs = g1 blocks[:8], q = g1 logits")}
    for comp in ("Luma", "Chroma"):
        luma = comp == "Luma"
        x = O.luma_input(y) if luma else O.chroma_input(y, u, v)
        # the driver's own tensor prep (Inference_QBD.py:194-200) for the chroma input, cross-check
        if not luma:
            xb = torch.FloatTensor(np.expand_dims(y, 1))
            xb = torch.cat([torch.nn.functional.max_pool2d(xb, 2), torch.FloatTensor(np.expand_dims(u, 1)),
                            torch.FloatTensor(np.expand_dims(v, 1))], 1)
            assert torch.equal(xb, x)
        for qp in (22, 27, 32, 37):
            wq = ref_weights(comp, "Q", qp)
            net_q = R.ref_net(comp + "_Q", wq)
            with torch.no_grad():
                q = net_q(x)
                q_or = O.q_forward(wq, x, luma)
            assert (q - q_or).abs().max().item() < 1e-4, "oracle Q restatement drifted"
            g1["qt_%s_%d" % (comp, qp)] = q.numpy()
            wbd = synth.synth_msbd_weights(comp, qp)
            net_bd = R.ref_net(comp + "_MSBD", wbd)
            taps = {}
            with torch.no_grad():
                o = net_bd(x[:8], q[:8])
                o_or = O.msbd_forward(wbd, x[:8], q[:8], luma, taps=taps)
            for a, b in zip(o, o_or):
                assert (a - b).abs().max().item() < 1e-4, "oracle MSBD restatement drifted"
            for i in range(3):
                g2["out%d_%s_%d" % (i, comp, qp)] = o[i].numpy()
            if qp == 22:  # a few intermediates of the restatement (already checked equal at the outputs)
                g2["x5_%s" % comp] = taps["x5"][:2].numpy()
                g2["x_att0_%s" % comp] = taps["x_att0"][:2].numpy()
                g2["out1_raw_%s" % comp] = taps["out1_raw"][:2].numpy()
            # inference_pre_QBD regrouping (Metrics.py:399-402) through the reference function itself
            if qp == 22:
                loader = [(x[:8],)]
                qq, bt, dire = Met.inference_pre_QBD(loader, net_q, net_bd)
                g2["pre_qt_%s" % comp] = qq.numpy(); g2["pre_bt_%s" % comp] = bt.numpy(); g2["pre_dire_%s" % comp] = dire.numpy()
            print("G1/G2", comp, qp, "qt range %.2f..%.2f" % (q.min(), q.max()))
    np.savez_compressed(os.path.join(OUT, "g1_qt.npz"), **g1)
    np.savez_compressed(os.path.join(OUT, "g2_msbd.npz"), **g2)


# ----------------------------------------------------------------------------------------------- G2b
def gen_g2b():
    """The reference's MTT modules holding the trained-like weights: all eight (component, QP) nets on the 16 G1 blocks with the G1
    logits as q.  Also pins (i) the per-tensor activation maxima the GPU tests compare the library's own range report with, and
    (ii) that the power-of-two stress variants (trunk_gain, gate_gain) leave the reference's logits bit-identical."""
    g1 = dict(np.load(os.path.join(OUT, "g1_qt.npz")))
    y, u, v = g1["block_y"], g1["block_u"], g1["block_v"]
    out = {"meta": np.array(META + "; MSBD weights = trained_like.msbd_weights(comp, qp); inputs = g1 blocks, q = g1 logits")}
    for comp in ("Luma", "Chroma"):
        luma = comp == "Luma"
        x = O.luma_input(y) if luma else O.chroma_input(y, u, v)
        for qp in (22, 27, 32, 37):
            q = torch.from_numpy(g1["qt_%s_%d" % (comp, qp)])
            wbd = trained_like.msbd_weights(comp, qp)
            net = R.ref_net(comp + "_MSBD", wbd)
            taps = {}
            with torch.no_grad():
                o = net(x, q)
                o_or = O.msbd_forward(wbd, x, q, luma, taps=taps)
                net2 = R.ref_net(comp + "_MSBD", trained_like.msbd_weights(comp, qp, trunk_gain=64.0, gate_gain=16.0))
                o2 = net2(x, q)
            for a, b in zip(o, o_or):
                assert (a - b).abs().max().item() < 1e-4, "oracle MSBD restatement drifted (trained-like weights)"
            for a, b in zip(o, o2):
                assert torch.equal(a, b), "power-of-two gains changed the reference's logits"
            for i in range(3):
                out["out%d_%s_%d" % (i, comp, qp)] = o[i].numpy()
            xb1 = taps["x5"] * taps["x_att0"]
            xb3 = taps["x4"] * taps["x_att1"]
            out["amax_%s_%d" % (comp, qp)] = np.array([taps[k].abs().max().item() for k in ("x3", "x4", "x5", "x_att0", "x_att1")] +
                                                      [xb1.abs().max().item(), xb3.abs().max().item()], np.float32)
            print("G2b", comp, qp, "logits %.2f..%.2f" % (min(t.min() for t in o), max(t.max() for t in o)),
                  "max |x3 x4 x5 att0 att1 xb1 xb3| =", " ".join("%.0f" % t for t in out["amax_%s_%d" % (comp, qp)]))
    np.savez_compressed(os.path.join(OUT, "g2b_msbd_trained_like.npz"), **out)


# ------------------------------------------------------------------------------------------------ G3
def adversarial_maps(rng):
    """Hand-built edge cases: exact +-0.5 directions, x.5 depths (half-to-even), -0.0,
    QT maps with qt<depth holes, deep TT.TT nests, large/negative depths (np.round has no clamp)."""
    qs, bs, ds = [], [], []

    def add(q, b, d):
        qs.append(q.astype(np.float32)); bs.append(b.astype(np.float32)); ds.append(d.astype(np.float32))

    for cf in (1, 2):
        for _ in range(6):
            q, b, d = synth.random_partition_maps(rng, cf)
            add(q, b + 0.5, d)                       # every depth on a .5 boundary (0.5->0, 1.5->2, 2.5->2)
            add(q, b - 0.5, d * 0.5)                 # directions exactly +-0.5 (inclusive threshold)
            add(q, b, -0.0 * np.ones_like(d))        # -0.0 directions
            add(q, -b, d)                            # negative depths
            add(q, b * 3.0, d)                       # depths far above the legal range
            add(q, np.where(rng.random(b.shape) < 0.3, b + 1, b), np.where(rng.random(d.shape) < 0.3, -d, d))
        # QT holes: child quadrant value smaller than its depth
        for _ in range(6):
            q, b, d = synth.random_partition_maps(rng, cf)
            q2 = q.copy(); q2[rng.integers(0, 8), rng.integers(0, 8)] = 0; q2[rng.integers(0, 8), :] = rng.integers(0, 4)
            add(q2, b, d)
            add(rng.integers(0, 4, size=(8, 8)), b, d)     # fully random QT map
        # deep TT.TT nests on an unsplit 64x64 (QT depth 0): TT-H then TT-V in every part, then TT again
        q = np.zeros((8, 8)); b = np.zeros((3, 16, 16)); d = np.zeros((3, 16, 16))
        cur = np.zeros((16, 16))
        cur[0:4] += 2; cur[4:12] += 1; cur[12:16] += 2; b[0] = cur; d[0] = 1
        cur[:, 0:4] += 2; cur[:, 4:12] += 1; cur[:, 12:16] += 2; b[1] = cur; d[1] = -1
        cur2 = cur.copy(); cur2[4:12, 4:12] += 1; b[2] = cur2; d[2, 4:12, 4:12] = 1
        add(q, b, d)
        add(q, b + rng.normal(0, 0.2, b.shape), d + rng.normal(0, 0.2, d.shape))
        # ambiguous direction: all candidates survive -> large trees
        q = np.zeros((8, 8)); b = np.stack([np.ones((16, 16)), 2 * np.ones((16, 16)), 3 * np.ones((16, 16))]); d = np.zeros((3, 16, 16))
        add(q, b, d)
        add(np.ones((8, 8)), b, d)
        add(2 * np.ones((8, 8)), b, d)
        add(3 * np.ones((8, 8)), b, d)
    return np.stack(qs), np.stack(bs), np.stack(ds)


def run_ref_m2p(qt, bt, dr, cf):
    n = len(qt)
    hor = np.zeros((n, 16, 16), np.uint8); ver = np.zeros((n, 16, 16), np.uint8); dout = np.zeros((n, 3, 16, 16), np.int8)
    for i in range(n):
        h, v, d = M2P.map_to_parititon(qt[i], bt[i], dr[i], cf)
        hor[i], ver[i], dout[i] = h, v, d
    return hor, ver, dout


def gen_g3():
    rng = np.random.default_rng(33)
    out = {"meta": np.array(META + "; inputs: q* = int16/64 quantised valid partitions + noise; r* = raw float32; a* = adversarial")}
    for cf in (1, 2):
        # quantised to multiples of 1/64: exact .5 ties and equal-error ties are frequent
        parts = [synth.random_partition_batch(224, 1000 + cf * 10 + k, cf, s) for k, s in enumerate((0.0, 0.15, 0.3, 0.45))]
        qt = np.concatenate([p[0] for p in parts]); bt = np.concatenate([p[1] for p in parts]); dr = np.concatenate([p[2] for p in parts])
        bt_q = np.rint(bt * 64).astype(np.int16); dr_q = np.rint(dr * 64).astype(np.int16)
        bt = (bt_q / 64.0).astype(np.float32); dr = (dr_q / 64.0).astype(np.float32)
        hor, ver, dout = run_ref_m2p(qt, bt, dr, cf)
        ho, vo, do, leaves = P.map_to_partition(qt, bt, dr, cf)
        assert np.array_equal(hor, ho) and np.array_equal(ver, vo) and np.array_equal(dout, do), "oracle m2p mismatch (quantised)"
        out.update({"q_qt_cf%d" % cf: qt.astype(np.int8), "q_bt64_cf%d" % cf: bt_q, "q_dire64_cf%d" % cf: dr_q,
                    "q_hor_cf%d" % cf: hor, "q_ver_cf%d" % cf: ver, "q_dout_cf%d" % cf: dout, "q_leaves_cf%d" % cf: leaves})
        print("G3 quantised cf", cf, len(qt), "leaves mean %.1f max %d" % (leaves.mean(), leaves.max()))
        # raw float32 noise
        qt, bt, dr = synth.random_partition_batch(128, 2000 + cf, cf, 0.2)
        hor, ver, dout = run_ref_m2p(qt, bt, dr, cf)
        ho, vo, do, leaves = P.map_to_partition(qt, bt, dr, cf)
        assert np.array_equal(hor, ho) and np.array_equal(ver, vo) and np.array_equal(dout, do), "oracle m2p mismatch (raw)"
        out.update({"r_qt_cf%d" % cf: qt.astype(np.int8), "r_bt_cf%d" % cf: bt, "r_dire_cf%d" % cf: dr,
                    "r_hor_cf%d" % cf: hor, "r_ver_cf%d" % cf: ver, "r_dout_cf%d" % cf: dout, "r_leaves_cf%d" % cf: leaves})
    qt, bt, dr = adversarial_maps(rng)
    half = len(qt) // 2
    for cf, sl in ((1, slice(0, half)), (2, slice(half, None))):
        hor, ver, dout = run_ref_m2p(qt[sl], bt[sl], dr[sl], cf)
        ho, vo, do, leaves = P.map_to_partition(qt[sl], bt[sl], dr[sl], cf)
        assert np.array_equal(hor, ho) and np.array_equal(ver, vo) and np.array_equal(dout, do), "oracle m2p mismatch (adversarial)"
        out.update({"a_qt_cf%d" % cf: qt[sl].astype(np.int8), "a_bt_cf%d" % cf: bt[sl], "a_dire_cf%d" % cf: dr[sl],
                    "a_hor_cf%d" % cf: hor, "a_ver_cf%d" % cf: ver, "a_dout_cf%d" % cf: dout, "a_leaves_cf%d" % cf: leaves})
        print("G3 adversarial cf", cf, len(hor), "leaves max", leaves.max())
    # large candidate trees, both chroma factors (the adversarial set's largest tree has 6288 leaves for cf 1 but 1572 for cf 2):
    # ambiguous maps - flat cumulative depths 1, 2, 3 (+ small noise, + random QT depth), directions near 0 so that every split mode
    # survives can_split_mode_list - are searched with the (fast) oracle for the inputs with the most leaves; the REFERENCE then
    # produces the expected outputs for the 24 largest per chroma factor.
    for cf in (1, 2):
        cq, cb, cd = [], [], []
        for k in range(1500):
            qd = int(rng.integers(0, 3))
            q = np.full((8, 8), qd, np.float32)
            if rng.random() < 0.3:
                q[:4, :4] = min(qd + 1, 3)
            base = np.stack([np.full((16, 16), qd * 0 + 1.0), np.full((16, 16), 2.0), np.full((16, 16), 3.0)]).astype(np.float32)
            b = base + rng.normal(0, float(rng.choice([0.0, 0.05, 0.2])), base.shape).astype(np.float32)
            d = rng.normal(0, float(rng.choice([0.0, 0.1, 0.3])), base.shape).astype(np.float32)
            if rng.random() < 0.5:      # one quadrant with a definite structure, the rest ambiguous
                d[:, :8, :8] = 1.0
            cq.append(q); cb.append(b); cd.append(d)
        cq, cb, cd = np.stack(cq), np.stack(cb), np.stack(cd)
        _, _, _, leaves = P.map_to_partition(cq, cb, cd, cf)
        top = np.argsort(-leaves, kind="stable")[:24]
        qt, bt, dr = cq[top], cb[top], cd[top]
        hor, ver, dout = run_ref_m2p(qt, bt, dr, cf)
        ho, vo, do, leaves = P.map_to_partition(qt, bt, dr, cf)
        assert np.array_equal(hor, ho) and np.array_equal(ver, vo) and np.array_equal(dout, do), "oracle m2p mismatch (large trees)"
        out.update({"t_qt_cf%d" % cf: qt.astype(np.int8), "t_bt_cf%d" % cf: bt, "t_dire_cf%d" % cf: dr,
                    "t_hor_cf%d" % cf: hor, "t_ver_cf%d" % cf: ver, "t_dout_cf%d" % cf: dout, "t_leaves_cf%d" % cf: leaves})
        print("G3 large trees cf", cf, len(hor), "leaves min %d max %d" % (leaves.min(), leaves.max()))
    total = sum(out[k].shape[0] for k in out if k.endswith(("_hor_cf1", "_hor_cf2")))
    print("G3 total triples", total)
    assert total >= 2000          # SURVEY 8(c) contract
    np.savez_compressed(os.path.join(OUT, "g3_m2p.npz"), **out)



# ----------------------------------------------------------------------------------------------- G3b
G3B_MAGS = (50.0, 99.5, 100.0, 100.5, 300.0, 1.0e4, 3.0e38)


def g3b_inputs(cf, seed):
    """Valid random partitions (+ N(0, 0.15)) whose depth / direction logits are pushed to the magnitudes a net with trained-scale
    activations can emit (tests/test_gpu_trained_like.py sees +-300 on a checkerboard), and non-finite values.  Variants per magnitude
    X and sign s (the thresholds of can_split_mode_list - 0.7 / 0.3 of a region, Map2Partition.py:142-199 - tolerate outliers, so the
    searches stay non-trivial):
      sprinkle   8..25 % of the cells of bt (all three layers, independently) = s*X
      region     one random rectangle of one bt layer = s*X
      dire       10..30 % of the direction cells = s*X
      scale      the whole triple's bt multiplied so that its largest value is X (s = -1: negated)
      direscale  dire * X  (th_round keeps the structure, the float32 error sums carry the magnitude)
      qtlogit    QT logits: cells of depth >= 1 pushed to +1e4, depth-0 cells to -1e4 (clamp(round(.), 0, 3), Metrics.py:632)
    and for NaN, +inf, -inf: sprinkled into bt, into dire, into the QT logits (single cells, 2x2 windows, whole quadrants, everything);
    and raw random bit patterns ("bits")."""
    rng = np.random.default_rng(seed)
    qs, bs, ds, tags = [], [], [], []

    def base():
        q, b, d = synth.random_partition_maps(rng, cf)
        # noise in steps of 1/64 (exact in float32): ties at .5 and equal-error ties are frequent, and the fixture compresses
        b = b + np.rint(rng.normal(0, 0.15, b.shape) * 64) / 64
        d = d + np.rint(rng.normal(0, 0.15, d.shape) * 64) / 64
        ql = q + np.rint(rng.normal(0, 0.2, q.shape) * 64) / 64
        return ql.astype(np.float32), b.astype(np.float32), d.astype(np.float32)

    def add(tag, q, b, d):
        qs.append(q.astype(np.float32)); bs.append(b.astype(np.float32)); ds.append(d.astype(np.float32)); tags.append(tag)

    def sprinkle(a, frac, val):
        a = a.copy()
        a[rng.random(a.shape) < frac] = val
        return a

    def region(a, val):
        a = a.copy()
        k = rng.integers(0, 3); x = rng.integers(0, 14); y = rng.integers(0, 14)
        a[k, x:x + rng.integers(1, 9), y:y + rng.integers(1, 9)] = val
        return a

    with np.errstate(over="ignore", invalid="ignore"):
        for X in G3B_MAGS:
            for s in (1.0, -1.0):
                for _ in range(4):
                    q, b, d = base(); add("sprinkle", q, sprinkle(b, rng.uniform(0.08, 0.25), s * X), d)
                    q, b, d = base(); add("region", q, region(b, s * X), d)
                    q, b, d = base(); add("dire", q, b, sprinkle(d, rng.uniform(0.1, 0.3), s * X))
                    q, b, d = base(); add("scale", q, b * np.float32(s * X / max(np.abs(b).max(), 1e-6)), d)
                    q, b, d = base(); add("direscale", q, b, d * np.float32(s * X))
        for _ in range(24):
            q, b, d = base()
            add("qtlogit", np.where(np.rint(q) >= 1, np.float32(1e4), np.float32(-1e4)), b, d)
            q, b, d = base()
            add("qtlogit", np.where(rng.random(q.shape) < 0.2, np.float32(rng.choice([-1e4, 1e4])), q), b, d)
        for val in (np.nan, np.inf, -np.inf):
            for _ in range(10):
                q, b, d = base(); add("nf_bt", q, sprinkle(b, rng.uniform(0.02, 0.3), val), d)
                q, b, d = base(); add("nf_bt", q, region(b, val), d)
                q, b, d = base(); add("nf_dire", q, b, sprinkle(d, rng.uniform(0.02, 0.3), val))
                q, b, d = base(); b2 = b.copy(); b2[0, 0, 0] = val; add("nf_bt", q, b2, d)        # one cell of the first leaf's region
                q, b, d = base(); q2 = q.copy(); q2[rng.integers(0, 8), rng.integers(0, 8)] = val; add("nf_qt", q2, b, d)
                q, b, d = base(); q2 = q.copy(); x = 2 * rng.integers(0, 4); y = 2 * rng.integers(0, 4); q2[x:x + 2, y:y + 2] = val; add("nf_qt", q2, b, d)
                q, b, d = base(); q2 = q.copy(); x = 4 * rng.integers(0, 2); y = 4 * rng.integers(0, 2); q2[x:x + 4, y:y + 4] = val; add("nf_qt", q2, b, d)
                q, b, d = base(); add("nf_qt", sprinkle(q, rng.uniform(0.02, 0.5), val), b, d)
                # 13..15 true zeros beside the non-finite cells: check_square_unity's "whole map = 0" branch swallows a NaN (Metrics.py:626-627)
                q2 = np.full((8, 8), 0.1, np.float32)
                for c in rng.choice(16, size=int(rng.integers(1, 4)), replace=False):
                    q2[2 * (c // 4), 2 * (c % 4)] = val
                q, b, d = base(); add("nf_qt", q2, b, d)
            q, b, d = base(); add("nf_qt", np.full_like(q, val), b, d)
            q, b, d = base(); add("nf_bt", q, np.full_like(b, val), d)
            q, b, d = base(); add("nf_dire", q, b, np.full_like(d, val))
            q, b, d = base(); add("nf_all", np.full_like(q, val), np.full_like(b, val), np.full_like(d, val))
        # raw random BIT PATTERNS (every float32 class at once: NaNs with payloads, denormals, huge, tiny, both zeros): 24 triples of pure
        # random bits, 16 valid partitions with 10 % / 20 % of their cells replaced by random bits
        def bits(shape):
            return rng.integers(0, 2 ** 32, size=shape, dtype=np.uint64).astype(np.uint32).view(np.float32)
        for _ in range(24):
            add("bits", bits((8, 8)), bits((3, 16, 16)), bits((3, 16, 16)))
        for k in range(16):
            q, b, d = base()
            f = 0.1 * (1 + k % 2)
            for arr in (q, b, d):
                m = rng.random(arr.shape) < f
                arr[m] = bits(int(m.sum()))
            add("bits", q, b, d)
    return np.stack(qs), np.stack(bs), np.stack(ds), np.array(tags)


def gen_g3b():
    """The reference's own seq_post_process arithmetic (Metrics.py:764-774: eli_structual_error, then map_to_parititon per block, then
    the casts of get_sequence_partition_for_VTM, Map2Partition.py:401-404) on g3b_inputs.  What the reference does with non-finite
    values, pinned here because the kernel reproduces it (pmp.h, "non-finite logits"):
      * bt NaN / +inf: np.round keeps it, `comp_map == 0` and `< 0` are False -> the cell counts as "deeper than any candidate";
        -inf: `< 0` True.  dire NaN: th_round leaves NaN, neither == 1 nor == -1 -> counts as 0; +-inf -> +-1.
      * a non-finite value inside a QT leaf makes every leaf's error inf or NaN; `error_list.index(min(error_list))` then returns 0 -
        Python's min keeps its first argument unless a later one compares less.
      * QT logits: max_pool2d propagates NaN, round and clamp keep it; check_square_unity's comparisons are False on it (a quadrant
        holding one is left alone; the 13..15-zeros branch overwrites it); set_partition_vector does nothing for a NaN depth (no
        edges, directions 0); `.astype(np.uint8)` of NaN gives 0 on x86-64 (numpy warns "invalid value encountered in cast")."""
    import warnings
    out = {"meta": np.array(META + "; inputs = tools/gen_golden.py g3b_inputs(cf, 3300 + cf); qt = RAW logits (eli_structual_error applied)"),
           "mags": np.array(G3B_MAGS, np.float32)}
    for cf in (1, 2):
        ql, bt, dr, tags = g3b_inputs(cf, 3300 + cf)
        n = len(ql)
        with torch.no_grad():
            fixed = Met.eli_structual_error(torch.from_numpy(ql[:, None])).numpy()[:, 0]
        hor = np.zeros((n, 16, 16), np.uint8); ver = np.zeros((n, 16, 16), np.uint8); dout = np.zeros((n, 3, 16, 16), np.int8)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for i in range(n):
                hor[i], ver[i], dout[i] = M2P.map_to_parititon(fixed[i], bt[i], dr[i], cf)
            q8 = fixed.astype(np.float64).astype(np.uint8)                  # seq_qt_map is float64 (Map2Partition.py:388,403)
            # the oracle, while we are here
            of = P.eli_structural_error(ql).reshape(-1, 8, 8)
            assert np.array_equal(of, fixed, equal_nan=True), "oracle eli mismatch on the range set"
            ho, vo, do, leaves = P.map_to_partition(of, bt, dr, cf)
        assert np.array_equal(hor, ho) and np.array_equal(ver, vo) and np.array_equal(dout, do), "oracle m2p mismatch (range set)"
        assert q8[np.isnan(fixed)].sum() == 0
        out.update({"qt_cf%d" % cf: ql, "bt_cf%d" % cf: bt, "dire_cf%d" % cf: dr, "tag_cf%d" % cf: tags, "fixed_cf%d" % cf: fixed,
                    "q8_cf%d" % cf: q8, "hor_cf%d" % cf: hor, "ver_cf%d" % cf: ver, "dout_cf%d" % cf: dout, "leaves_cf%d" % cf: leaves})
        print("G3b cf", cf, n, "triples; leaves mean %.1f max %d; NaN cells in the fixed QT maps: %d" %
              (leaves.mean(), leaves.max(), int(np.isnan(fixed).sum())))
    np.savez_compressed(os.path.join(OUT, "g3b_m2p_range.npz"), **out)


# ------------------------------------------------------------------------------------------------ G4
def gen_g4():
    rng = np.random.default_rng(44)
    n = 4096
    qt = rng.normal(1.2, 1.3, size=(n, 1, 8, 8)).astype(np.float32)
    # make many pooled cells land near / on .5 boundaries and outside 0..3
    qt[:512] = np.rint(qt[:512] * 2) / 2.0
    qt[512:640] *= 3.0
    qt[640:700] = -np.abs(qt[640:700]) * 0.3          # rounds to -0.0
    qt[700:760] = np.abs(qt[700:760]) * 0.2           # all-zero blocks
    for i in range(760, 1400):                        # structured: mostly one depth with a few outliers
        base = rng.integers(0, 4)
        qt[i] = base + rng.normal(0, 0.25, size=(1, 8, 8))
        k = rng.integers(0, 6)
        for _ in range(k):
            qt[i, 0, rng.integers(0, 8), rng.integers(0, 8)] = rng.integers(0, 4)
    with torch.no_grad():
        ref = Met.eli_structual_error(torch.from_numpy(qt)).numpy()
    assert np.array_equal(ref, P.eli_structural_error(qt)), "oracle eli mismatch"
    np.savez_compressed(os.path.join(OUT, "g4_eli.npz"), qt=qt, out=ref.astype(np.int8), meta=np.array(META))
    print("G4", n, "hist", np.bincount(ref.astype(np.int64).ravel()))


# ------------------------------------------------------------------------------------------------ G5
def gen_g5():
    """Tiny 2-frame 128x64 'sequence' (2 blocks per frame) through seq_post_process -> exact text bytes."""
    W, H, F = 128, 64, 2
    for comp, cf in (("Luma", 1), ("Chroma", 2)):
        qt_maps, bt, dr = synth.random_partition_batch(F * 2, 55 + cf, cf, 0.2)
        # raw logits whose 2x2 max-pool rounds to the map: use the map itself plus small noise
        rng = np.random.default_rng(5 + cf)
        qt_logits = (qt_maps + rng.normal(0, 0.2, qt_maps.shape)).astype(np.float32)[:, None]
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "x.txt")
            Met.seq_post_process(torch.from_numpy(qt_logits), bt, dr, comp, F, W, H, p)
            ref_bytes = open(p, "rb").read()
            p2 = os.path.join(td, "y.txt")
            P.seq_post_process(qt_logits, bt, dr, comp, F, W, H, p2)
            assert open(p2, "rb").read() == ref_bytes, "oracle writer mismatch"
        np.savez_compressed(os.path.join(OUT, "g5_seq_%s.npz" % comp), qt=qt_logits, bt=bt, dire=dr,
                            W=W, H=H, F=F, sha256=np.array(hashlib.sha256(ref_bytes).hexdigest()), meta=np.array(META))
        with open(os.path.join(OUT, "g5_partitionmat_%s.txt" % comp), "wb") as f:
            f.write(ref_bytes)
        print("G5", comp, len(ref_bytes), "bytes", ref_bytes.count(b"\n"), "lines")


# ------------------------------------------------------------------------------------------------ G6
def gen_g6():
    """output_block_yuv on 136x72 frames (not multiples of 64: remainder dropped), 8-bit and 10-bit."""
    out = {"meta": np.array(META)}
    for bd in (8, 10):
        y, u, v = synth.recipe_r_frames(3, 72, 136, 60 + bd, bitdepth=bd)
        if bd == 10:  # force the half-even and clip cases of round(x/4): x%4==2 and x>=1022
            y[0, 0, :8] = [2, 6, 10, 14, 1021, 1022, 1023, 1018]
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "f.yuv")
            with open(p, "wb") as f:
                for i in range(3):
                    f.write(y[i].tobytes()); f.write(u[i].tobytes()); f.write(v[i].tobytes())
            by, bu, bv = Inf.output_block_yuv(p, 136, 72, 64, 4, 3, 1, is10bit=(bd == 10))
            # temporal sub-sampling path of import_yuv420 (Inference_QBD.py:78-102)
            by2, bu2, bv2 = Inf.output_block_yuv(p, 136, 72, 64, 4, 3, 2, is10bit=(bd == 10))
        oy, ou, ov = P.cut_blocks(y, u, v, bd)
        assert np.array_equal(by, oy) and np.array_equal(bu, ou) and np.array_equal(bv, ov), "oracle cutter mismatch"
        oy2, ou2, ov2 = P.cut_blocks(y[::2], u[::2], v[::2], bd)
        assert np.array_equal(by2, oy2) and np.array_equal(bu2, ou2) and np.array_equal(bv2, ov2)
        out.update({"y%d" % bd: y, "u%d" % bd: u, "v%d" % bd: v, "by%d" % bd: by, "bu%d" % bd: bu, "bv%d" % bd: bv})
        print("G6", bd, by.shape, bu.shape)
    np.savez_compressed(os.path.join(OUT, "g6_cut.npz"), **out)


# ------------------------------------------------------------------------------------------------ G7
def gen_g7():
    src = "/root/reference/codec/demo/PartitionMat/RaceHorses_416x240_30_Luma_QP22_PartitionMat.txt"
    lines_per_frame = 5 * (48 * 96) + 24 * 48
    with open(src, "rb") as f:
        data = f.read()
    lines = data.split(b"\n")
    assert lines[-1] == b"" and (len(lines) - 1) == 38 * lines_per_frame
    first = b"\n".join(lines[:lines_per_frame]) + b"\n"
    with open(os.path.join(OUT, "g7_racehorses_luma_qp22_frame0.txt"), "wb") as f:
        f.write(first)
    print("G7", len(first), "bytes")


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2b", "g3", "g3b", "g4", "g5", "g6", "g7"]
    if "g1" in which: gen_g1_g2()
    if "g2b" in which: gen_g2b()
    if "g3" in which: gen_g3()
    if "g3b" in which: gen_g3b()
    if "g4" in which: gen_g4()
    if "g5" in which: gen_g5()
    if "g6" in which: gen_g6()
    if "g7" in which: gen_g7()
    print("done ->", OUT)
