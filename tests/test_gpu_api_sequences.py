"""A seeded random walk through the C ABI's state space: weights replaced between calls (benign, trained-like, range-stress), device calls left
in flight, host-pointer calls in between, chunk size, launch fusion, overlap mode, activation scales and datapath switched at random - and
every result compared, BIT FOR BIT, with what a fresh context in default settings returns for the same weights, blocks, datapath and
scale setting.  None of the knobs may change a result (include/pmp.h): this is the test that notices when two features meet badly
(the range guard's deferred re-runs, the load-time calibration, the second stream of overlap mode, the context's shared logit buffers)."""
import numpy as np
import pytest

import conftest  # noqa: F401 - puts tools/ on sys.path
import trained_like  # tools/trained_like.py: test-weight data (round 6: out of the product package)
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


def _weights(kind):
    from pmp_vvc_tip2023_amd import synth
    if kind == "benign":
        return synth.synth_msbd_weights("Luma", 22)
    if kind == "trained":
        return trained_like.msbd_weights("Luma", 22)
    if kind == "trained_k":
        return trained_like.msbd_weights("Luma", 22, trunk_gain=1024.0, gate_gain=64.0)
    w = dict(synth.synth_msbd_weights("Luma", 22))             # "stress": tests/test_gpu_parity.py's range-stress construction
    K = np.float32(2.0 ** 17)
    for k in ("conv_b1_1", "conv_b1_2", "conv_b1_3"):
        w[k + ".weight"] = (w[k + ".weight"] * K).astype(np.float32)
        w[k + ".bias"] = (w[k + ".bias"] * K).astype(np.float32)
    for t in ("trunk_B1.0", "trunk_B2.0", "trunk_B3.0"):
        for k in (".left.0.weight", ".shortcut.0.weight"):
            w[t + k] = (w[t + k] / K).astype(np.float32)
    return w


@pytest.mark.parametrize("seed", [20250, 7, 424242])
def test_random_api_sequences_are_result_neutral(seed):
    from pmp_vvc_tip2023_amd import engine, synth
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    pool_y, _, _ = synth.recipe_r_blocks(1300, 99)
    pool_y[:16] = golden("g1_qt.npz")["block_y"]
    d_pool = torch.from_numpy(pool_y).to(dev)
    ref_cache = {}

    def reference(kind, prec, scales, lo, n):
        """Records of blocks [lo, lo + n) on a fresh context in default settings (one call, nothing else in flight)."""
        key = (kind, prec, scales)
        if key not in ref_cache:
            e = engine.Engine(0, allow_synthetic_mtt=True)
            try:
                e.set_precision(prec)
                e.set_activation_scales(scales)
                e.load("Luma", 22, msbd_weights=_weights(kind))
                rec = torch.empty((pool_y.shape[0], 1344), dtype=torch.uint8, device=dev)
                e.infer_postprocess_records_device("Luma", 22, d_pool.data_ptr(), None, None, pool_y.shape[0], rec.data_ptr())
                e.synchronize()
                ref_cache[key] = rec.cpu().numpy()
            finally:
                e.close()
        return ref_cache[key][lo:lo + n]

    e = engine.Engine(0, allow_synthetic_mtt=True)
    try:
        state = {"kind": "benign", "prec": "f16x3", "scales": True}
        e.load("Luma", 22, msbd_weights=_weights("benign"))
        in_flight = []                                            # (records tensor, expected) of device calls not yet synchronised
        checked = 0
        for step in range(90):
            op = rng.choice(["device", "device", "device", "host", "host_pp", "weights", "chunk", "fusion", "overlap", "scales", "precision", "sync"])
            if op == "device":
                n = int(rng.choice([1, 5, 37, 130, 300, 1100]))
                lo = int(rng.integers(0, pool_y.shape[0] - n + 1))
                rec = torch.zeros((n, 1344), dtype=torch.uint8, device=dev)
                e.infer_postprocess_records_device("Luma", 22, d_pool[lo:].data_ptr(), None, None, n, rec.data_ptr())
                in_flight.append((rec, reference(state["kind"], state["prec"], state["scales"], lo, n), step, dict(state)))
            elif op in ("host", "host_pp"):
                n = int(rng.choice([1, 6, 40]))
                lo = int(rng.integers(0, pool_y.shape[0] - n + 1))
                want = reference(state["kind"], state["prec"], state["scales"], lo, n)
                if op == "host":
                    hor, ver, q8, d8 = e.infer_postprocess("Luma", 22, pool_y[lo:lo + n])
                else:
                    qt, bt, dire = e.inference_pre_QBD("Luma", 22, pool_y[lo:lo + n])
                    hor, ver, q8, d8 = e.post_process(qt, bt, dire, "Luma")
                got = np.concatenate([hor.reshape(n, -1), ver.reshape(n, -1), q8.reshape(n, -1), d8.reshape(n, -1).view(np.uint8)], axis=1)
                assert np.array_equal(got, want), "step %d: host call (%s) differs under %s" % (step, op, state)
                checked += 1
            elif op == "weights":
                state["kind"] = str(rng.choice(["benign", "trained", "trained_k", "stress"]))
                e.load("Luma", 22, msbd_weights=_weights(state["kind"]))      # replacing a net settles the calls in flight first
            elif op == "chunk":
                e.set_chunk(int(rng.choice([3, 64, 500, 4096])))
            elif op == "fusion":
                e.set_fusion(int(rng.integers(0, 4)))
            elif op == "overlap":
                e.set_overlap(bool(rng.integers(0, 2)))
            elif op == "scales":
                state["scales"] = bool(rng.integers(0, 2))
                e.set_activation_scales(state["scales"])
            elif op == "precision":
                state["prec"] = str(rng.choice(["f16x3", "f16x3", "fp32", "bf16x6"]))
                e.set_precision(state["prec"])
            if op == "sync" or len(in_flight) >= 4 or step == 89:
                e.synchronize()
                for rec, want, st, s0 in in_flight:
                    assert np.array_equal(rec.cpu().numpy(), want), "device call of step %d differs (state then: %s)" % (st, s0)
                    checked += 1
                in_flight = []
        assert checked >= 20
    finally:
        e.close()


def test_random_api_sequences_chroma():
    """The same walk, shorter, on the chroma nets (three input planes, 32x32 trunks: every launch of the MTT net past the stem is a fused or
    32x32 kernel there)."""
    from pmp_vvc_tip2023_amd import engine, synth
    rng = np.random.default_rng(31337)
    dev = torch.device("cuda:0")
    y, u, v = synth.recipe_r_blocks(1100, 98)
    d = [torch.from_numpy(a).to(dev) for a in (y, u, v)]
    kinds = {"benign": lambda: synth.synth_msbd_weights("Chroma", 27), "trained": lambda: trained_like.msbd_weights("Chroma", 27),
             "trained_k": lambda: trained_like.msbd_weights("Chroma", 27, trunk_gain=4096.0, gate_gain=16.0, att_gain=8.0)}
    refs = {}

    def reference(kind, prec):
        if (kind, prec) not in refs:
            e = engine.Engine(0, allow_synthetic_mtt=True)
            try:
                e.set_precision(prec)
                e.load("Chroma", 27, msbd_weights=kinds[kind]())
                rec = torch.empty((y.shape[0], 1344), dtype=torch.uint8, device=dev)
                e.infer_postprocess_records_device("Chroma", 27, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), y.shape[0], rec.data_ptr())
                e.synchronize()
                refs[(kind, prec)] = rec.cpu().numpy()
            finally:
                e.close()
        return refs[(kind, prec)]

    e = engine.Engine(0, allow_synthetic_mtt=True)
    try:
        kind, prec = "benign", "f16x3"
        e.load("Chroma", 27, msbd_weights=kinds[kind]())
        pending = []
        for step in range(45):
            op = rng.choice(["device", "device", "weights", "chunk", "fusion", "overlap", "precision", "sync"])
            if op == "device":
                n = int(rng.choice([2, 33, 257, 1030]))
                lo = int(rng.integers(0, y.shape[0] - n + 1))
                rec = torch.zeros((n, 1344), dtype=torch.uint8, device=dev)
                e.infer_postprocess_records_device("Chroma", 27, d[0][lo:].data_ptr(), d[1][lo:].data_ptr(), d[2][lo:].data_ptr(), n, rec.data_ptr())
                pending.append((rec, reference(kind, prec)[lo:lo + n], step, kind, prec))
            elif op == "weights":
                kind = str(rng.choice(list(kinds)))
                e.load("Chroma", 27, msbd_weights=kinds[kind]())
            elif op == "chunk":
                e.set_chunk(int(rng.choice([7, 128, 4096])))
            elif op == "fusion":
                e.set_fusion(int(rng.integers(0, 4)))
            elif op == "overlap":
                e.set_overlap(bool(rng.integers(0, 2)))
            elif op == "precision":
                prec = str(rng.choice(["f16x3", "f16x3", "fp32"]))
                e.set_precision(prec)
            if op == "sync" or len(pending) >= 3 or step == 44:
                e.synchronize()
                for rec, want, st, k0, p0 in pending:
                    assert np.array_equal(rec.cpu().numpy(), want), "chroma device call of step %d differs (%s, %s)" % (st, k0, p0)
                pending = []
        assert e.saturation_reruns() == 0
    finally:
        e.close()
