"""One-command acceptance of a model directory - made for the day the trained MTT nets (*_BD_*.pkl, absent from the reference checkout:
/root/reference/.MISSING_LARGE_BLOBS, trained_models/README.md, Inference_QBD.py:211-222) appear.  Run ON AN MI355X:

    python tools/accept_bd_weights.py <dir holding <Comp>_Q_<qp>.{pkl,pmpw} and <Comp>_BD_<qp>.{pkl,pmpw}> [--out DIR] [--blocks 512]
                                      [--json verdict.json] [--device 0]

  1. convert     every .pkl to the product's .pmpw container (tools/convert_weights.py; .pmpw files are copied) into --out
                 (default <dir>/pmpw); names and shapes checked against the reference's state_dict layout (Model_QBD.py:59-253)
  2. calibrate   the f16x3 activation-scale exponents of every (QT, MTT) pair, written into the MTT manifests together with the
                 fingerprints of the tensors they belong to (tools/calibrate_pmpw.py; include/pmp.h "Activation scales")
  3. verify      per (component, QP): --blocks fresh recipe-R blocks (+ a flat, a saturated, a white-noise and a 2-px checkerboard block)
                 through the torch-CPU oracle and through the HIP path on ALL THREE datapaths: max |logit - oracle| against north_star's
                 absolute 1e-3, blocks over it, the exponents chosen, per-tensor activation maxima of the calibration, range-guard re-runs
                 (a re-run is correct but 2.8x slower: reported, not a parity failure), split flags of the device logits against the oracle's
                 post-processing (bit-exact), end-to-end flag differences against the oracle's own logits (informational)
  4. verdict     one JSON object (stdout, and --json): "ok" = every pair inside the tolerance on every datapath with bit-exact flags

The oracle (oracle/) is the checker here, as in tests/ - this is an acceptance TEST, run by the person who received the files."""
import argparse
import json
import os
import shutil
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np

TOL = 1e-3
DATAPATHS = ("f16x3", "bf16x6", "fp32")


def expected_shapes(net):
    """{tensor name: shape} of the reference's state_dict for this net (module. stripped)."""
    from pmp_vvc_tip2023_amd import synth
    comp, kind = net.split("_", 1)
    if kind == "MSBD":
        return dict(synth.msbd_tensor_shapes(comp))
    luma = comp == "Luma"
    k1, cin = (9, 1) if luma else (5, 3)
    kq = 5 if luma else 3
    sh = {"conv_q1.weight": (32, cin, k1, k1), "conv_q1.bias": (32,), "conv_q2.weight": (1, 8, 3, 3), "conv_q2.bias": (1,)}
    for i, (ci, co, k) in enumerate(((32, 64, kq), (64, 64, kq), (64, 32, 3), (128, 32, 3), (32, 32, 3), (32, 8, 3)), 1):
        sh["resblock_q%d.left.0.weight" % i] = (co, ci, k, k)
        sh["resblock_q%d.left.2.weight" % i] = (co, co, k, k)
        if ci != co:
            sh["resblock_q%d.shortcut.0.weight" % i] = (co, ci, 1, 1)
    return sh


def convert(src, out, log):
    import convert_weights
    os.makedirs(out, exist_ok=True)
    made = convert_weights.convert_dir(src, out, log)
    for fn in sorted(os.listdir(src)):
        if fn.endswith(".pmpw") and os.path.abspath(src) != os.path.abspath(out) and not os.path.exists(os.path.join(out, fn)):
            shutil.copy(os.path.join(src, fn), os.path.join(out, fn))
            made.append(os.path.join(out, fn))
    from pmp_vvc_tip2023_amd import weights as W
    problems, pairs = [], []
    for comp in ("Luma", "Chroma"):
        for qp in W.QPS:
            have = {}
            for net in (comp + "_Q", comp + "_MSBD"):
                p = os.path.join(out, "%s_%d.pmpw" % (W.ref_net_name(net), qp))
                if not os.path.isfile(p):
                    continue
                man, tens = W.load_pmpw(p)
                want = expected_shapes(net)
                miss = sorted(set(want) - set(tens)); extra = sorted(set(tens) - set(want))
                bad = sorted(k for k in want if k in tens and tuple(tens[k].shape) != tuple(want[k]))
                nonfinite = sorted(k for k in tens if not np.isfinite(tens[k]).all())
                if man.get("net") != net or man.get("qp") != qp or miss or bad or nonfinite:
                    problems.append({"file": p, "missing": miss, "wrong_shape": bad, "non_finite": nonfinite, "unexpected": extra,
                                     "manifest": {"net": man.get("net"), "qp": man.get("qp")}})
                have[net] = p
            if len(have) == 2:
                pairs.append((comp, qp))
            elif len(have) == 1:
                problems.append({"pair": "%s QP%d" % (comp, qp), "error": "only %s is present: a pass needs both nets" % list(have)[0]})
    return pairs, problems


def special_blocks(y, u, v, seed):
    """The first four blocks become the extremes tests/test_gpu_trained_like.py uses: flat black, flat white, white noise, a 2-px checkerboard."""
    rng = np.random.default_rng(seed)
    y[0] = 0; u[0] = 0; v[0] = 0
    y[1] = 255; u[1] = 255; v[1] = 255
    y[2] = rng.integers(0, 256, y[2].shape); u[2] = rng.integers(0, 256, u[2].shape); v[2] = rng.integers(0, 256, v[2].shape)
    y[3] = np.where((np.arange(68)[:, None] // 2 + np.arange(68)[None, :] // 2) % 2, 255, 0)


def verify_pair(eng, out, comp, qp, blocks, log):
    import torch
    from oracle import nets_torch as O, postproc as P
    from pmp_vvc_tip2023_amd import synth, weights as W
    luma = comp == "Luma"
    y, u, v = synth.recipe_r_blocks(blocks, 8800 + qp + (5 if luma else 0))
    special_blocks(y, u, v, qp)
    wq = W.load_pmpw(os.path.join(out, "%s_Q_%d.pmpw" % (comp, qp)))[1]
    man_b, wb = W.load_pmpw(os.path.join(out, "%s_BD_%d.pmpw" % (comp, qp)))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    t0 = time.time()
    oq, obt, od = O.infer_qbd(wq, wb, O.luma_input(y) if luma else O.chroma_input(y, u, v), luma, batch=64)
    with np.errstate(invalid="ignore"):
        o_flags = P.seq_post_process(oq, obt, od, comp, 1, 64 * blocks, 64, None)
    t_oracle = time.time() - t0
    res = {"blocks": blocks, "oracle_s": round(t_oracle, 1), "manifest_act_exp": man_b.get("act_exp"), "manifest_act_fp": man_b.get("act_fp"),
           "logit_range": {"qt": [float(oq.min()), float(oq.max())], "bt": [float(obt.min()), float(obt.max())], "dire": [float(od.min()), float(od.max())]},
           "datapaths": {}}
    ok = True
    for prec in DATAPATHS:
        eng.set_precision(prec)
        eng.load(comp, qp)                     # from the .pmpw files of --out (the engine's weight_dir): the manifest's exponents are used
        eng.clear_saturation()
        hor, ver, q8, d8, qt, bt, dire = eng.infer_postprocess(comp, qp, y, u, v, want_logits=True)
        per_block = np.maximum(np.abs(qt - oq).reshape(blocks, -1).max(1),
                               np.maximum(np.abs(bt - obt).reshape(blocks, -1).max(1), np.abs(dire - od).reshape(blocks, -1).max(1)))
        with np.errstate(invalid="ignore"):
            dh, dv, dq, dd = P.seq_post_process(qt, bt, dire, comp, 1, 64 * blocks, 64, None)
        exact = bool(np.array_equal(hor, dh) and np.array_equal(ver, dv) and np.array_equal(d8, dd) and
                     np.array_equal(q8, np.nan_to_num(dq, nan=0.0).astype(np.uint8)))
        e2e = int(((hor != o_flags[0]).any(axis=(1, 2)) | (ver != o_flags[1]).any(axis=(1, 2)) | (d8 != o_flags[3]).any(axis=(1, 2, 3)) |
                   (q8 != np.nan_to_num(o_flags[2], nan=0.0).astype(np.uint8)).any(axis=(1, 2))).sum())
        # north_star's ABSOLUTE 1e-3 on every natural (recipe-R) block; the four synthetic extremes get it relative to |logit| / 8 beyond
        # Map2Partition's operating range (|logit| <= 8: a net that emits +-300 on a checkerboard - float32 itself: the torch oracle is
        # 6.6e-4 from an fp64 evaluation there); how many natural blocks leave the operating range is reported
        mag = np.maximum(np.abs(oq).reshape(blocks, -1).max(1), np.maximum(np.abs(obt).reshape(blocks, -1).max(1), np.abs(od).reshape(blocks, -1).max(1)))
        tol_b = TOL * np.where(np.arange(blocks) < 4, np.maximum(1.0, mag / 8.0), 1.0)     # relative only for the four synthetic extremes
        over = per_block >= tol_b
        worst = int(np.argmax(per_block / tol_b))
        r = {"max_abs_err": {"qt": float(np.abs(qt - oq).max()), "bt": float(np.abs(bt - obt).max()), "dire": float(np.abs(dire - od).max())},
             "worst_block": worst, "worst_block_max_abs_logit": float(max(np.abs(obt[worst]).max(), np.abs(od[worst]).max(), np.abs(oq[worst]).max())),
             "worst_block_err": float(per_block[worst]), "blocks_over_tolerance": int(over.sum()), "per_block_p99": float(np.quantile(per_block, 0.99)),
             "natural_blocks_outside_operating_range": int((mag[4:] > 8.0).sum()),
             "max_abs_err_inside_operating_range": float(per_block[mag <= 8.0].max()) if (mag <= 8.0).any() else None,
             "flags_bit_exact_on_device_logits": exact, "blocks_differing_end_to_end": e2e,
             "saturation_reruns": int(eng.saturation_reruns()), "saturated": bool(eng.saturated()),
             "within_tolerance": bool(not over.any())}
        if prec == "f16x3":
            rep = eng.activation_report(comp, qp)
            r["activation_exps"] = rep["exps"]
            r["stays_on_the_default_datapath"] = r["saturation_reruns"] == 0
        res["datapaths"][prec] = r
        ok = ok and r["within_tolerance"] and exact
        log("  %s QP%d %-6s max |logit - oracle| %.2e (worst block %d), %d of %d blocks over the tolerance, flags on device logits %s, %d blocks differ end to end, re-runs %d"
            % (comp, qp, prec, per_block.max(), worst, r["blocks_over_tolerance"], blocks, "bit-exact" if exact else "DIFFER", e2e, r["saturation_reruns"]))
    res["ok"] = ok
    return res


def calibration_record(out, comp, qp, device):
    """The per-tensor activation maxima behind the exponents: a calibrating load of the same tensors (the files carry exponents only)."""
    from pmp_vvc_tip2023_amd import engine, weights as W
    e = engine.Engine(device, weight_dir=out)
    try:
        e.load(comp, qp, msbd_weights=W.load_pmpw(os.path.join(out, "%s_BD_%d.pmpw" % (comp, qp)))[1])
        rep = e.activation_report(comp, qp)
    finally:
        e.close()
    return {"exps": rep["exps"], "segment_amax": [float(m) for m in rep["seg_amax"]], "tensor_amax": {n: m for n, _, m in rep["tensors"]}}


def accept(model_dir, out=None, blocks=512, device=0, log=print, convert_only=False):
    out = out or os.path.join(model_dir, "pmpw")
    verdict = {"model_dir": os.path.abspath(model_dir), "out": os.path.abspath(out), "tolerance": TOL, "pairs": {}}
    pairs, problems = convert(model_dir, out, log)
    verdict["convert"] = {"pairs": ["%s QP%d" % p for p in pairs], "problems": problems}
    if convert_only or problems or not pairs:
        verdict["ok"] = bool(pairs) and not problems and convert_only
        if not pairs:
            verdict["error"] = "no (QT, MTT) pair found"
        return verdict
    import calibrate_pmpw
    done = calibrate_pmpw.calibrate_dir(out, device, log=log)
    verdict["calibrate"] = {os.path.basename(p): e for p, e in done}
    from pmp_vvc_tip2023_amd import engine
    eng = engine.Engine(device, weight_dir=out)
    try:
        for comp, qp in pairs:
            key = "%s QP%d" % (comp, qp)
            verdict["pairs"][key] = verify_pair(eng, out, comp, qp, blocks, log)
            verdict["pairs"][key]["calibration"] = calibration_record(out, comp, qp, device)
            if verdict["pairs"][key]["calibration"]["exps"] != verdict["pairs"][key]["manifest_act_exp"]:
                verdict["pairs"][key]["ok"] = False
                verdict["pairs"][key]["error"] = "the manifest's exponents are not what a calibrating load chooses"
    finally:
        eng.close()
    verdict["ok"] = all(p["ok"] for p in verdict["pairs"].values())
    verdict["on_default_datapath"] = all(p["datapaths"]["f16x3"]["stays_on_the_default_datapath"] for p in verdict["pairs"].values())
    return verdict


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("model_dir")
    ap.add_argument("--out", default=None)
    ap.add_argument("--blocks", type=int, default=512)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--json", default=None)
    ap.add_argument("--convert-only", action="store_true", help="steps 1 only (no GPU needed): convert and check names / shapes / pairs")
    a = ap.parse_args()
    v = accept(a.model_dir, a.out, a.blocks, a.device, log=lambda *m: print(*m, file=sys.stderr, flush=True), convert_only=a.convert_only)
    txt = json.dumps(v, indent=1)
    if a.json:
        open(a.json, "w").write(txt + "\n")
    print(txt)
    sys.exit(0 if v["ok"] else 1)
