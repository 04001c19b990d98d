"""Full-size parity campaign (opt-in: the file name keeps it out of the default collection; run it as
    python -m pytest tests/campaign_gpu.py -m gpu -s -q
on the GPU box, about 8 minutes of host time for the torch oracle).  BASELINE.json configs[1]'s size - 4096 fresh recipe-R blocks - for
every component and QP on the default datapath: logits of the HIP path against the torch fp32 oracle (tolerance 1e-3), the HIP
post-processing of the DEVICE logits against the C oracle's (bit-exact, every block), and the end-to-end flags against the oracle's own
(only blocks with a value inside the logit difference of a rounding boundary may differ).  Prints one line per net; the lines of the
last run are kept in profiles/."""
import os

import numpy as np
import pytest

import conftest  # noqa: F401 - puts tools/ on sys.path
import trained_like  # tools/trained_like.py: test-weight data (round 6: out of the product package)
import torch

TOL = 1e-3
N = int(os.environ.get("PMP_CAMPAIGN_BLOCKS", "4096"))
PRECISION = os.environ.get("PMP_CAMPAIGN_PRECISION", "f16x3")      # f16x3 (default datapath) | bf16x6 | fp32
MTT_WEIGHTS = os.environ.get("PMP_CAMPAIGN_MTT", "synthetic")      # synthetic (uniform, seed = qp) | trained_like (trained_like.msbd_weights, round 5)


@pytest.fixture(scope="module")
def eng():
    from pmp_vvc_tip2023_amd import engine
    e = engine.Engine(0, allow_synthetic_mtt=True)
    e.set_precision(PRECISION)
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("comp", ["Luma", "Chroma"])
@pytest.mark.parametrize("qp", [22, 27, 32, 37])
def test_full_size_parity(eng, comp, qp):
    from oracle import nets_torch as O, postproc as P
    from pmp_vvc_tip2023_amd import synth, weights as W
    P.build()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    luma = comp == "Luma"
    y, u, v = synth.recipe_r_blocks(N, 5000 + qp + (11 if not luma else 0))
    wq, _ = W.load_net_weights(comp + "_Q", qp)
    if MTT_WEIGHTS == "trained_like":
        wbd, src = trained_like.msbd_weights(comp, qp), "trained-like (tools/trained_like.py)"
        eng.load(comp, qp, msbd_weights=wbd)
        print("\n       trained-like MTT weights: activation exponents %s" % eng.activation_report(comp, qp)["exps"], flush=True)
    else:
        wbd, src = W.load_net_weights(comp + "_MSBD", qp, allow_synthetic=True)
    x = O.luma_input(y) if luma else O.chroma_input(y, u, v)
    oq, obt, od = O.infer_qbd(wq, wbd, x, luma, batch=64)
    hor, ver, q8, d8, qt, bt, dire = eng.infer_postprocess(comp, qp, y, u, v, want_logits=True)
    assert not eng.saturated() and eng.saturation_reruns() == 0
    err = (float(np.abs(qt - oq).max()), float(np.abs(bt - obt).max()), float(np.abs(dire - od).max()))
    strict = os.environ.get("PMP_CAMPAIGN_REPORT_ONLY", "0") != "1"       # report-only: print every figure below before judging (profiles/)
    assert max(err) < TOL or not strict, "%s QP%d logits off by %s" % (comp, qp, err)
    # (i) post-processing on identical inputs: the device logits through the C oracle - bit-exact, every block
    dh, dv, dq8, dd8 = P.seq_post_process(qt, bt, dire, comp, 1, 64 * N, 64, None)
    exact = np.array_equal(hor, dh) and np.array_equal(ver, dv) and np.array_equal(q8, dq8.astype(np.uint8)) and np.array_equal(d8, dd8)
    assert exact, "%s QP%d: HIP post-processing differs from the oracle's on the same logits" % (comp, qp)
    # (ii) end to end against the oracle's own logits: only near-boundary blocks may differ
    oh, ov, oq8, od8 = P.seq_post_process(oq, obt, od, comp, 1, 64 * N, 64, None)
    margin = max(2e-4, 2 * max(err))
    pooled = oq.reshape(-1, 4, 2, 4, 2).max(axis=(2, 4))
    risky = ((np.abs(pooled - np.floor(pooled) - 0.5) < margin).any(axis=(1, 2)) | (np.abs(obt - np.floor(obt) - 0.5) < margin).any(axis=(1, 2, 3)) |
             (np.abs(np.abs(od) - 0.5) < margin).any(axis=(1, 2, 3)))
    bad = ((hor != oh).any(axis=(1, 2)) | (ver != ov).any(axis=(1, 2)) | (q8 != oq8.astype(np.uint8)).any(axis=(1, 2)) | (d8 != od8).any(axis=(1, 2, 3)))
    assert not (bad & ~risky).any()
    cells = int((hor != oh).sum() + (ver != ov).sum())
    worst = int(np.argmax(np.maximum(np.abs(qt - oq).reshape(N, -1).max(1), np.maximum(np.abs(bt - obt).reshape(N, -1).max(1), np.abs(dire - od).reshape(N, -1).max(1)))))
    per_block = np.maximum(np.abs(qt - oq).reshape(N, -1).max(1), np.maximum(np.abs(bt - obt).reshape(N, -1).max(1), np.abs(dire - od).reshape(N, -1).max(1)))
    print("\n       [%s] per-block max error: median %.2e, p99 %.2e, p99.9 %.2e, max %.2e at block %d (max |logit| there %.1f, of the batch %.1f)"
          % (PRECISION, float(np.median(per_block)), float(np.quantile(per_block, 0.99)), float(np.quantile(per_block, 0.999)), float(per_block.max()), worst,
             float(max(np.abs(oq[worst]).max(), np.abs(obt[worst]).max(), np.abs(od[worst]).max())), float(max(np.abs(oq).max(), np.abs(obt).max(), np.abs(od).max()))), flush=True)
    # the worst blocks against convolutions accumulated in fp64 (activations still stored as fp32): how far is the torch fp32 oracle
    # itself from that on the same blocks?  (the nets are ill-conditioned on a few blocks: any two fp32 summation orders differ there)
    import torch.nn.functional as F

    def conv64(xx, w, b, pad):
        return F.conv2d(xx.double(), w.double(), None if b is None else b.double(), padding=pad).float()
    idx = np.argsort(per_block)[-32:]
    rq, rbt, rd = O.infer_qbd(wq, wbd, x[torch.from_numpy(np.sort(idx))], luma, batch=32, conv=conv64)
    si = np.sort(idx)
    e_hip = max(np.abs(qt[si] - rq).max(), np.abs(bt[si] - rbt).max(), np.abs(dire[si] - rd).max())
    e_orc = max(np.abs(oq[si] - rq).max(), np.abs(obt[si] - rbt).max(), np.abs(od[si] - rd).max())
    print("       the 32 worst blocks against fp64-accumulated convolutions: HIP path %.2e, torch fp32 oracle %.2e" % (e_hip, e_orc), flush=True)
    if not strict:      # the other two datapaths on the same blocks: is the tail the datapath's or the net's?
        for prec in ("fp32", "bf16x6"):
            eng.set_precision(prec)
            q2, b2, d2 = eng.inference_pre_QBD(comp, qp, y, u, v)
            pb2 = np.maximum(np.abs(q2 - oq).reshape(N, -1).max(1), np.maximum(np.abs(b2 - obt).reshape(N, -1).max(1), np.abs(d2 - od).reshape(N, -1).max(1)))
            e2 = max(np.abs(q2[si] - rq).max(), np.abs(b2[si] - rbt).max(), np.abs(d2[si] - rd).max())
            print("       [%s] vs the torch oracle: max %.2e (block %d), p99.9 %.2e, blocks over 1e-3: %d; the same 32 blocks against fp64 accumulation: %.2e"
                  % (prec, float(pb2.max()), int(np.argmax(pb2)), float(np.quantile(pb2, 0.999)), int((pb2 >= TOL).sum()), e2), flush=True)
        eng.set_precision(PRECISION)
        print("       [%s] blocks over 1e-3 against the torch oracle: %d of %d; worst block %d: |HIP - fp64| %.2e, |oracle - fp64| %.2e, |HIP - oracle| %.2e"
              % (PRECISION, int((per_block >= TOL).sum()), N, worst,
                 float(max(np.abs(qt[worst] - rq[list(si).index(worst)]).max(), np.abs(bt[worst] - rbt[list(si).index(worst)]).max(), np.abs(dire[worst] - rd[list(si).index(worst)]).max())),
                 float(max(np.abs(oq[worst] - rq[list(si).index(worst)]).max(), np.abs(obt[worst] - rbt[list(si).index(worst)]).max(), np.abs(od[worst] - rd[list(si).index(worst)]).max())),
                 float(per_block[worst])), flush=True)
    assert e_hip < TOL
    print("%-6s QP%d  %d blocks: max |logit - oracle| qt %.2e bt %.2e dire %.2e | post-processing of the device logits bit-exact (%d flags) | "
          "end to end %d of %d blocks differ (all among the %d with a value within %.1e of a rounding boundary; %d edge cells) | MTT weights: %s"
          % (comp, qp, N, err[0], err[1], err[2], hor.size + ver.size + q8.size + d8.size, int(bad.sum()), N, int(risky.sum()), margin, cells, src), flush=True)
    assert bad.sum() <= max(8, N // 64)
    assert max(err) < TOL or not strict
