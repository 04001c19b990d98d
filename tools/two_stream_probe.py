"""Graph-level overlap probe: does the step get faster when two half-batches are in flight on two streams, so that
the small launches of one half (16x16 / 8x8 layers, stems, glue, post-processing: 12 % of the step at <= 216 TFLOP/s) run beside the
64x64 convolutions of the other?  Two contexts (= two streams, two arenas), each with half of the blocks, enqueued alternately, against
one context with all of them.  Run on the GPU box:  python tools/two_stream_probe.py [blocks] [comp]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pmp_vvc_tip2023_amd import engine, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
comp = sys.argv[2] if len(sys.argv) > 2 else "Luma"
dev = torch.device("cuda:0")
y, u, v = synth.recipe_r_blocks(n, 1)
d_y, d_u, d_v = (torch.from_numpy(a).to(dev) for a in (y, u, v))
chroma = comp == "Chroma"


def run(engs, parts, reps=6):
    recs = [torch.empty((hi - lo, 1344), dtype=torch.uint8, device=dev) for lo, hi in parts]

    def once():
        for e, (lo, hi), r in zip(engs, parts, recs):
            e.infer_postprocess_records_device(comp, 22, d_y[lo:hi].data_ptr(), d_u[lo:hi].data_ptr() if chroma else None,
                                               d_v[lo:hi].data_ptr() if chroma else None, hi - lo, r.data_ptr())
        for e in engs:
            e.synchronize()
    once(); once()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    return (time.perf_counter() - t0) / reps * 1e3, torch.cat(recs).cpu().numpy()


def make(k, chunk=None):
    es = []
    for _ in range(k):
        e = engine.Engine(0, allow_synthetic_mtt=True)
        e.load(comp, 22)
        if chunk:
            e.set_chunk(chunk)
        es.append(e)
    return es

one = make(1)
for rnd in range(2):
    t1, r1 = run(one, [(0, n)])
    print("round %d: one stream, %d blocks: %.2f ms" % (rnd, n, t1), flush=True)
    for k in (2, 3, 4):
        es = make(k)
        parts = [(i * n // k, (i + 1) * n // k) for i in range(k)]
        tk, rk = run(es, parts)
        print("round %d: %d streams x %d blocks: %.2f ms (%+.1f %%)  records identical: %s" % (rnd, k, n // k, tk, 100 * (tk - t1) / t1, np.array_equal(r1, rk)), flush=True)
        for e in es:
            e.close()

# Round 4: the same two streams DE-PHASED - no synchronisation between repetitions, stream B starts behind a pass on a fraction of its blocks,
# so that one half's QT net (3x3 layers: HBM-heavy) tends to run beside the other half's MTT net (5x5 layers: MFMA-heavy).
if os.environ.get("PMP_TWO_STREAM_OFFSET", "1") == "1":
    reps = 8
    es = make(2)
    parts = [(0, n // 2), (n // 2, n)]
    recs = [torch.empty((hi - lo, 1344), dtype=torch.uint8, device=dev) for lo, hi in parts]

    def enq(e, lo, hi, r):
        e.infer_postprocess_records_device(comp, 22, d_y[lo:hi].data_ptr(), d_u[lo:hi].data_ptr() if chroma else None,
                                           d_v[lo:hi].data_ptr() if chroma else None, hi - lo, r.data_ptr())
    for frac in (0.0, 0.25, 0.5):
        for rnd in range(2):
            for e, (lo, hi), r in zip(es, parts, recs):
                enq(e, lo, hi, r)
            for e in es:
                e.synchronize()
            t0 = time.perf_counter()
            if frac:
                lo, hi = parts[1]
                enq(es[1], lo, lo + int((hi - lo) * frac), recs[1])
            for _ in range(reps):
                for e, (lo, hi), r in zip(es, parts, recs):
                    enq(e, lo, hi, r)
            for e in es:
                e.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            t1s, _ = run(one, [(0, n)])
            work = reps + frac / 2
            print("de-phased by %.2f of a half pass, no sync between %d repetitions: %.2f ms per %d blocks (one stream now: %.2f ms, %+.1f %%)"
                  % (frac, reps, dt / work, n, t1s, 100 * (dt / work - t1s) / t1s), flush=True)
    for e in es:
        e.close()
