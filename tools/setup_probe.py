"""Where the fixed per-job time goes (run on the GPU box): context creation, weight loading per net, the first pass (workspace
allocation + code-object load), a later pass, teardown."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pmp_vvc_tip2023_amd import engine, synth
t = time.perf_counter
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for rnd in range(2):
    t0 = t(); eng = engine.Engine(0, allow_synthetic_mtt=True); t1 = t()
    print("round %d: Engine() %.1f ms" % (rnd, (t1 - t0) * 1e3))
    for comp in ("Luma", "Chroma"):
        for qp in (22, 27, 32, 37):
            t0 = t(); eng.load(comp, qp); print("  load %s %d: %.1f ms" % (comp, qp, (t() - t0) * 1e3))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 15840
    y, u, v = synth.recipe_r_blocks(min(n, 2048), 1)
    reps = -(-n // y.shape[0])
    d_y = torch.from_numpy(np.concatenate([y] * reps)[:n]).cuda()
    rec = torch.empty((n, 1344), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for k in range(3):
        t0 = t(); eng.infer_postprocess_records_device("Luma", 22, d_y.data_ptr(), None, None, n, rec.data_ptr()); t1 = t(); eng.synchronize(); t2 = t()
        print("  luma pass %d: enqueue %.1f ms, total %.1f ms, workspace %.2f GB" % (k, (t1 - t0) * 1e3, (t2 - t0) * 1e3, eng.workspace_bytes() / 1e9))
    t0 = t(); eng.close(); print("  close %.1f ms" % ((t() - t0) * 1e3))
