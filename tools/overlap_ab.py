#!/usr/bin/env python3
"""Overlap mode (PMP_OVERLAP=1: the chunks of a call alternate between two streams / two workspaces) against the default, on the GPU box:
records of a 4096-block call bit-identical, then alternating bench.py runs of both settings (fresh processes).
    python tools/overlap_ab.py [rounds] [comp]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
comp = sys.argv[2] if len(sys.argv) > 2 else "Luma"


def check():
    import numpy as np, torch
    from pmp_vvc_tip2023_amd import engine, synth
    n = 4096 + 37
    y, u, v = synth.recipe_r_blocks(n, 3)
    dev = torch.device("cuda:0")
    d = [torch.from_numpy(a).to(dev) for a in (y, u, v)]
    out = []
    for ov in ("0", "1"):
        os.environ["PMP_OVERLAP"] = ov
        e = engine.Engine(0, allow_synthetic_mtt=True)
        e.load(comp, 22)
        rec = torch.empty((n, 1344), dtype=torch.uint8, device=dev)
        ch = comp == "Chroma"
        for _ in range(2):
            e.infer_postprocess_records_device(comp, 22, d[0].data_ptr(), d[1].data_ptr() if ch else None, d[2].data_ptr() if ch else None, n, rec.data_ptr())
        e.synchronize()
        out.append(rec.cpu().numpy().copy())
        e.close()
    print("records of %d blocks identical with and without overlap: %s" % (n, np.array_equal(out[0], out[1])), flush=True)


check()
res = {"0": [], "1": []}
for r in range(rounds):
    for ov in ("0", "1"):
        env = dict(os.environ, PMP_OVERLAP=ov)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--cpu-sample", "0", "--no-extras", "--comp", comp],
                           capture_output=True, text=True, env=env)
        if p.returncode != 0:
            print(p.stderr[-2000:]); raise SystemExit(1)
        d = json.loads(p.stdout.strip().splitlines()[-1])
        res[ov].append(d["ms_per_step"])
        print("round %d PMP_OVERLAP=%s %.3f ms/step  dominant %.1f TF" % (r, ov, d["ms_per_step"], d["roofline"]["achieved"]), flush=True)
for ov in ("0", "1"):
    v = sorted(res[ov])
    print("PMP_OVERLAP=%s median %.3f ms/step (min %.3f)" % (ov, v[len(v) // 2], v[0]))
