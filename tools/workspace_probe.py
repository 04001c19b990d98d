import sys; sys.path.insert(0, '.')
import numpy as np
from pmp_vvc_tip2023_amd import engine, synth
y,u,v = synth.recipe_r_blocks(4096, 1)
for prec in ("f16x3","bf16x6","fp32"):
    for comp in ("Luma","Chroma"):
        e = engine.Engine(0, allow_synthetic_mtt=True); e.set_precision(prec)
        e.infer_postprocess(comp, 22, y[:4], u[:4], v[:4]); w4 = e.workspace_bytes()
        e.infer_postprocess(comp, 22, y, u, v); w = e.workspace_bytes()
        print("%s %s: 4 blocks %.2f MB; 4096 blocks %.2f GB = %.3f MB/block" % (prec, comp, w4/2**20, w/2**30, w/4096/2**20), flush=True)
        e.close()
