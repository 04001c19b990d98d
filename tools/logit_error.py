"""Max |logit - reference golden| per datapath over all 16 nets (run on the GPU box; reads tests/golden only)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pmp_vvc_tip2023_amd import engine

g1 = np.load(os.path.join(ROOT, "tests", "golden", "g1_qt.npz"))
g2 = np.load(os.path.join(ROOT, "tests", "golden", "g2_msbd.npz"))
eng = engine.Engine(0, allow_synthetic_mtt=True)
for mode in ("f16x3", "bf16x6", "fp32"):
    eng.set_precision(mode)
    worst_q, worst_m, where = 0.0, 0.0, ""
    for comp in ("Luma", "Chroma"):
        for qp in (22, 27, 32, 37):
            qt, bt, dire = eng.inference_pre_QBD(comp, qp, g1["block_y"], g1["block_u"], g1["block_v"])
            eq = float(np.abs(qt - g1["qt_%s_%d" % (comp, qp)]).max())
            em = 0.0
            for k in range(3):
                ref = g2["out%d_%s_%d" % (k, comp, qp)]
                em = max(em, float(np.abs(bt[:8, k] - ref[:, 0]).max()), float(np.abs(dire[:8, k] - ref[:, 1]).max()))
            if eq > worst_q:
                worst_q, where = eq, "%s QP%d" % (comp, qp)
            worst_m = max(worst_m, em)
    print("%-7s max|dQT| = %.2e (%s)   max|dMTT| = %.2e   (16 golden blocks x 8 QT nets, 8 blocks x 8 MTT nets)" % (mode, worst_q, where, worst_m), flush=True)
eng.close()
