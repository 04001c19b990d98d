"""Convert the reference's trained_models/*_Q_*.pkl into weights/*.pmpw (run once in the build container).
Usage: python tools/convert_weights.py [/root/reference/trained_models] [weights/]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmp_vvc_tip2023_amd import weights as W

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/trained_models"
dst = sys.argv[2] if len(sys.argv) > 2 else W.default_weight_dir()
os.makedirs(dst, exist_ok=True)
for fn in sorted(os.listdir(src)):
    if not fn.endswith(".pkl"):
        continue
    comp, kind, qp = fn[:-4].split("_")
    net = comp + ("_Q" if kind == "Q" else "_MSBD")
    t = W.load_pkl(os.path.join(src, fn))
    out = os.path.join(dst, fn[:-4] + ".pmpw")
    W.save_pmpw(out, net, int(qp), t, source="reference trained_models/" + fn)
    print(out, len(t), "tensors", sum(a.size for a in t.values()), "params")
