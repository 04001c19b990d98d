"""Convert the reference's trained_models/*.pkl into .pmpw containers (weights.py), names as the reference's: <Comp>_{Q,BD}_<qp>.
Usage: python tools/convert_weights.py [/root/reference/trained_models] [weights/]      (the shipped weights/ were made this way)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmp_vvc_tip2023_amd import weights as W


def convert_dir(src, dst, log=print):
    """Every <Comp>_{Q,BD}_<qp>.pkl under src -> dst/<same stem>.pmpw (tensors bit for bit, `module.` stripped).  Returns the files written."""
    os.makedirs(dst, exist_ok=True)
    done = []
    for fn in sorted(os.listdir(src)):
        if not fn.endswith(".pkl"):
            continue
        try:
            comp, kind, qp = fn[:-4].split("_")
            qp = int(qp)
        except ValueError:
            log("skipped (not <Comp>_{Q,BD}_<qp>.pkl): " + fn)
            continue
        if comp not in ("Luma", "Chroma") or kind not in ("Q", "BD"):
            log("skipped (not <Comp>_{Q,BD}_<qp>.pkl): " + fn)
            continue
        net = comp + ("_Q" if kind == "Q" else "_MSBD")
        t = W.load_pkl(os.path.join(src, fn))
        out = os.path.join(dst, fn[:-4] + ".pmpw")
        W.save_pmpw(out, net, qp, t, source="reference trained_models/" + fn)
        done.append(out)
        log("%s: %d tensors, %d params" % (out, len(t), sum(a.size for a in t.values())))
    return done


if __name__ == "__main__":
    convert_dir(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/trained_models", sys.argv[2] if len(sys.argv) > 2 else W.default_weight_dir())
