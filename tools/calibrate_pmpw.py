"""Write the f16x3 activation-scale exponents into the MTT-net weight files of a model directory (run once, on an MI355X, after
tools/convert_weights.py): every <Comp>_BD_<qp>.pmpw (or .pkl) whose QT partner is present is loaded, the library's calibration pass runs
(include/pmp.h, "Activation scales"), and the file is re-written as .pmpw with "act_exp" in its manifest.  A file that carries its exponents
is loaded without a calibration pass (pmp_load_weights_file), which takes ~20 ms per net off every job's loading.  The manifest also gets
"act_fp": the fingerprints of the MTT tensors and of the QT partner the calibration ran with - the library ignores exponents whose
fingerprints do not match what it has loaded (a QT net replaced since, tensors edited) and calibrates again.
Usage: python tools/calibrate_pmpw.py <model dir> [--device 0]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmp_vvc_tip2023_amd import engine, weights as W


def calibrate_dir(model_dir, device=0, log=print):
    eng = engine.Engine(device, weight_dir=model_dir)
    done = []
    try:
        for comp in ("Luma", "Chroma"):
            for qp in W.QPS:
                try:
                    kq, pq = W.find_net_weights(comp + "_Q", qp, model_dir)
                    kb, pb = W.find_net_weights(comp + "_MSBD", qp, model_dir)
                except FileNotFoundError:
                    continue
                wb = W.load_pmpw(pb)[1] if kb == "pmpw" else W.load_pkl(pb)
                wq = W.load_pmpw(pq)[1] if kq == "pmpw" else W.load_pkl(pq)
                eng.load(comp, qp, msbd_weights=wb)            # the QT net from its file; the MTT tensors as read here (any stored exponents are ignored)
                rep = eng.activation_report(comp, qp)
                out = os.path.join(model_dir, "%s_%d.pmpw" % (W.ref_net_name(comp + "_MSBD"), qp))
                W.save_pmpw(out, comp + "_MSBD", qp, wb, source="calibrated from " + os.path.basename(pb), act_exp=rep["exps"], qt_partner=wq)
                done.append((out, rep["exps"]))
                log("%s: act_exp %s (segment maxima %s)" % (out, rep["exps"], ["%.3g" % m for m in rep["seg_amax"]]))
    finally:
        eng.close()
    return done


if __name__ == "__main__":
    dev = int(sys.argv[sys.argv.index("--device") + 1]) if "--device" in sys.argv else 0
    calibrate_dir(sys.argv[1], dev)
