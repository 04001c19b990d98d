import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, ctypes as C
from pmp_vvc_tip2023_amd import engine, synth, weights as W, _lib
from oracle import nets_torch as O
y, u, v = synth.recipe_r_blocks(12, 3)
wq, _ = W.load_net_weights("Luma_Q", 22)
wb = synth.synth_msbd_weights("Luma", 22)
oq, obt, od = O.infer_qbd(wq, wb, O.luma_input(y), True)
for order in (("f16x3", "f16x3", "bf16x6", "bf16x6", "fp32", "bf16x6"), ("bf16x6", "f16x3"), ("fp32", "f16x3", "bf16x6")):
    e = engine.Engine(0, allow_synthetic_mtt=True)
    e.set_precision(order[0])
    e.load_pretrain_model("Luma_Q", 22, wq)
    e.load_pretrain_model("Luma_MSBD", 22, wb)
    for prec in order:
        e.set_precision(prec)
        qt, bt, dire = e.inference_pre_QBD("Luma", 22, y)
        print(prec, "qt err %.3g bt err %.3g reruns %d" % (np.abs(qt - oq).max(), np.abs(bt - obt).max(), e.saturation_reruns()), flush=True)
    e.close()
    print("--")
