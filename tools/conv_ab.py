"""In-process A/B timing of conv kernel variants on the full luma step (run on the GPU box).
Usage: python tools/conv_ab.py [rounds] [variants...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pmp_vvc_tip2023_amd import engine, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
variants = [int(v) for v in sys.argv[2:]] or [0, 1, 2]
n = 1024
eng = engine.Engine(0, allow_synthetic_mtt=True)
eng.load("Luma", 22)
dev = torch.device("cuda:0")
y, _, _ = synth.recipe_r_blocks(n, 1)
d_y = torch.from_numpy(y).to(dev)
hor = torch.empty((n, 256), dtype=torch.uint8, device=dev); ver = torch.empty_like(hor)
q8 = torch.empty((n, 64), dtype=torch.uint8, device=dev); d8 = torch.empty((n, 768), dtype=torch.int8, device=dev)
qt = torch.empty((n, 64), device=dev); bt = torch.empty((n, 768), device=dev); dire = torch.empty((n, 768), device=dev)
ref = None
for r in range(rounds):
    for v in variants:
        eng.lib.pmp_debug_set_conv_variant(v)
        eng.infer_postprocess_device("Luma", 22, d_y.data_ptr(), None, None, n, hor.data_ptr(), ver.data_ptr(), q8.data_ptr(), d8.data_ptr(),
                                     qt.data_ptr(), bt.data_ptr(), dire.data_ptr())
        eng.synchronize()
        out = [t.cpu().numpy().copy() for t in (qt, bt, dire, hor, ver)]
        if ref is None:
            ref = out
        same = all(np.array_equal(a, b) for a, b in zip(ref, out))
        eng.ktime_enable(0xFFFF)
        for _ in range(3):
            eng.infer_postprocess_device("Luma", 22, d_y.data_ptr(), None, None, n, hor.data_ptr(), ver.data_ptr(), q8.data_ptr(), d8.data_ptr())
        kt = eng.ktime()
        eng.ktime_enable(0)
        tot = sum(ms for _, ms, _ in kt.values()) / 3
        s = " ".join("%s %.2fms %.0fTF" % (k.replace("conv_mfma_", ""), ms / 3, fl / (ms * 1e-3) / 1e12 if ms else 0) for k, (l, ms, fl) in kt.items() if ms / 3 > 0.3)
        print("round %d variant %d bitwise-same=%s total %.2f ms/step | %s" % (r, v, same, tot, s), flush=True)
