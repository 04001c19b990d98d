#!/usr/bin/env python3
"""Same-box, same-process A/B of the launch fusions (pmp_debug_set_fusion: 1 = all, 2 = 16x16 tails only (chain16.hip), 3 = 32x32 ResidualBlocks
only (rbfuse32.hip), 0 = launch per layer): interleaved rounds of the device-resident step (pmp_infer_postprocess_records_device), luma and
chroma, 4096 blocks.  Prints ms per step per arm and round, the median, and the per-class kernel times of one step per arm.
    python tools/fusion_ab.py [rounds] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pmp_vvc_tip2023_amd import engine, synth


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    n = 4096
    dev = torch.device("cuda:0")
    y, u, v = synth.recipe_r_blocks(n, 1)
    d = [torch.from_numpy(a).to(dev) for a in (y, u, v)]
    rec = torch.empty((n, 1344), dtype=torch.uint8, device=dev)
    e = engine.Engine(0, allow_synthetic_mtt=True)
    e.set_precision("f16x3")
    for comp in ("Luma", "Chroma"):
        e.load(comp, 22)
        pu, pv = (d[1].data_ptr(), d[2].data_ptr()) if comp == "Chroma" else (None, None)

        def run(k):
            for _ in range(k):
                e.infer_postprocess_records_device(comp, 22, d[0].data_ptr(), pu, pv, n, rec.data_ptr())
            e.synchronize()
        ARMS = (1, 2, 3, 0)
        NAME = {1: "all fused", 2: "tails only", 3: "rb32 only", 0: "per-layer"}
        ms = {a: [] for a in ARMS}
        ref = {}
        for on in ARMS:
            e.set_fusion(on)
            run(2)
            ref[on] = rec.clone()
        assert all(torch.equal(ref[a], ref[0]) for a in ARMS), "records differ between the fused and the per-layer path"
        for r in range(rounds):
            for on in ARMS:
                e.set_fusion(on)
                run(1)
                t = time.perf_counter()
                run(steps)
                ms[on].append((time.perf_counter() - t) / steps * 1e3)
        for on in ARMS:
            print("%-6s %-10s ms/step per round: %s   median %.3f" % (comp, NAME[on], " ".join("%.3f" % x for x in ms[on]), float(np.median(ms[on]))), flush=True)
        for on in ARMS:
            e.set_fusion(on)
            e.ktime_enable(0xFFFF)
            run(1)
            kt = e.ktime()
            e.ktime_enable(0)
            print("   %-10s %s | launches %d" % (NAME[on], "  ".join("%s %.3f ms" % (k.replace("conv_mfma_", ""), v[1]) for k, v in kt.items()),
                                               sum(v[0] for v in kt.values())), flush=True)
    e.close()


if __name__ == "__main__":
    main()
