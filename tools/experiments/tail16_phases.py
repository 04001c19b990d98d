"""Phase timing of the round-6 16x16 tail (tools/experiments/tail16_phases.hip: branch16_kernel with s_memtime stamps) - run on the GPU box.
Prints, per phase, the median over blocks and waves of the time between consecutive stamps, as a share of the block's residency."""
import ctypes, os, subprocess, sys, time
import numpy as np
import torch
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
so = os.path.join(here, "_tail16_phases.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(root, "pmp_vvc_tip2023_amd", "csrc"),
                       "-I" + os.path.join(root, "pmp_vvc_tip2023_amd", "csrc", "hooks"), "-I" + os.path.join(root, "include"), "-o", so, os.path.join(here, "tail16_phases.hip")])
lib = ctypes.CDLL(so)
lib.phases_launch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
N, NST = 4096, 24
dev = "cuda"
x = (torch.randn(2, N * 4 * 4096, device=dev) * 0.5).half()
w = (torch.randn(200000, device=dev) * 4).half()
hw = torch.randn(400, device=dev)
bt = torch.zeros(N * 768, device=dev); dire = torch.zeros_like(bt)
st = torch.zeros((N, 4, NST), dtype=torch.int64, device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    assert lib.phases_launch(x.data_ptr(), N * 4 * 4096, bt.data_ptr(), dire.data_ptr(), w.data_ptr(), hw.data_ptr(), st.data_ptr(), N, 0) == 0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
s = st.cpu().numpy().astype(np.float64)
names = ["start: wstart, fetch, clear, barrier", "park X01 (+ first fetch latency), barrier", "conv1a 9 K-steps", "barrier, park X23, barrier", "conv1b 9 K-steps",
         "epilogue 1, barrier", "conv2 9 K-steps", "barrier, park X01, barrier", "shortcut 2 K-steps, barrier", "epilogue 2, barrier", "RB1 (3 passes, 2 barriers)",
         "epilogue, barrier", "RB2 (3 passes, 2 barriers)", "epilogue fp32, barrier", "head"]
tot = s[:, :, 15] - s[:, :, 0]
tick_ns = dt * 1e9 / (np.median(tot) * (N / 512.0))      # 512 blocks resident at a time: kernel time ~ (N / 512) x one block's residency
print("kernel %.1f us; block residency median %.0f ticks (~%.1f us if a tick is 10 ns; implied %.2f ns per tick)" % (dt * 1e6, np.median(tot), np.median(tot) * 0.01, tick_ns))
for k, nm in enumerate(names):
    d = s[:, :, k + 1] - s[:, :, k]
    print("  %-46s median %7.0f ticks  %5.1f %%   (p10 %6.0f  p90 %6.0f)" % (nm, np.median(d), 100 * np.median(d) / np.median(tot), np.percentile(d, 10), np.percentile(d, 90)))
