"""Phase timing of the round-6 16x16 tails (tools/experiments/tail16_phases.hip = branch16_kernel and qt_rest16_kernel of csrc/chain16.hip with
s_memtime stamps; regenerate the .hip with tools/experiments/gen_tail16_phases.py after changing the kernels) - run on the GPU box.
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
lib.qrest_launch.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int]
N, NST = 4096, 24
dev = "cuda"
x = (torch.randn(2, N * 4 * 4096, device=dev) * 0.5).half()
w = (torch.randn(200000, device=dev) * 4).half()
hw = torch.randn(400, device=dev)
bt = torch.zeros(N * 768, device=dev); dire = torch.zeros_like(bt)
st = torch.zeros((N, 4, NST), dtype=torch.int64, device=dev)


def report(title, launch, names):
    st.zero_()
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        assert launch() == 0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = st.cpu().numpy().astype(np.float64)
    last = len(names)
    tot = s[:, :, last] - s[:, :, 0]
    print("%s: kernel %.1f us; block residency median %.0f ticks (%.2f ns per tick if 512 blocks are resident at a time)" % (title, dt * 1e6, np.median(tot), dt * 1e9 / (np.median(tot) * N / 512.0)))
    for k, nm in enumerate(names):
        d = s[:, :, k + 1] - s[:, :, k]
        print("  %-64s median %7.0f ticks  %5.1f %%   (p10 %6.0f  p90 %6.0f)" % (nm, np.median(d), 100 * np.median(d) / np.median(tot), np.percentile(d, 10), np.percentile(d, 90)))


report("branch16", lambda: lib.phases_launch(x.data_ptr(), N * 4 * 4096, bt.data_ptr(), dire.data_ptr(), w.data_ptr(), hw.data_ptr(), st.data_ptr(), N, 0),
       ["start: weight ring, both fetches, border clear, barrier", "trunk_B.0 = t16_rb64 (5 passes, 6 barriers)", "its second epilogue, barrier", "trunk_B.1 (3 passes, 2 barriers)",
        "epilogue, barrier", "trunk_B.2 (3 passes, 2 barriers)", "epilogue fp32, barrier", "head"])
x5 = torch.rand(N * 2 * 4096, device=dev) * 100
qt = torch.zeros(N * 64, device=dev)
fw = torch.randn(4000, device=dev) * 0.05
report("qt_rest16", lambda: lib.qrest_launch(x5.data_ptr(), qt.data_ptr(), w.data_ptr(), fw.data_ptr(), st.data_ptr(), N),
       ["border clear, multi-scale pool (3 barriers)", "q4 conv1: 4 x (build pair, barrier, 9 K-steps, barrier)", "epilogue, barrier, conv2 9 K-steps",
        "q4 shortcut: 4 K-steps from registers", "barrier, clear C D borders, epilogue x7, barrier", "q5: 2 passes + barriers", "pool epilogue, q6 weights to LDS, barrier, two direct 8x8 convolutions", "head (64 threads)"])
