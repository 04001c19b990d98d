"""HBM access-pattern probe (run on the GPU box): a 1 GiB tensor ([n 1024][8 group planes][64][64] pixels of 32 B) copied by 256-thread
workgroups (a) linearly, 64 KB each, (b) in the convolution kernels' pattern - one 16x16 tile, 16 rows x 512 B at a 2 KB stride in each of
the 8 planes, (c) the same plus the halo reads of a 3x3 kernel, (d) from a tile-major layout (8 KB contiguous per tile and plane)."""
import ctypes, os, subprocess, sys, time
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "_tile_copy.so")
if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "tile_copy.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "tile_copy.hip")])
lib = ctypes.CDLL(so)
lib.probe_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a = torch.randint(0, 255, (nb * 8 * 64 * 64 * 32,), dtype=torch.uint8, device="cuda")
b = torch.empty_like(a)
names = ["linear, 64 KB per workgroup", "16x16 tile pattern (512 B segments, 2 KB stride)", "tile pattern + 3x3 halo reads", "tile-major layout (8 KB contiguous per plane)"]
for rnd in range(2):
    for mode in range(4):
        for _ in range(2):
            assert lib.probe_launch(mode, a.data_ptr(), b.data_ptr(), nb) == 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            lib.probe_launch(mode, a.data_ptr(), b.data_ptr(), nb)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        if mode != 2:
            assert torch.equal(a, b)
        print("%-52s %.3f ms  %.2f TB/s (read + write of %.2f GB each)" % (names[mode], dt * 1e3, 2 * a.numel() / dt / 1e12, a.numel() / 1e9), flush=True)
