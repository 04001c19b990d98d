"""How long hipMalloc / hipFree of a workspace-sized buffer take on this pool, and whether a big hipMalloc in one thread holds up
HIP calls of another (run on the GPU box)."""
import ctypes as C, threading, time, sys
import torch
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipFree.argtypes = [C.c_void_p]
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
GB = 1 << 30
for size in (10 * GB, 5 * GB, 2 * GB, 10 * GB):
    ts = []
    for k in range(6):
        p = C.c_void_p()
        t0 = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), size); t1 = time.perf_counter()
        x = torch.empty(1 << 20, device="cuda").fill_(1); torch.cuda.synchronize()
        t2 = time.perf_counter(); hip.hipFree(p); t3 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t3 - t2) * 1e3))
    print("%2d GB: " % (size // GB) + "  ".join("malloc %.0f free %.0f" % t for t in ts) + " ms", flush=True)
# a big hipMalloc on a side thread while the main thread launches small kernels and copies
def side():
    p = C.c_void_p(); t0 = time.perf_counter(); hip.hipMalloc(C.byref(p), 10 * GB); print("side thread: hipMalloc(10 GB) %.0f ms" % ((time.perf_counter() - t0) * 1e3), flush=True); hip.hipFree(p)
for rnd in range(3):
    th = threading.Thread(target=side); th.start()
    worst = 0.0; n = 0
    while th.is_alive():
        t0 = time.perf_counter(); y = torch.empty(1 << 16, device="cuda").fill_(2); torch.cuda.synchronize(); worst = max(worst, time.perf_counter() - t0); n += 1
    th.join()
    print("main thread: %d small launches meanwhile, slowest %.1f ms" % (n, worst * 1e3), flush=True)
