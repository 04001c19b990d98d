"""What this box's HBM sustains for plain streaming (run on the GPU box): device-to-device copy, read-only reduction and fill of 4 GiB
buffers through torch.  The 3x3 64->64 convolution moves 6.35 GB per 4096-block launch (PMC) in 1.85 ms = 3.4 TB/s next to its MFMAs."""
import time, torch
n = 1 << 30   # floats = 4 GiB
a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
a.fill_(1.0); torch.cuda.synchronize()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
tc = t(lambda: b.copy_(a)); print("copy 4 GiB -> 4 GiB: %.2f ms = %.2f TB/s (read + write)" % (tc * 1e3, 2 * 4 * n / tc / 1e12))
tr = t(lambda: a.sum()); print("sum of 4 GiB:        %.2f ms = %.2f TB/s (read)" % (tr * 1e3, 4 * n / tr / 1e12))
tf = t(lambda: b.fill_(2.0)); print("fill 4 GiB:          %.2f ms = %.2f TB/s (write)" % (tf * 1e3, 4 * n / tf / 1e12))
