#!/usr/bin/env python3
"""Do kernels of different HIP streams overlap on this box?  Chains of dependent single-workgroup spin kernels (torch.cuda._sleep:
pure latency, ~25 us each, no resources to speak of) alone, on two / three / six streams, and beside a chip-filling kernel chain."""
import time
import torch

dev = "cuda:0"
big = [torch.zeros(64 << 20, device=dev) for _ in range(3)]  # 256 MB each: a chip-filling elementwise kernel (~0.07 ms)
CYC = 60000


def spin(n):
    for _ in range(n):
        torch.cuda._sleep(CYC)


def fill(x, n):
    for _ in range(n):
        x.add_(1.0)


def run(fns, streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f, s in zip(fns, streams):
        with torch.cuda.stream(s):
            f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


s = [torch.cuda.Stream() for _ in range(8)]
spin(20); fill(big[0], 5)
torch.cuda.synchronize()
N = 60
print("ms: 1 spin chain | 2 on one stream | 2 streams | 3 streams | 6 streams | fill alone | fill + 1 spin chain | fill + 3 spin chains")
for rep in range(3):
    r = [run([lambda: spin(N)], [s[0]]),
         run([lambda: spin(N), lambda: spin(N)], [s[0], s[0]]),
         run([lambda: spin(N)] * 2, s[:2]),
         run([lambda: spin(N)] * 3, s[:3]),
         run([lambda: spin(N)] * 6, s[:6]),
         run([lambda: fill(big[0], 20)], [s[0]]),
         run([lambda: fill(big[0], 20), lambda: spin(N)], s[:2]),
         run([lambda: fill(big[0], 20)] + [lambda: spin(N)] * 3, s[:4])]
    print(" ".join(f"{v:8.3f}" for v in r))

# which streams share a hardware queue?  pairs of spin chains: ~1.6 ms = side by side, ~3.1 ms = one behind the other
M = 8
print("pair matrix (ms), streams in creation order:")
for i in range(M):
    row = []
    for j in range(M):
        row.append(run([lambda: spin(30)] * 2, [s[i], s[j]]) if j > i else 0.0)
    print(" ".join(f"{v:6.2f}" for v in row))
print("current/default stream vs each:", " ".join(f"{run([lambda: spin(30)] * 2, [torch.cuda.current_stream(), s[j]]):6.2f}" for j in range(M)))
