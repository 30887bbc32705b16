"""PCIe ceilings on the GPU box: pinned host <-> device copies of 100 MB and of 8 MiB pieces (torch / hipMemcpyAsync)."""
import time, torch
dev = torch.device("cuda", 0)
n = 100_000_000
h = torch.empty(n, dtype=torch.uint8, pin_memory=True); h.zero_()
d = torch.empty(n, dtype=torch.uint8, device=dev)
def rate(f, reps=10):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return n / 1e9 / ((time.perf_counter() - t0) / reps)
print("H2D 100 MB one copy   : %.1f GB/s" % rate(lambda: d.copy_(h, non_blocking=True)))
print("D2H 100 MB one copy   : %.1f GB/s" % rate(lambda: h.copy_(d, non_blocking=True)))
B = 8 << 20
def pieces_h2d():
    for o in range(0, n, B): d[o:o + B].copy_(h[o:o + B], non_blocking=True)
def pieces_d2h():
    for o in range(0, n, B): h[o:o + B].copy_(d[o:o + B], non_blocking=True)
print("H2D 8 MiB pieces      : %.1f GB/s" % rate(pieces_h2d))
print("D2H 8 MiB pieces      : %.1f GB/s" % rate(pieces_d2h))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h2 = torch.empty(n, dtype=torch.uint8, pin_memory=True); d2 = torch.empty(n, dtype=torch.uint8, device=dev)
def both():
    with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
print("H2D + D2H concurrently: %.1f GB/s each way" % rate(both))
