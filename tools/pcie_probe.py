"""round 6: what the PCIe link of the box gives a host-resident caller — pinned and pageable, both directions, by transfer size;
and whether H2D and D2H on two streams overlap"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda", 0)
def bw(src, dst, n=20, stream=None):
    torch.cuda.synchronize()
    for _ in range(3): dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    return dt * 1e6, src.numel() / dt / 1e9
for mb in (0.04, 1, 3, 4, 6, 12, 24, 36):
    nb = int(mb * 1e6)
    hp = torch.empty(nb, dtype=torch.uint8).pin_memory(); hg = torch.empty(nb, dtype=torch.uint8); d = torch.empty(nb, dtype=torch.uint8, device=dev)
    r = [bw(hp, d), bw(d, hp), bw(hg, d), bw(d, hg)]
    print(f"{mb:6.2f} MB  pinned H2D {r[0][0]:7.1f} us {r[0][1]:5.1f} GB/s | pinned D2H {r[1][0]:7.1f} us {r[1][1]:5.1f} GB/s | "
          f"pageable H2D {r[2][0]:7.1f} us {r[2][1]:5.1f} GB/s | pageable D2H {r[3][0]:7.1f} us {r[3][1]:5.1f} GB/s", flush=True)
# both directions at once on two streams
nb = 24_000_000
hp = torch.empty(nb, dtype=torch.uint8).pin_memory(); d = torch.empty(nb, dtype=torch.uint8, device=dev)
hp2 = torch.empty(nb, dtype=torch.uint8).pin_memory(); d2 = torch.empty(nb, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    with torch.cuda.stream(s1): d.copy_(hp, non_blocking=True)
    with torch.cuda.stream(s2): hp2.copy_(d2, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"24 MB H2D + 24 MB D2H concurrently: {dt * 1e6:.1f} us -> {2 * nb / dt / 1e9:.1f} GB/s both ways", flush=True)
# two H2D streams at once (does a second SDMA engine help?)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    with torch.cuda.stream(s1): d.copy_(hp, non_blocking=True)
    with torch.cuda.stream(s2): d2.copy_(hp2, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"2 x 24 MB H2D on two streams: {dt * 1e6:.1f} us -> {2 * nb / dt / 1e9:.1f} GB/s", flush=True)
