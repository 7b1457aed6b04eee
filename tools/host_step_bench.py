"""round 6: the host-resident step (pinned AABBs + origins + directions in, CSR out) by chunk count, beside the pageable synchronous path"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bvh_amd import Bvh, Context, HostStep, RayBatch, testbase as tb
from bvh_amd._lib import TUNE_HOST_CHUNKS
from oracle import orc
dev = torch.device("cuda", 0)
ctx = Context(0)
_, aabbs = tb.create_n_cubes(10000)
R = 1_000_000
k = np.arange(0, R, dtype=np.uint64); b = tb.default_bounds()
o = tb.next_point3_at(2 * k + 1, b); d = tb.next_point3_at(2 * k + 2, b)
rays = orc.create_rays(0, R)
bvh = Bvh.from_aabbs(aabbs, ctx); bvh.flatten_in_place()
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
rb = RayBatch(R, np.float32, host=rays)
ms = t(lambda: (bvh.rebuild(aabbs, flatten=True), bvh.traverse_batch(rb, fetch=True)))
print(f"pageable, synchronous entry points (36 B/ray): {ms:.4f} ms  {R / ms / 1e3:.0f} Mrays/s", flush=True)
hs = HostStep(bvh, len(aabbs), R, np.float32)
hs.aabbs[:] = aabbs; hs.origins[:] = o; hs.directions[:] = d
def med(fn, blocks=7, n=20):
    v = sorted(t(fn, n) for _ in range(blocks))
    return v[len(v) // 2], v[0], v[-1]
from bvh_amd._lib import TUNE_HOST_ZERO_COPY
for zc in (2,):
  ctx.set_tuning(TUNE_HOST_ZERO_COPY, zc)
  for fused in (True, False):
    for ch in (0, 3):
        ctx.set_tuning(TUNE_HOST_CHUNKS, ch)
        ms, lo, hi = med(lambda: hs.run(fused=fused))
        print(f"pinned, zero_copy={zc}, {'bvhgpu_build_traverse_host' if fused else 'rebuild_flat_async + traverse_host'} (24 B/ray), chunks={ch}: median {ms:.4f} ms "
              f"[{lo:.4f} .. {hi:.4f}]  {R / ms / 1e3:.0f} Mrays/s  total={hs.total}", flush=True)
ctx.set_tuning(TUNE_HOST_ZERO_COPY, 2)
hs6 = HostStep(bvh, len(aabbs), R, np.float32, od6=True)
hs6.aabbs[:] = aabbs; hs6.origins[:] = o; hs6.directions[:] = d
for zc in (2, 0):
  ctx.set_tuning(TUNE_HOST_ZERO_COPY, zc)
  for fused in (True, False):
    for ch in (0, 2, 3, 4):
        ctx.set_tuning(TUNE_HOST_CHUNKS, ch)
        ms, lo, hi = med(lambda: hs6.run(fused=fused))
        print(f"pinned OD6, zero_copy={zc}, {'bvhgpu_build_traverse_host' if fused else 'rebuild_flat_async + traverse_host'} (24 B/ray, one array), chunks={ch}: median {ms:.4f} ms "
              f"[{lo:.4f} .. {hi:.4f}]  {R / ms / 1e3:.0f} Mrays/s  total={hs6.total}", flush=True)
ctx.set_tuning(TUNE_HOST_ZERO_COPY, 2); ctx.set_tuning(TUNE_HOST_CHUNKS, 0)
ctx.set_tuning(TUNE_HOST_CHUNKS, 0)
# pageable origins / directions through the same entry point
off = np.zeros(R + 1, np.uint32); idx = np.zeros(1 << 20, np.uint32)
ms = t(lambda: (bvh.rebuild(aabbs, flatten=True), bvh.traverse_host(o, d, off, idx)))
print(f"pageable, bvhgpu_traverse_host (24 B/ray): {ms:.4f} ms  {R / ms / 1e3:.0f} Mrays/s", flush=True)
# Ray structs, pinned
from bvh_amd.api import pinned_array
from bvh_amd._lib import RAY_F32
pr = pinned_array(ctx, (R,), RAY_F32); pr[:] = rays
def step_structs():
    bvh.rebuild_async(hs.aabbs)
    bvh.traverse_host(pr, None, hs.offsets, hs.indices)
ms = t(step_structs)
print(f"pinned, bvhgpu_traverse_host on Ray structs (36 B/ray): {ms:.4f} ms  {R / ms / 1e3:.0f} Mrays/s", flush=True)
