"""round 6 diagnosis: the harness step (ray generation + rebuild + closest-hit walk over items) under BVHGPU_TUNE_FLATTEN_LAZY 0 / 1"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bvh_amd import Bvh, Context, RayBatch, testbase as tb
from bvh_amd._lib import TRAVERSE_CLOSEST, TRAVERSE_TRIANGLES, TRAVERSE_RAYS_READY, TUNE_FLATTEN_LAZY
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
tris, aabbs = tb.create_n_cubes(10000)
R = 1_000_000
d_a = torch.from_numpy(aabbs).to(dev)
d_t = torch.from_numpy(np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)).to(dev)
buf = torch.empty(R * 36, dtype=torch.uint8, device=dev)
bounds = tb.default_bounds()
rays = RayBatch.generate(0, R, bounds, buf, np.float32, ctx)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for lazy in (1, 0, 1, 0):
    ctx.set_tuning(TUNE_FLATTEN_LAZY, lazy)
    bvh = Bvh.from_aabbs(d_a, ctx); bvh.flatten_in_place(); bvh.set_triangles(d_t)
    for name, fl, gen in (("index", TRAVERSE_RAYS_READY, False), ("closest", TRAVERSE_CLOSEST, True), ("closest-nogen", TRAVERSE_CLOSEST, False),
                          ("triangles", TRAVERSE_TRIANGLES, True), ("closest again", TRAVERSE_CLOSEST, True)):
        def step():
            if gen: RayBatch.generate(0, R, bounds, buf, np.float32, ctx)
            bvh.rebuild_async(d_a)
            return bvh.traverse_async(rays, flags=fl).wait()
        ms = t(step)
        print(f"lazy={lazy} {name:14s} {ms:.4f} ms/step  {R / ms / 1e3:.0f} Mrays/s  levels={bvh.build_levels}", flush=True)
    bvh.close()
