"""developer: HIP-event time of the default wide walk on configs[1] for the library in BVH_AMD_SO (median of 9)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bvh_amd import Bvh, Context, RayBatch, testbase as tb
from bvh_amd._lib import RAY_F32
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
bounds = tb.default_bounds()
_, aabbs = tb.create_n_cubes(10_000, bounds)
bvh = Bvh.from_aabbs(torch.from_numpy(aabbs).to(dev), ctx); bvh.flatten_in_place()
buf = torch.empty(R * RAY_F32.itemsize, dtype=torch.uint8, device=dev)
rays = RayBatch.generate(0, R, bounds, buf, np.float32, ctx)
ctx.enable_timing(True)
ts = []
for _ in range(12):
    st = bvh.traverse_batch(rays, fetch=False)[3]; ts.append(ctx.last_timings()["traverse_kernel_ms"])
print(os.environ.get("BVH_AMD_SO", "default"), R, "kernel ms median", round(float(np.median(ts[3:])), 4), "hits", st["hits"])
