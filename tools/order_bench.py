#!/usr/bin/env python3
"""Walk-kernel time of every traversal order on BASELINE configs[1] (120 000 triangles, 1 M rays): flat-array order
(k_traverse_lds / k_traverse) vs the BvhNode-array walks (child-ordered iterator with a 32-entry LDS stack, heap-driven
best-first).  HIP-event times of the walk kernel only."""
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
_, aabbs = tb.create_n_cubes(10000, bounds)
bvh = Bvh.from_aabbs(torch.from_numpy(aabbs).to(dev), ctx); bvh.flatten_in_place()
buf = torch.empty(R * RAY_F32.itemsize, dtype=torch.uint8, device=dev)
rays = RayBatch.generate(0, R, bounds, buf, np.float32, ctx)
ctx.enable_timing(True)
for name, kw in (("flat (LDS top, persistent)", {}), ("flat (one ray per lane)", {"coherent": True}), ("nearest child", {"order": "nearest"}),
                 ("farthest child", {"order": "farthest"}), ("nearest heap", {"order": "nearest_heap"})):
    ts = []
    for _ in range(6):
        bvh.traverse_batch(rays, fetch=False, **kw)
        ts.append(ctx.last_timings()["traverse_kernel_ms"])
    print(f"{name:28s} walk {np.median(ts):.4f} ms  ({R / np.median(ts) / 1e3:.0f} Mrays/s)")
