"""round 6: the wide walk by batch size, one ray per lane against the rays spread over all workgroup slots (BVHGPU_TUNE_WIDE_MIN_RAYS_PER_WG)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bvh_amd import Bvh, Context, RayBatch, testbase as tb
from bvh_amd._lib import TUNE_WIDE_MIN_RAYS_PER_WG
dev = torch.device("cuda", 0)
ctx = Context(0)
_, aabbs = tb.create_n_cubes(10000)
bvh = Bvh.from_aabbs(aabbs, ctx); bvh.flatten_in_place()
buf = torch.empty(2_000_000 * 36, dtype=torch.uint8, device=dev)
ctx.enable_timing(True)
for R in (32768, 65536, 125_000, 250_000, 437_500, 600_000, 1_000_000):
    rays = RayBatch.generate(0, R, tb.default_bounds(), buf, np.float32, ctx)
    row = []
    for knob in (0, 64, 128, 256, 512):
        ctx.set_tuning(TUNE_WIDE_MIN_RAYS_PER_WG, knob)
        ts = []
        for _ in range(30):
            h = bvh.traverse_async(rays); h.wait()
            ts.append(ctx.last_timings()["traverse_kernel_ms"])
        row.append(f"{knob}: {np.median(ts) * 1e3:6.1f}")
    print(f"R = {R:8d}  walk kernel us by min rays per workgroup  " + "   ".join(row), flush=True)
