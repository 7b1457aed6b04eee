#!/usr/bin/env python3
"""f64 (configs[4]) traversal: LDS slots / workgroup size sweep of k_traverse_lds (56 B per slot, so the f32 default of 2048
slots leaves room for only one 1024-thread workgroup per CU)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bvh_amd import Bvh, Context, RayBatch, testbase as tb
from bvh_amd._lib import RAY_F64

R = 1_000_000
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
bounds = tb.default_bounds()
_, aabbs = tb.create_n_cubes(10000, bounds)
bvh = Bvh.from_aabbs(torch.from_numpy(aabbs.astype(np.float64)).to(dev), ctx); bvh.flatten_in_place()
buf = torch.empty(R * RAY_F64.itemsize, dtype=torch.uint8, device=dev)
rays = RayBatch.generate(0, R, bounds, buf, np.float64, ctx)
ctx.enable_timing(True)
ref = None
for threads in (1024, 512, 256):
    for slots in (256, 512, 1024, 1400, 2048, 2880):
        ctx.set_tuning(4, slots); ctx.set_tuning(5, threads)
        ts = []
        for _ in range(6):
            st = bvh.traverse_batch(rays, fetch=False)[3]
            ts.append(ctx.last_timings()["traverse_kernel_ms"])
        if ref is None: ref = st["hits"]
        assert st["hits"] == ref
        print(f"threads {threads:5d} slots {slots:5d}: walk {np.median(ts):.4f} ms")
