#!/usr/bin/env python3
"""Scale sweep on one MI355X: create_n_cubes(c) scenes from 12 k to 12 M triangles x 1 M / 10 M (/ 100 M) create_ray rays.
HIP-event times of build, flatten and traversal (CSR in HBM); two walk kernels must agree on the hit count."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bvh_amd import Bvh, Context, RayBatch, testbase as tb
from bvh_amd._lib import RAY_F32

dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
bounds = tb.default_bounds()
big = "--big" in sys.argv
print("| triangles | build ms | flatten ms | rays | traverse ms | Mrays/s (traverse) | hits | build levels |")
print("|---:|---:|---:|---:|---:|---:|---:|---:|")
for cubes in (1_000, 10_000, 100_000, 1_000_000):
    t0 = time.time()
    _, a = tb.create_n_cubes(cubes, bounds)
    aabbs = torch.from_numpy(a).to(dev)
    bvh = Bvh.from_aabbs(aabbs, ctx); bvh.flatten_in_place()
    ctx.enable_timing(True)
    bt, ft = [], []
    for _ in range(5):
        bvh.rebuild(aabbs); bvh.flatten_in_place()
        t = ctx.last_timings(); bt.append(t["build_ms"]); ft.append(t["flatten_ms"])
    for R in (1_000_000, 10_000_000) + ((100_000_000,) if big and cubes == 10_000 else ()):
        buf = torch.empty(R * RAY_F32.itemsize, dtype=torch.uint8, device=dev)
        rays = RayBatch.generate(0, R, bounds, buf, np.float32, ctx)
        tt = []
        for _ in range(3):
            st = bvh.traverse_batch(rays, fetch=False, stats=False)[3]
            tt.append(ctx.last_timings()["traverse_total_ms"])
        h1 = st["hits"]
        h0 = bvh.traverse_batch(rays, fetch=False, coherent=True)[3]["hits"]   # the one-ray-per-lane kernel
        assert h0 == h1, (h0, h1)
        print(f"| {len(a)} | {np.median(bt):.3f} | {np.median(ft):.3f} | {R} | {np.median(tt):.3f} | {R / np.median(tt) / 1e3:.0f} | {h1} | {bvh.build_levels} |", flush=True)
        del rays, buf
    ctx.enable_timing(False)
    bvh.close(); del aabbs
    torch.cuda.empty_cache()
