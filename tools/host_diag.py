"""developer tool: where the host's time goes in one step of BASELINE configs[1] (async build + traverse + one wait)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bvh_amd import Bvh, Context, RayBatch, testbase as tb  # noqa: E402
from bvh_amd._lib import RAY_F32  # noqa: E402

R = 1_000_000
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
bounds = tb.default_bounds()
_, aabbs = tb.create_n_cubes(10_000, bounds)
a = torch.from_numpy(aabbs).to(dev)
buf = torch.empty(R * RAY_F32.itemsize, dtype=torch.uint8, device=dev)
rays = RayBatch.generate(0, R, bounds, buf, np.float32, ctx)
bvh = Bvh.from_aabbs(a, ctx)
bvh.flatten_in_place()
hits = None
rows = []
for i in range(300):
    t0 = time.perf_counter()
    bvh.rebuild_async(a)
    t1 = time.perf_counter()
    hits = bvh.traverse_async(rays)
    t2 = time.perf_counter()
    hits.wait()
    t3 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
r = np.array(rows[50:]) * 1e6
print("us per step (median): enqueue build %.1f  enqueue traverse %.1f  wait %.1f  total %.1f" % tuple(np.median(r, axis=0)))
# the same with the GPU idle at every enqueue (sync first): pure host cost of the enqueues
rows = []
for i in range(100):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bvh.rebuild_async(a)
    t1 = time.perf_counter()
    hits = bvh.traverse_async(rays)
    t2 = time.perf_counter()
    hits.wait()
    t3 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
r = np.array(rows[20:]) * 1e6
print("after an explicit sync     : enqueue build %.1f  enqueue traverse %.1f  wait %.1f  total %.1f" % tuple(np.median(r, axis=0)))
