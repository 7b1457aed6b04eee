"""Developer diagnostic: a few traversals of configs[1] (for rocprofv3 --pmc passes).  argv: variant wpc refill reps"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bvh_amd import Bvh, Context, RayBatch, testbase as tb  # noqa: E402
from bvh_amd._lib import RAY_F32  # noqa: E402

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 1
wpc = int(sys.argv[2]) if len(sys.argv) > 2 else 32
refill = int(sys.argv[3]) if len(sys.argv) > 3 else 1
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
R = 1_000_000
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
ctx.set_tuning(0, variant); ctx.set_tuning(1, wpc); ctx.set_tuning(2, refill)
bounds = tb.default_bounds()
_, aabbs = tb.create_n_cubes(10000, bounds)
bvh = Bvh.from_aabbs(torch.from_numpy(aabbs).to(dev), ctx)
bvh.flatten_in_place()
buf = torch.empty(R * RAY_F32.itemsize, dtype=torch.uint8, device=dev)
rays = RayBatch.generate(0, R, bounds, buf, np.float32, ctx)
for _ in range(reps):
    bvh.traverse_batch(rays, fetch=False)
torch.cuda.synchronize()
print("done")
