"""developer tool: 60 rebuilds of BASELINE configs[1]'s scene (run under rocprofv3 --kernel-trace for per-kernel times)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bvh_amd import Bvh, Context, testbase as tb  # noqa: E402

dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
_, aabbs = tb.create_n_cubes(10_000, tb.default_bounds())
a = torch.from_numpy(aabbs).to(dev)
bvh = Bvh.from_aabbs(a, ctx)
ctx.enable_timing(True)
ts = []
for _ in range(60):
    try:
        bvh.rebuild(a)
        ts.append(ctx.last_timings()["build_ms"])
    except Exception as e:  # noqa: BLE001  (timing-only variants may build nothing)
        ts.append(float("nan")); err = e
print("build_ms median", np.nanmedian(ts), "min", np.nanmin(ts))
