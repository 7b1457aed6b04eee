"""developer check: build time of the two level-tier schedules (one launch per level / k_bin + k_split) over scene sizes"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bvh_amd import Bvh, Context, testbase as tb
from bvh_amd._lib import TUNE_BUILD_LEVEL_LAUNCHES

dev = torch.device("cuda", 0)
for cubes in [int(x) for x in sys.argv[1:]] or (10_000, 30_000, 100_000, 300_000, 1_000_000):
    _, a = tb.create_n_cubes(cubes, tb.default_bounds())
    aabbs = torch.from_numpy(a).to(dev)
    row = []
    for launches in (1, 2):
        ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
        ctx.set_tuning(TUNE_BUILD_LEVEL_LAUNCHES, launches)
        bvh = Bvh.from_aabbs(aabbs, ctx)
        ctx.enable_timing(True)
        bt = []
        for _ in range(7):
            bvh.rebuild(aabbs)
            bt.append(ctx.last_timings()["build_ms"])
        row.append(float(np.median(bt)))
    print(f"{12 * cubes:9d} triangles: one launch per level {row[0]:8.3f} ms   two launches {row[1]:8.3f} ms", flush=True)
