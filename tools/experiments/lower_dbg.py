import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from bvh_amd import Bvh, Context, testbase as tb
from bvh_amd._lib import TUNE_BUILD_LOWER_FUSED
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
ctx.enable_timing(True)
for cubes in (2000, 10000):
    _, aabbs = tb.create_n_cubes(cubes)
    a_dev = torch.from_numpy(aabbs).to(dev)
    for knob in (0, 1, 256, 300, 512, 768):
        ctx.set_tuning(TUNE_BUILD_LOWER_FUSED, knob)
        bvh = Bvh.from_aabbs(a_dev, ctx)
        bt = []
        for rep in range(10):
            bvh.rebuild(a_dev); bvh.flatten_in_place(); bt.append(ctx.last_timings()["build_ms"])
        print(cubes * 12, "knob", knob, "build ms", round(float(np.median(bt)), 4), flush=True)
        bvh.close()
