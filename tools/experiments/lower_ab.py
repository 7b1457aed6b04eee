"""A/B of BVHGPU_TUNE_BUILD_LOWER_FUSED (k_lower: workgroup tier + wave tier in one launch) on create_n_cubes scenes: build ms per setting,
byte equality of the BvhNode array across settings and against the oracle.   python tools/lower_ab.py [cubes ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bvh_amd import Bvh, Context, testbase as tb  # noqa: E402
from bvh_amd._lib import TUNE_BUILD_LOWER_FUSED  # noqa: E402
from oracle import orc  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [10_000, 2_000, 20_000]
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
ctx.enable_timing(True)
for cubes in sizes:
    _, aabbs = tb.create_n_cubes(cubes)
    a_dev = torch.from_numpy(aabbs).to(dev)
    ref = orc.build(aabbs, threads=min(16, orc.max_threads())).nodes.tobytes()
    for knob in (0, 1, 0, 1):
        ctx.set_tuning(TUNE_BUILD_LOWER_FUSED, knob)
        bvh = Bvh.from_aabbs(a_dev, ctx)
        bt, ft = [], []
        for rep in range(30):
            bvh.rebuild(a_dev); bvh.flatten_in_place()
            t = ctx.last_timings(); bt.append(t["build_ms"]); ft.append(t["flatten_ms"])
        same = bvh.nodes.tobytes() == ref
        print(f"{12 * cubes:8d} triangles  fused {knob}: build {np.median(bt):.4f} ms (min {min(bt):.4f})  flatten {np.median(ft):.4f}  levels {bvh.build_levels}  nodes == oracle: {same}", flush=True)
        bvh.close()
