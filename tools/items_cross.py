"""developer: where 16 items per ray stop paying against whole rays — wide-walk kernel and total time over batch sizes on the 120 k-cube scene
(few hits) and on the stand-in scene (hit-heavy): python tools/items_cross.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from bvh_amd import Bvh, Context, RayBatch, scene, testbase as tb
from bvh_amd._lib import RAY_F32, TUNE_WIDE_ITEMS_LOG4
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
ctx.enable_timing(True)
def run(label, bvh, rays):
    for items in (0, 2):
        ctx.set_tuning(TUNE_WIDE_ITEMS_LOG4, items)
        ts = []; tt = []
        for _ in range(6):
            st = bvh.traverse_batch(rays, fetch=False)[3]; ts.append(ctx.last_timings()["traverse_kernel_ms"]); tt.append(ctx.last_timings()["traverse_total_ms"])
        print(f"{label:28s} items 4^{items}: kernel {np.median(ts[2:]):8.4f} ms total {np.median(tt[2:]):8.4f}  hits {st['hits']}", flush=True)
b2 = tb.default_bounds(); _, a2 = tb.create_n_cubes(10_000, b2)
bvh2 = Bvh.from_aabbs(torch.from_numpy(a2).to(dev), ctx); bvh2.flatten_in_place()
buf = torch.empty(8_000_000 * RAY_F32.itemsize, dtype=torch.uint8, device=dev)
for n in (500_000, 1_000_000, 2_000_000, 3_000_000, 4_000_000, 8_000_000):
    run(f"cubes120k {n/1e6:.1f}M rays", bvh2, RayBatch.generate(0, n, b2, buf, np.float32, ctx))
_, aabbs_np, bounds = scene.parse_obj(scene.make_atrium_obj(16))
bvh = Bvh.from_aabbs(torch.from_numpy(aabbs_np).to(dev), ctx); bvh.flatten_in_place()
for n in (500_000, 1_000_000, 2_000_000, 4_000_000):
    run(f"atrium incoherent {n/1e6:.1f}M", bvh, RayBatch.generate(62_500_000, n, bounds, buf, np.float32, ctx))
