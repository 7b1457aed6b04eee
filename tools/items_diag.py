"""developer: wide-walk kernel time vs items-per-ray on the large-batch configs: python tools/items_diag.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bvh_amd import Bvh, Context, RayBatch, scene, testbase as tb
from bvh_amd._lib import RAY_F32, TUNE_WIDE_ITEMS_LOG4, TUNE_WIDE_WG_PER_CU
from bvh_amd.api import camera
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
ctx.enable_timing(True)
def run(label, bvh, rays):
    for items in (0, 1, 2):
        for wg in (2, 1):
            ctx.set_tuning(TUNE_WIDE_ITEMS_LOG4, items); ctx.set_tuning(TUNE_WIDE_WG_PER_CU, wg)
            ts = []
            for _ in range(6):
                st = bvh.traverse_batch(rays, fetch=False)[3]; ts.append(ctx.last_timings()["traverse_kernel_ms"])
            print(f"{label:28s} items 4^{items} wg/cu {wg}: kernel {np.median(ts[2:]):8.4f} ms  hits {st['hits']}", flush=True)
_, aabbs_np, bounds = scene.parse_obj(scene.make_atrium_obj(16))
bvh = Bvh.from_aabbs(torch.from_numpy(aabbs_np).to(dev), ctx); bvh.flatten_in_place()
R = 12_500_000
buf = torch.empty(R * RAY_F32.itemsize, dtype=torch.uint8, device=dev)
c = (bounds[:3] + bounds[3:]) * 0.5
cam = camera(c, c + np.array([1.0, -0.15, 0.25]), fov_y_deg=70.0, aspect=4000 / 2500)
run("atrium 10M primary", bvh, RayBatch.primary(cam, 4000, 2500, 0, 10_000_000, buf, np.float32, ctx))
run("atrium 12.5M incoherent", bvh, RayBatch.generate(62_500_000, R, bounds, buf, np.float32, ctx))
b2 = tb.default_bounds(); _, a2 = tb.create_n_cubes(10_000, b2)
bvh2 = Bvh.from_aabbs(torch.from_numpy(a2).to(dev), ctx); bvh2.flatten_in_place()
for n in (2_000_000, 4_000_000, 8_000_000):
    run(f"cubes120k {n//1000000}M rays", bvh2, RayBatch.generate(0, n, b2, buf, np.float32, ctx))
