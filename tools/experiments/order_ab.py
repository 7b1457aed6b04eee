"""A/B of BVHGPU_TUNE_WIDE_ORDER_RAYS (whole-ray batches walked in the order of a counting sort by origin cell + direction octant, one
eighth of the order per XCD) on the stand-in scene's incoherent stream and on the 120 k-cube scene.  Prints walk / total ms per setting and
checks that the CSR bytes do not depend on it.   python tools/order_ab.py [rays_millions ...]"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bvh_amd import Bvh, Context, RayBatch, scene, testbase as tb  # noqa: E402
from bvh_amd._lib import RAY_F32, TUNE_WIDE_ORDER_RAYS, WALK_ORDERED  # noqa: E402

sizes = [int(float(a) * 1e6) for a in sys.argv[1:]] or [12_500_000]
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
ctx.enable_timing(True)
_, aabbs_s, bounds_s = scene.parse_obj(scene.make_atrium_obj(16))
_, aabbs_c = tb.create_n_cubes(10_000)
for scene_name, aabbs, bounds in (("standin", aabbs_s, bounds_s), ("cubes120k", aabbs_c, tb.default_bounds())):
    bvh = Bvh.from_aabbs(torch.from_numpy(aabbs).to(dev), ctx)
    bvh.flatten_in_place()
    for n in sizes:
        buf = torch.empty(n * RAY_F32.itemsize, dtype=torch.uint8, device=dev)
        rays = RayBatch.generate(62_500_000 if n <= 12_500_000 else 0, n, bounds, buf, np.float32, ctx)
        ref = None
        for knob in (0, 1, 2, 3, 0, 1):
            ctx.set_tuning(TUNE_WIDE_ORDER_RAYS, knob)
            ks, ts = [], []
            for rep in range(4):
                st = bvh.traverse_batch(rays, fetch=False)[3]
                t = ctx.last_timings()
                ks.append(t["traverse_kernel_ms"]); ts.append(t["traverse_total_ms"])
            off, idx, _, st = bvh.traverse_batch(rays)
            sig = hashlib.sha256(off.tobytes() + idx.tobytes()).hexdigest()[:16]
            ref = ref or sig
            print(f"{scene_name:9s} {n / 1e6:6.1f} M rays  order {knob}: walk(+order) {np.median(ks):8.3f} ms  total {np.median(ts):8.3f} ms  hits {st['hits']}  "
                  f"ordered={bool(st['walk'] & WALK_ORDERED)}  csr {sig} same={sig == ref}", flush=True)
        del buf, rays
        torch.cuda.empty_cache()
    bvh.close()
