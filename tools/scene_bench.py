"""Developer benchmark for BASELINE.json configs[2]/[3] on the Sponza STAND-IN (bvh_amd.scene.make_atrium_obj):
10 M coherent primary rays and a 12.5 M-ray incoherent shard, CSR and fused-closest-hit modes, per traversal
variant.  Usage on the GPU box:  python tools/scene_bench.py [detail] [variants...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bvh_amd import Bvh, Context, RayBatch, scene  # noqa: E402
from bvh_amd._lib import RAY_F32  # noqa: E402
from bvh_amd.api import camera  # noqa: E402

detail = int(sys.argv[1]) if len(sys.argv) > 1 else 16
variants = [int(v) for v in sys.argv[2:]] or [0, 2]
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
tris, aabbs, bounds = scene.parse_obj(scene.make_atrium_obj(detail))
print(f"stand-in atrium detail {detail}: {len(tris)} triangles, bounds {bounds.tolist()}")
bvh = Bvh.from_aabbs(torch.from_numpy(aabbs).to(dev), ctx)
ctx.enable_timing(True)
bvh.rebuild(torch.from_numpy(aabbs).to(dev)); bvh.flatten_in_place()
print("build+flatten ms:", {k: round(v, 4) for k, v in ctx.last_timings().items() if k in ("build_ms", "flatten_ms")})
bvh.set_triangles(tris)
c = (bounds[:3] + bounds[3:]) * 0.5
cam = camera(c, c + np.array([1.0, -0.15, 0.25]), fov_y_deg=70.0, aspect=4000 / 2500)
W, H = 4000, 2500
buf = torch.empty(12_500_000 * RAY_F32.itemsize, dtype=torch.uint8, device=dev)


def timed(fn, reps=5):
    ts = []
    out = None
    for _ in range(reps):
        out = fn()
        ts.append(ctx.last_timings()["traverse_kernel_ms"])
    return out, float(np.median(ts))


ref = {}
for name, mk in (("primary 10M", lambda: RayBatch.primary(cam, W, H, 0, W * H, buf, np.float32, ctx)),
                 ("incoherent 12.5M", lambda: RayBatch.generate(62_500_000, 12_500_000, bounds, buf, np.float32, ctx))):
    rays = mk()
    for v in variants:
        ctx.set_tuning(0, v)
        (cl, prim, st), ms = timed(lambda: bvh.closest_hits(rays, stats=False))
        _, _, st = bvh.closest_hits(rays, stats=True)
        key = (name, "closest")
        sig = (cl.tobytes(), prim.tobytes())
        same = ref.setdefault(key, sig) == sig
        print(f"{name:17s} variant {v} closest: {ms:8.3f} ms  {rays.n / ms / 1e3:8.1f} Mrays/s  visited/ray {st['visited'] / rays.n:6.1f} "
              f"cands/ray {st['hits'] / rays.n:5.2f} util {st['device_steps'] / 64 / max(st['wave_steps'], 1):.3f} same={same}", flush=True)
        (_, _, _, st2), ms2 = timed(lambda: bvh.traverse_batch(rays, fetch=False), reps=3)
        print(f"{name:17s} variant {v} CSR    : {ms2:8.3f} ms  {rays.n / ms2 / 1e3:8.1f} Mrays/s  hits {st2['hits']}", flush=True)
