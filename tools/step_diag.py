"""developer: where does a step's wall time go on the large-batch configs?  python tools/step_diag.py [primary|incoherent]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bvh_amd import Bvh, Context, RayBatch, scene
from bvh_amd._lib import RAY_F32, TRAVERSE_COHERENT, TUNE_TRAVERSE_VARIANT
from bvh_amd.api import camera
which = sys.argv[1] if len(sys.argv) > 1 else "primary"
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
_, aabbs_np, bounds = scene.parse_obj(scene.make_atrium_obj(16))
aabbs = torch.from_numpy(aabbs_np).to(dev)
if which == "primary":
    R = 10_000_000
    c = (bounds[:3] + bounds[3:]) * 0.5
    cam = camera(c, c + np.array([1.0, -0.15, 0.25]), fov_y_deg=70.0, aspect=4000 / 2500)
    buf = torch.empty(R * RAY_F32.itemsize, dtype=torch.uint8, device=dev)
    rays = RayBatch.primary(cam, 4000, 2500, 0, R, buf, np.float32, ctx)
else:
    R = 12_500_000
    buf = torch.empty(R * RAY_F32.itemsize, dtype=torch.uint8, device=dev)
    rays = RayBatch.generate(62_500_000, R, bounds, buf, np.float32, ctx)
bvh = Bvh.from_aabbs(aabbs, ctx); bvh.flatten_in_place()
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
ctx.enable_timing(True)
for variant in (3, 2, 0):
    ctx.set_tuning(TUNE_TRAVERSE_VARIANT, variant)
    for coh in (False, True):
        st = bvh.traverse_batch(rays, fetch=False, coherent=coh)[3]
        ms = t(lambda: bvh.traverse_batch(rays, fetch=False, coherent=coh))
        tm = ctx.last_timings()
        print(f"{which} variant {variant} coherent {coh}: wall {ms:7.3f} ms  kernel {tm['traverse_kernel_ms']:7.3f}  total {tm['traverse_total_ms']:7.3f}  hits {st['hits']}", flush=True)
ctx.set_tuning(TUNE_TRAVERSE_VARIANT, 3)
print("sync  rebuild(flatten)+traverse wall", round(t(lambda: (bvh.rebuild(aabbs, flatten=True), bvh.traverse_batch(rays, fetch=False))), 3))
print("async rebuild+traverse+wait      wall", round(t(lambda: (bvh.rebuild_async(aabbs), bvh.traverse_async(rays).wait())), 3))
print("async (coherent flag)            wall", round(t(lambda: (bvh.rebuild_async(aabbs), bvh.traverse_async(rays, flags=TRAVERSE_COHERENT).wait())), 3))
print("rebuild only wall", round(t(lambda: bvh.rebuild(aabbs, flatten=True)), 3))
