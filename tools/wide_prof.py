"""developer tool: per-wave phase times inside k_traverse_wide.  Build the instrumented library first:
   python bvh_amd/build_ext.py --variant tools/libbvh_wideprof.so BVH_WIDE_PROFILE
   then on the GPU box: python tools/wide_prof.py [rays] [items_log4] [cubes | primary | incoherent]   (the last two: the stand-in scene,
   10 M pinhole rays / a 12.5 M-ray shard of the incoherent stream — give rays = 0 for their own counts)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BVH_AMD_SO"] = os.path.join(ROOT, "tools", "libbvh_wideprof.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bvh_amd import Bvh, Context, RayBatch, _lib, scene, testbase as tb  # noqa: E402
from bvh_amd.api import camera  # noqa: E402
from bvh_amd._lib import RAY_F32, TUNE_WIDE_ITEMS_LOG4  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
items = int(sys.argv[2]) if len(sys.argv) > 2 else -1
lib = _lib.load()
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
ctx.set_tuning(TUNE_WIDE_ITEMS_LOG4, items)
what = sys.argv[3] if len(sys.argv) > 3 else "cubes"
if what == "cubes":
    bounds = tb.default_bounds()
    _, aabbs = tb.create_n_cubes(10_000, bounds)
else:
    _, aabbs, bounds = scene.parse_obj(scene.make_atrium_obj(16))
bvh = Bvh.from_aabbs(torch.from_numpy(aabbs).to(dev), ctx)
bvh.flatten_in_place()
if what == "primary":
    W, H = 4000, 2500
    R = R or W * H
    buf = torch.empty(W * H * RAY_F32.itemsize, dtype=torch.uint8, device=dev)
    c = (bounds[:3] + bounds[3:]) * 0.5
    rays = RayBatch.primary(camera(c, c + np.array([1.0, -0.15, 0.25]), fov_y_deg=70.0, aspect=W / H), W, H, 0, R, buf, np.float32, ctx)
else:
    R = R or 12_500_000
    buf = torch.empty(R * RAY_F32.itemsize, dtype=torch.uint8, device=dev)
    rays = RayBatch.generate(62_500_000 if what == "incoherent" else 0, R, bounds, buf, np.float32, ctx)
for _ in range(3):
    bvh.traverse_batch(rays, fetch=False, coherent=(what == "primary"))
n_waves = 8192
out = (C.c_ulonglong * (4 * n_waves))()
lib.bvhgpu_debug_wide_prof(out, 4 * n_waves)
a = np.array(out[:], dtype=np.float64).reshape(-1, 4)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
us = lambda x: (x - t0) / 100.0  # noqa: E731  100 MHz wall clock
start, pro, end, steps = us(a[:, 0]), us(a[:, 1]), us(a[:, 2]), a[:, 3]
print(f"waves {len(a)}  rays {R}  items_log4 {items}")
print(f"wave start      : mean {start.mean():7.2f}  max {start.max():7.2f} us")
print(f"prologue end    : mean {pro.mean():7.2f}  max {pro.max():7.2f} us   (duration mean {(pro - start).mean():6.2f})")
print(f"wave end        : mean {end.mean():7.2f}  p50 {np.percentile(end, 50):7.2f}  p90 {np.percentile(end, 90):7.2f}  max {end.max():7.2f} us")
print(f"walk steps/wave : mean {steps.mean():6.1f}  max {steps.max():6.0f};  walk time per step (mean over waves) {((end - pro) / np.maximum(steps, 1)).mean():6.3f} us")
util = (C.c_ulonglong * (16 * n_waves))()
lib.bvhgpu_debug_wide_util(util, 16 * n_waves)
u = np.array(util[:], dtype=np.float64).reshape(-1, 16)
u = u[u[:, 0] > 0].sum(axis=0)
ws = u[0]
print(f"lane utilisation over {ws:.0f} wave-steps (last launch): lanes on an inner node {u[1] / ws:5.1f} of 64, lanes holding an item {u[8] / ws:5.1f}; "
      f"steps with a resident fetch {u[2] / ws:.2f}, with a non-resident fetch {u[3] / ws:.2f}, with a slow push {u[4] / ws:.2f} ({u[5] / ws:.1f} lanes), "
      f"with a leaf report {u[7] / ws:.2f} ({u[6] / ws:.2f} lanes), with a pop from the HBM stack part {u[11] / ws:.3f}; "
      f"rounds on the exact path {u[10] / max(u[9], 1):.3f}; boxes hit per tested node {u[12] / max(u[1], 1):.2f}, nodes with no hit {u[13] / max(u[1], 1):.2f}")
wg_end = end.reshape(-1, 16).max(axis=1) if len(end) % 16 == 0 else end
wg_pro = pro.reshape(-1, 16).max(axis=1) if len(pro) % 16 == 0 else pro
print(f"per subtree (wg % 16) end: " + " ".join(f"{wg_end[j::16].mean():.0f}" for j in range(16)))
print(f"workgroup end   : mean {wg_end.mean():7.2f}  min {wg_end.min():7.2f}  max {wg_end.max():7.2f} us")

if len(wg_end) >= 512:
    print("workgroup end by workgroup range (mean / max): " + "  ".join(f"[{a}:{a + 64}) {wg_end[a:a + 64].mean():.0f}/{wg_end[a:a + 64].max():.0f}" for a in range(0, 512, 64)))
    wg_steps = steps.reshape(-1, 16).mean(axis=1)
    print("prologue end by range: " + "  ".join(f"[{a}:{a + 64}) {wg_pro[a:a + 64].mean():.1f}" for a in range(0, 512, 64)))
    print("mean wave-steps by range: " + "  ".join(f"[{a}:{a + 64}) {wg_steps[a:a + 64].mean():.1f}" for a in range(0, 512, 64)))
    order = np.argsort(wg_end)
    print("slowest workgroups:", order[-12:][::-1].tolist(), " fastest:", order[:8].tolist())
