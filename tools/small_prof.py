"""developer tool: per-wave timeline of k_small (the builder's wave tier).  Build the instrumented library first:
   python bvh_amd/build_ext.py --variant /root/repo/tools/libbvh_smallprof.so BVH_SMALL_PROFILE
   then on the GPU box: python tools/small_prof.py [n_cubes]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BVH_AMD_SO"] = os.path.join(ROOT, "tools", "libbvh_smallprof.so")
import numpy as np  # noqa: E402

from bvh_amd import Bvh, _lib, testbase as tb  # noqa: E402

cubes = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
lib = _lib.load()
_, aabbs = tb.create_n_cubes(cubes)
bvh = Bvh.from_aabbs(aabbs)
for _ in range(5):
    bvh.rebuild(aabbs)
W = 8192
out = (C.c_ulonglong * (20 * W))()
lib.bvhgpu_debug_small_prof(out, 20 * W)
a = np.array(out[:], dtype=np.float64).reshape(-1, 20)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
us = lambda x: (x - t0) / 100.0  # noqa: E731  100 MHz wall clock
start, loaded, end, shapes = us(a[:, 0]), us(a[:, 1]), us(a[:, 18]), a[:, 19]
lv = a[:, 2:18]
nlev = (lv > 0).sum(axis=1)
print(f"waves with a first item {len(a)}  shapes per item: mean {shapes.mean():.1f}  min {shapes.min():.0f}  max {shapes.max():.0f}")
print(f"entry           : mean {start.mean():6.2f}  max {start.max():6.2f} us")
print(f"shapes loaded   : mean {loaded.mean():6.2f}  max {loaded.max():6.2f} us   (entry -> loaded: mean {(loaded - start).mean():5.2f})")
print(f"loop iterations : mean {nlev.mean():5.2f}  p90 {np.percentile(nlev, 90):.0f}  max {nlev.max():.0f}")
prev = a[:, 1]
for L in range(16):
    m = lv[:, L] > 0
    if m.sum() == 0:
        break
    d = (lv[m, L] - prev[m]) / 100.0
    print(f"  iteration {L:2d}: {m.sum():5d} waves, {d.mean():5.2f} us mean, {np.percentile(d, 90):5.2f} p90")
    prev = np.where(m, lv[:, L], prev)
print(f"exit            : mean {end.mean():6.2f}  p50 {np.percentile(end, 50):6.2f}  p90 {np.percentile(end, 90):6.2f}  max {end.max():6.2f} us")
