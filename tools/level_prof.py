"""developer tool: phase stamps inside k_level at one level.  Build first:
   python bvh_amd/build_ext.py --variant /root/repo/tools/libbvh_levelprof.so BVH_LEVEL_PROFILE=<level>
   then on the GPU box: python tools/level_prof.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BVH_AMD_SO"] = os.path.join(ROOT, "tools", "libbvh_levelprof.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bvh_amd import Bvh, Context, _lib, testbase as tb  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
for k, v in os.environ.items():   # BVH_TUNE_<knob>=<value>, like bench.py (e.g. BVH_TUNE_16=64: the persistent level tier)
    if k.startswith("BVH_TUNE_"):
        ctx.set_tuning(int(k[9:]), int(v))
_, aabbs = tb.create_n_cubes(10_000, tb.default_bounds())
a = torch.from_numpy(aabbs).to(dev)
bvh = Bvh.from_aabbs(a, ctx)
for _ in range(5):
    bvh.rebuild(a)
n = 2 * 8 * 1024
out = (C.c_ulonglong * n)()
lib.bvhgpu_debug_level_prof.argtypes = [C.c_void_p, C.c_size_t]
lib.bvhgpu_debug_level_prof(out, n)
both = np.array(out[:], dtype=np.float64).reshape(2, -1, 8)
t0 = both[0][both[0][:, 0] > 0][:, 0].min()
us = lambda x: (x - t0) / 100.0  # noqa: E731
names = ["pass entry", "tile map read", "wave 0 at the barrier (ranks done)", "selection done", "shapes done", "flush done", "end", "statistics merged (selection starts)"]
for which, p in enumerate(both):      # the profiled level and the one after it, on ONE clock: the distance between "end" of the first and
    live = p[p[:, 1] > 0]             # "pass entry" of the second is the launch boundary (or, in the persistent tier, the group barrier)
    idle = p[(p[:, 0] > 0) & (p[:, 1] == 0)]
    if not len(live):
        continue
    print(f"--- level {'P' if which == 0 else 'P + 1'}: workgroups with a tile {len(live)}, without {len(idle)}")
    for i, nme in enumerate(names):
        col = us(live[:, i])
        print(f"{nme:58s} mean {col.mean():7.2f}  min {col.min():7.2f}  p50 {np.percentile(col, 50):7.2f}  p90 {np.percentile(col, 90):7.2f}  p99 {np.percentile(col, 99):7.2f}  max {col.max():7.2f} us")
    # where the late workgroups lose their time: phase durations of the ten that end last against the median workgroup
    order = [0, 1, 7, 3, 2, 4, 5, 6]          # stamps in the order they are taken
    dur = np.diff(us(live[:, order]), axis=1)
    med = np.median(dur, axis=0)
    slow = np.argsort(live[:, 6])[-10:][::-1]
    print("phase durations (entry→tile record→statistics→selection→ranks→shapes→flush→end), median workgroup: " + " ".join(f"{x:5.2f}" for x in med))
    for k in slow:
        wg = int(np.nonzero((p[:, 6] == live[k, 6]) & (p[:, 0] == live[k, 0]))[0][0])
        print(f"   workgroup {wg:4d} ends {us(live[k, 6]):6.2f}: " + " ".join(f"{x:5.2f}" for x in dur[k]))
