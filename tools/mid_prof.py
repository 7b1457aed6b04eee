"""developer tool: per-phase cycle counts inside k_mid (python bvh_amd/build_ext.py --variant /root/repo/tools/libbvh_midprof.so BVH_PROFILE_MID=1)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "tools", "libbvh_midprof.so") if len(sys.argv) < 2 else sys.argv[1]
os.environ["BVH_AMD_SO"] = so
import numpy as np
from bvh_amd import Bvh, testbase as tb, _lib
lib = _lib.load()
_, aabbs = tb.create_n_cubes(10000)
bvh = Bvh.from_aabbs(aabbs)
out = (C.c_ulonglong * 8)()
lib.bvhgpu_debug_mid_prof(out, 1)
for _ in range(5):
    bvh.rebuild(aabbs)
lib.bvhgpu_debug_mid_prof(out, 0)
n = max(out[0], 1)
print("levels(block0, 5 builds):", out[0])
for i, name in ((1, "bucket+scan"), (2, "sort/move"), (3, "stats"), (6, "select (4a)"), (4, "nodes+children (4b)"), (7, "  of which queue atomics"), (5, "reseg")):
    print(f"  {name:12s} {out[i] / n:10.0f} cycles/level")
