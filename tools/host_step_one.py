import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bvh_amd import Bvh, Context, HostStep, testbase as tb
from bvh_amd._lib import TUNE_HOST_CHUNKS
ctx = Context(0)
ctx.set_tuning(TUNE_HOST_CHUNKS, int(os.environ.get("HOST_CHUNKS", "0")))
_, aabbs = tb.create_n_cubes(10000)
R = 1_000_000
k = np.arange(0, R, dtype=np.uint64); b = tb.default_bounds()
bvh = Bvh.from_aabbs(aabbs, ctx); bvh.flatten_in_place()
hs = HostStep(bvh, len(aabbs), R, np.float32)
hs.aabbs[:] = aabbs; hs.origins[:] = tb.next_point3_at(2 * k + 1, b); hs.directions[:] = tb.next_point3_at(2 * k + 2, b)
for _ in range(5): hs.run()
t0 = time.perf_counter()
for _ in range(30): hs.run()
print("ms/step", (time.perf_counter() - t0) / 30 * 1e3)
