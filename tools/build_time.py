"""developer check: build time (median of 7 rebuilds) and a hash of the BvhNode array by scene size, for whichever library is loaded
(BVH_AMD_SO=<variant>; tools/ab_builds.sh runs it over several builds side by side).   python tools/build_time.py [cubes ...]"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bvh_amd import Bvh, Context, testbase as tb  # noqa: E402

dev = torch.device("cuda", 0)
sizes = [int(x) for x in sys.argv[1:]] or [30_000, 100_000, 300_000, 1_000_000]
row = []
for cubes in sizes:
    _, a = tb.create_n_cubes(cubes, tb.default_bounds())
    aabbs = torch.from_numpy(a).to(dev)
    ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    bvh = Bvh.from_aabbs(aabbs, ctx)
    ctx.enable_timing(True)
    bt = []
    for _ in range(7):
        bvh.rebuild(aabbs)
        bt.append(ctx.last_timings()["build_ms"])
    row.append(f"{12 * cubes / 1e6:.2f} M: {float(np.median(bt)):7.3f} ms {hashlib.sha256(bvh.nodes.tobytes()).hexdigest()[:8]}")
    bvh.close(); ctx.close()
print(os.path.basename(os.environ.get("BVH_AMD_SO", "in-tree")), "  ".join(row), flush=True)
