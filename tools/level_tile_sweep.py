"""developer check: build time of the two-launch level schedule by tile size (BVHGPU_TUNE_BUILD_LEVEL_TILE) over scene sizes; the BvhNode array of
every tile size is compared byte by byte with the default's (a tile is a scheduling unit: the tree may not depend on it).
python tools/level_tile_sweep.py [cubes ...]"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bvh_amd import Bvh, Context, testbase as tb  # noqa: E402
from bvh_amd._lib import TUNE_BUILD_LEVEL_LAUNCHES, TUNE_BUILD_LEVEL_TILE  # noqa: E402

dev = torch.device("cuda", 0)
sizes = [int(x) for x in sys.argv[1:]] or [30_000, 100_000, 300_000, 1_000_000]
tiles = [int(x) for x in os.environ.get("TILES", "0,512,1024,2048,4096,8192").split(",")]
for cubes in sizes:
    _, a = tb.create_n_cubes(cubes, tb.default_bounds())
    aabbs = torch.from_numpy(a).to(dev)
    ref, row = None, []
    for tile in tiles:
        ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
        ctx.set_tuning(TUNE_BUILD_LEVEL_LAUNCHES, 2)
        ctx.set_tuning(TUNE_BUILD_LEVEL_TILE, tile)
        bvh = Bvh.from_aabbs(aabbs, ctx)
        ctx.enable_timing(True)
        bt = []
        for _ in range(7):
            bvh.rebuild(aabbs)
            bt.append(ctx.last_timings()["build_ms"])
        h = hashlib.sha256(bvh.nodes.tobytes()).hexdigest()
        ref = ref or h
        row.append(f"{tile}: {float(np.median(bt)):7.3f}" + ("" if h == ref else " NODES DIFFER"))
        bvh.close(); ctx.close()
    print(f"{12 * cubes:9d} triangles, build ms by tile  " + "   ".join(row), flush=True)
