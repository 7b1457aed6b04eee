"""developer check: the asynchronous step (rebuild + flatten + 1 M-ray index batch, everything resident in HBM) by scene size with a tuning knob at two values:
python tools/step_time.py <knob> <value_a> <value_b> [cubes ...]      e.g. 21 1 0 10000 50000 (BVHGPU_TUNE_FLATTEN_INLINE on / off)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bvh_amd import Bvh, Context, RayBatch, testbase as tb  # noqa: E402
from bvh_amd._lib import RAY_F32, TRAVERSE_RAYS_READY, TUNE_FLATTEN_LAZY  # noqa: E402

knob, va, vb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sizes = [int(x) for x in sys.argv[4:]] or [10_000, 30_000, 50_000, 100_000]
dev = torch.device("cuda", 0)
R = 1_000_000
for cubes in sizes:
    _, a = tb.create_n_cubes(cubes, tb.default_bounds())
    aabbs = torch.from_numpy(a).to(dev)
    row = []
    for rep in range(2):
        for v in (va, vb):
            ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
            ctx.set_tuning(TUNE_FLATTEN_LAZY, 3)
            ctx.set_tuning(knob, v)
            bvh = Bvh.from_aabbs(aabbs, ctx)
            bvh.flatten_in_place()
            buf = torch.empty(R * RAY_F32.itemsize, dtype=torch.uint8, device=dev)
            rays = RayBatch.generate(0, R, tb.default_bounds(), buf, np.float32, ctx)
            ts = []
            for block in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(40):
                    bvh.rebuild_async(aabbs)
                    st = bvh.traverse_async(rays, flags=TRAVERSE_RAYS_READY).wait()
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 40)
            row.append(f"knob={v}: {np.median(ts) * 1e3:.4f} ms")
            bvh.close(); ctx.close()
    print(f"{12 * cubes / 1e6:.2f} M triangles  " + "   ".join(row), flush=True)
