"""Developer diagnostic: time the traversal kernel under different tuning knobs and check that the CSR
result does not change.  Usage on the GPU box:  python tools/trav_sweep.py [n_cubes] [n_rays]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bvh_amd import Bvh, Context, RayBatch, testbase as tb  # noqa: E402
from bvh_amd._lib import RAY_F32  # noqa: E402

n_cubes = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
bounds = tb.default_bounds()
_, aabbs = tb.create_n_cubes(n_cubes, bounds)
bvh = Bvh.from_aabbs(torch.from_numpy(aabbs).to(dev), ctx)
bvh.flatten_in_place()
buf = torch.empty(R * RAY_F32.itemsize, dtype=torch.uint8, device=dev)
rays = RayBatch.generate(0, R, bounds, buf, np.float32, ctx)
torch.cuda.synchronize()


def run(variant, wpc=32, refill=1, reps=10, rpl=1, K=2048, thr=1024, split=1):
    ctx.set_tuning(0, variant); ctx.set_tuning(4, K); ctx.set_tuning(5, thr); ctx.set_tuning(6, split)
    ctx.enable_timing(False)
    off, idx, _, st = bvh.traverse_batch(rays, stats=True)
    ctx.enable_timing(True)
    ts = []
    for _ in range(reps):
        bvh.traverse_batch(rays, fetch=False)
        ts.append(ctx.last_timings()["traverse_kernel_ms"])
    ctx.enable_timing(False)
    return off, idx, st, float(np.median(ts)), float(np.min(ts))


ref = run(0)
print(f"variant 0 (one ray per lane): median {ref[3]:.4f} ms  min {ref[4]:.4f} ms  stats {ref[2]} util={ref[2]['device_steps'] / 64 / ref[2]['wave_steps']:.3f}")
for K, thr, split in ((2048, 1024, 0), (2048, 1024, 1), (5056, 1024, 1), (1280, 512, 1)):
    off, idx, st, med, mn = run(2, K=K, thr=thr, split=split)
    same = np.array_equal(off, ref[0]) and np.array_equal(idx, ref[1]) and {k: v for k, v in st.items() if k != 'wave_steps'} == {k: v for k, v in ref[2].items() if k != 'wave_steps'}
    print(f"lds-top K={K} threads={thr} split={split}: median {med:.4f} ms  min {mn:.4f} ms  same={same} util={st['device_steps'] / 64 / st['wave_steps']:.3f} wsteps={st['wave_steps']}", flush=True)
