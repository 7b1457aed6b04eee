"""Developer sweep of the wide walk's geometry knobs on BASELINE configs[1] (and its f64 twin):
python tools/wide_sweep.py [f32|f64] [rays]   — HIP-event kernel time (median of 7) per setting, hits checked equal."""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bvh_amd import Bvh, Context, RayBatch, testbase as tb  # noqa: E402
from bvh_amd._lib import (RAY_F32, RAY_F64, TUNE_TRAVERSE_VARIANT, TUNE_WIDE_ITEMS_LOG4, TUNE_WIDE_SLOTS, TUNE_WIDE_STACK_LDS,  # noqa: E402
                          TUNE_WIDE_THREADS, TUNE_WIDE_WG_PER_CU)

dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
npdt = np.float32 if dt == "f32" else np.float64
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
bounds = tb.default_bounds()
_, aabbs = tb.create_n_cubes(10_000, bounds)
a = torch.from_numpy(aabbs.astype(npdt)).to(dev)
buf = torch.empty(R * (RAY_F32 if dt == "f32" else RAY_F64).itemsize, dtype=torch.uint8, device=dev)
rays = RayBatch.generate(0, R, bounds, buf, npdt, ctx)
bvh = Bvh.from_aabbs(a, ctx)
bvh.flatten_in_place()
ctx.enable_timing(True)


def run(label):
    ts, tt = [], []
    hits = None
    for _ in range(7):
        st = bvh.traverse_batch(rays, fetch=False)[3]
        t = ctx.last_timings()
        ts.append(t["traverse_kernel_ms"]); tt.append(t["traverse_total_ms"])
        hits = st["hits"]
    print(f"{label:60s} kernel {np.median(ts):7.4f} ms  total {np.median(tt):7.4f} ms  hits {hits}", flush=True)
    return hits


ctx.set_tuning(TUNE_TRAVERSE_VARIANT, 2)
ref = run("variant 2 (binary, LDS top, 2 items/ray)")
ctx.set_tuning(TUNE_TRAVERSE_VARIANT, 3)
threads_opts = [1024, 512] if dt == "f32" else [512, 256]
for items, wg, thr, stack, slots in itertools.chain(
        itertools.product([2, 1, 0], [2, 1], threads_opts[:1], [8], [0]),
        itertools.product([2, 1], [2], threads_opts[1:], [8], [0]),
        itertools.product([2], [2], threads_opts[:1], [4, 6, 10, 12], [0]),
        itertools.product([2], [2], threads_opts[:1], [8], [85, 341])):
    if wg * thr > 2048:
        continue
    ctx.set_tuning(TUNE_WIDE_ITEMS_LOG4, items); ctx.set_tuning(TUNE_WIDE_WG_PER_CU, wg); ctx.set_tuning(TUNE_WIDE_THREADS, thr)
    ctx.set_tuning(TUNE_WIDE_STACK_LDS, stack); ctx.set_tuning(TUNE_WIDE_SLOTS, slots)
    try:
        h = run(f"wide items 4^{items} wg/cu {wg} threads {thr} stack_lds {stack} slots {slots or 'fit'}")
        if h != ref:
            print("   !!! hits differ", h, ref)
    except Exception as e:  # noqa: BLE001
        print("   failed:", e)
