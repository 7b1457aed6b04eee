"""Developer check of the wide walk's resident-slot count and LDS stack depth on BASELINE configs[1]:
python tools/slots_diag.py [f32|f64]  — HIP-event kernel time (median of 9) per setting, hits checked equal."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bvh_amd import Bvh, Context, RayBatch, testbase as tb  # noqa: E402
from bvh_amd._lib import RAY_F32, RAY_F64, TUNE_WIDE_SLOTS, TUNE_WIDE_STACK_LDS  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
R = 1_000_000
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
ref = None
for stack, slots in [(6, 0), (6, 0), (0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (6, 0), (8, 0), (6, 341), (6, 450)]:
    ctx.set_tuning(TUNE_WIDE_STACK_LDS, stack); ctx.set_tuning(TUNE_WIDE_SLOTS, slots)
    ts, tt = [], []
    for _ in range(9):
        st = bvh.traverse_batch(rays, fetch=False)[3]
        t = ctx.last_timings()
        ts.append(t["traverse_kernel_ms"]); tt.append(t["traverse_total_ms"])
    ref = st["hits"] if ref is None else ref
    flag = "" if st["hits"] == ref else "   !!! hits differ"
    print(f"stack_lds {stack:2d} slots {slots or 'fit':>4}: kernel {np.median(ts):7.4f} ms  total {np.median(tt):7.4f} ms  hits {st['hits']}{flag}", flush=True)
