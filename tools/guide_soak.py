"""developer soak of the f64 guide walk: grazing rays (aimed at faces / edges / corners of the shapes' boxes) on several scenes, the guide
walk's CSR against the f64 walk's on the same tree (BVHGPU_TUNE_WIDE_F64_GUIDE 1 / 0): python tools/guide_soak.py [rays per scene]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bvh_amd import Bvh, Context, RayBatch, scene, testbase as tb  # noqa: E402
from bvh_amd._lib import RAY_F64, TUNE_WIDE_F64_GUIDE, TUNE_WIDE_ITEMS_LOG4, WALK_F64_GUIDE  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
rng = np.random.default_rng(123)


def grazing(aabbs, n):
    lo, hi = aabbs[:, :3].min(axis=0), aabbs[:, 3:].max(axis=0)
    b = aabbs[rng.integers(0, len(aabbs), n)]
    pick = rng.integers(0, 3, (n, 3))
    u = rng.uniform(0, 1, (n, 3))
    tgt = np.where(pick == 0, b[:, :3], np.where(pick == 1, b[:, 3:], b[:, :3] + u * (b[:, 3:] - b[:, :3])))
    o = rng.uniform(lo - 0.5 * (hi - lo), hi + 0.5 * (hi - lo), (n, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(n, RAY_F64)
    rays["o"], rays["d"] = o, d
    with np.errstate(divide="ignore"):
        rays["inv"] = 1.0 / d
    return rays


scenes = []
for cubes, scale, shift in ((3000, 1.0, 0.0), (20000, 1.0, 0.0), (3000, 1e-3, 0.0), (3000, 1.0, 5000.0), (3000, 1e4, 0.0)):
    _, a = tb.create_n_cubes(cubes)
    scenes.append((f"{cubes} cubes x{scale:g} +{shift:g}", a.astype(np.float64) * scale + shift))
_, a32, _ = scene.parse_obj(scene.make_atrium_obj(8))
scenes.append(("stand-in atrium, detail 8", a32.astype(np.float64)))
bad = 0
for name, aabbs in scenes:
    for items in (2, 0):
        ctx = Context(0)
        ctx.set_tuning(TUNE_WIDE_ITEMS_LOG4, items)
        flat = Bvh.from_aabbs(aabbs, ctx).flatten()
        rays = grazing(aabbs, N)
        rb = RayBatch(len(rays), np.float64, host=rays)
        off, idx, _, st = flat.traverse_batch(rb)
        ctx.set_tuning(TUNE_WIDE_F64_GUIDE, 0)
        off0, idx0, _, st0 = flat.traverse_batch(rb)
        same = np.array_equal(off, off0) and np.array_equal(idx, idx0)
        bad += 0 if same else 1
        print(f"{name:32s} items 4^{items}: guide walk {'ran' if st['walk'] & WALK_F64_GUIDE else 'REPLAYED in f64'}, {len(idx)} hits, lists {'equal' if same else 'DIFFER'}", flush=True)
print("FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
