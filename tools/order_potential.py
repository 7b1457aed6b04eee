"""What could ray ordering buy at best?  The batch is sorted OUTSIDE the timed region (torch: keys, argsort, gather into a sorted copy) and the
unchanged walk (order knob 0: 64-ray blocks dealt round-robin to the workgroups, so the whole chip works on one stretch of the order at a
time) is timed on the sorted copy against the original stream.  Keys tried: origin cell (Morton) x direction octant at several
resolutions, octant-major, and a finer direction grid.   python tools/order_potential.py [rays_millions]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bvh_amd import Bvh, Context, RayBatch, scene, testbase as tb  # noqa: E402
from bvh_amd._lib import RAY_F32  # noqa: E402

n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 12_500_000
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
ctx.enable_timing(True)


def morton(c, bits):
    k = torch.zeros_like(c[:, 0])
    for b in range(bits):
        for a in range(3):
            k |= ((c[:, a] >> b) & 1) << (3 * b + a)
    return k


def keys(rays9, bounds, kind):
    o, d = rays9[:, 0:3], rays9[:, 3:6]
    lo = torch.tensor(bounds[:3], device=dev); hi = torch.tensor(bounds[3:], device=dev)
    oct_ = ((d[:, 0] < 0).long() | ((d[:, 1] < 0).long() << 1) | ((d[:, 2] < 0).long() << 2))
    def cell(bits):
        return ((o - lo) / (hi - lo) * (1 << bits)).long().clamp_(0, (1 << bits) - 1)
    if kind.startswith("cell"):
        bits = int(kind[4])
        return morton(cell(bits), bits) * 8 + oct_
    if kind.startswith("oct_cell"):
        bits = int(kind[8])
        return oct_ * (1 << (3 * bits)) + morton(cell(bits), bits)
    if kind.startswith("dir"):      # direction on a cube map face grid g x g (6 g^2 bins), then origin cell
        g, bits = int(kind[3:kind.index("c")]), int(kind[kind.index("c") + 1:])
        ad = d.abs(); ax = ad.argmax(dim=1)
        dm = d.gather(1, ax[:, None])[:, 0]
        u = d.gather(1, ((ax + 1) % 3)[:, None])[:, 0] / dm.abs(); v = d.gather(1, ((ax + 2) % 3)[:, None])[:, 0] / dm.abs()
        face = ax * 2 + (dm < 0).long()
        ui = ((u + 1) * 0.5 * g).long().clamp_(0, g - 1); vi = ((v + 1) * 0.5 * g).long().clamp_(0, g - 1)
        return ((face * g + ui) * g + vi) * (1 << (3 * bits)) + morton(cell(bits), bits)
    raise ValueError(kind)


which = os.environ.get("ORDER_SCENES", "standin,cubes1.2M").split(",")
scenes = []
if "standin" in which:
    _, aabbs_s, bounds_s = scene.parse_obj(scene.make_atrium_obj(16))
    scenes.append(("standin", aabbs_s, bounds_s))
if "cubes1.2M" in which:     # 1.2 M triangles: a scene that outgrows the L2s on the cube generator too
    scenes.append(("cubes1.2M", tb.create_n_cubes(100_000)[1], tb.default_bounds()))
if "cubes12m" in which:      # round 6: bench.py's beyond-BASELINE entry — 12 M triangles, the regime where the walk is HBM-bound (0.52 of peak)
    scenes.append(("cubes12m", tb.create_n_cubes(1_000_000)[1], tb.default_bounds()))
first_ray = int(os.environ.get("ORDER_FIRST", "62500000"))
kinds = os.environ.get("ORDER_KINDS", "none,cell3,cell4,cell5,oct_cell3,oct_cell4,dir4c3,dir8c2,dir8c3,dir16c2").split(",")
for scene_name, aabbs, bounds in scenes:
    bvh = Bvh.from_aabbs(torch.from_numpy(aabbs).to(dev), ctx)
    bvh.flatten_in_place()
    buf = torch.empty(n * RAY_F32.itemsize, dtype=torch.uint8, device=dev)
    rays = RayBatch.generate(first_ray, n, bounds, buf, np.float32, ctx)
    torch.cuda.synchronize()
    r9 = buf.view(torch.float32).view(n, 9)
    for kind in kinds:
        if kind == "none":
            sbuf = buf
        else:
            k = keys(r9, np.asarray(bounds, np.float32), kind)
            perm = torch.argsort(k)
            sbuf = r9[perm].contiguous().view(torch.uint8).view(-1)
            del k, perm
        torch.cuda.synchronize()
        sr = RayBatch.from_device(sbuf, n, np.float32)
        ks, ts = [], []
        for rep in range(4):
            st = bvh.traverse_batch(sr, fetch=False)[3]
            t = ctx.last_timings()
            ks.append(t["traverse_kernel_ms"]); ts.append(t["traverse_total_ms"])
        print(f"{scene_name:9s} {n / 1e6:5.1f} M rays sorted by {kind:10s}: walk {np.median(ks):7.3f} ms  total {np.median(ts):7.3f} ms  hits {st['hits']}", flush=True)
        if kind != "none":
            del sbuf
        torch.cuda.empty_cache()
    bvh.close()
