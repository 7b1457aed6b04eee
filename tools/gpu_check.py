"""Developer diagnostic (not a test): run every stage on the GPU, diff against the oracle, print details.
Usage on the GPU box:  python tools/gpu_check.py [n_cubes] [n_rays]
"""
import os
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bvh_amd  # noqa: E402
from bvh_amd import Bvh, FlatBvh, RayBatch, testbase as tb  # noqa: E402
from oracle import orc  # noqa: E402


def diff_nodes(g, o, name):
    if g.tobytes() == o.tobytes():
        print(f"  {name}: byte-identical ({len(g)} entries)")
        return True
    bad = [i for i in range(min(len(g), len(o))) if g[i].tobytes() != o[i].tobytes()]
    print(f"  {name}: MISMATCH in {len(bad)} of {len(o)} entries (len gpu {len(g)}); first: {bad[:5]}")
    for i in bad[:3]:
        print("    gpu   ", g[i])
        print("    oracle", o[i])
    return False


def run_case(aabbs, rays_host, label, dtype=np.float32):
    print(f"== {label}: n={len(aabbs)} rays={len(rays_host)} dtype={np.dtype(dtype).name}")
    ok = True
    aabbs = aabbs.astype(dtype)
    t0 = time.time()
    ot = orc.build(aabbs)
    oflat = orc.flatten(ot.nodes)
    print(f"  oracle build+flatten {time.time() - t0:.3f}s")
    t0 = time.time()
    bvh = Bvh.from_aabbs(aabbs)
    bvh.ctx.synchronize()
    print(f"  gpu build {time.time() - t0:.4f}s levels={bvh.build_levels}")
    ok &= diff_nodes(bvh.nodes, ot.nodes, "BvhNode array")
    sn = bvh.shape_nodes
    print("  shape->node map:", "identical" if np.array_equal(sn, ot.shape_node) else "MISMATCH")
    ok &= np.array_equal(sn, ot.shape_node)
    flat = bvh.flatten()
    ok &= diff_nodes(flat.nodes, oflat, "FlatNode array")
    rb = RayBatch(len(rays_host), dtype, host=rays_host)
    t0 = time.time()
    off, idx, ts, st = flat.traverse_batch(rb, want_t=True, stats=True)
    print(f"  gpu traverse {time.time() - t0:.4f}s stats={st}")
    ooff, oidx, ots, ost = orc.traverse_flat(oflat, aabbs, rays_host, want_t=True, threads=orc.max_threads())
    print(f"  oracle stats={ost}")
    same_off = np.array_equal(off, ooff)
    same_idx = np.array_equal(idx, oidx)
    print("  offsets:", "identical" if same_off else "MISMATCH", " indices:", "identical" if same_idx else "MISMATCH")
    ok &= same_off and same_idx
    if same_idx and len(idx):
        err = np.max(np.abs(ts - ots) / np.maximum(np.abs(ots), 1))
        print(f"  t-slice max rel err {err:.3e}")
    ok &= st["visited"] == ost["visited"] and st["leaf_visits"] == ost["leaf_visits"]
    print("  visited counters:", "match" if st["visited"] == ost["visited"] else f"MISMATCH {st} vs {ost}")
    return ok


def main():
    n_cubes = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    print("devices:", bvh_amd.device_count())
    results = {}
    cases = []
    boxes = tb.generate_aligned_boxes_aabbs()
    cases.append(("aligned boxes", boxes,
                  orc.make_rays([[-1000, 0, 0], [0, -1000, 0], [6, 0.5, 0]], [[1, 0, 0], [0, 1, 0], [-2, -1, 0]]), np.float32))
    _, a100 = tb.create_n_cubes(100)
    cases.append(("1200 tris", a100, orc.create_rays(0, 1000), np.float32))
    rng = np.random.default_rng(3)
    lo = rng.uniform(-50, 50, size=(5000, 3)).astype(np.float32)
    ext = rng.uniform(0, 4, size=(5000, 3)).astype(np.float32)
    lo[1000:1300] = lo[1000]
    ext[1000:1300] = ext[1000]
    rb = np.concatenate([lo, lo + ext], axis=1)
    ro = rng.uniform(-60, 60, size=(2000, 3)).astype(np.float32)
    rd = rng.normal(size=(2000, 3)).astype(np.float32)
    cases.append(("random boxes w/ 300 identical", rb, orc.make_rays(ro, rd), np.float32))
    _, abig = tb.create_n_cubes(n_cubes)
    cases.append((f"{n_cubes * 12} tris", abig, orc.create_rays(0, n_rays), np.float32))
    r64 = orc.create_rays(0, 20000)
    cases.append(("1200 tris f64", a100, orc.make_rays(r64["o"], r64["d"], np.float64), np.float64))
    cases.append((f"{n_cubes * 12} tris f64", abig, orc.make_rays(r64["o"], r64["d"], np.float64), np.float64))
    try:
        import torch
        from bvh_amd._lib import RAY_F32, RAY_F64
        ctx = bvh_amd.default_context()
        for dt, rdt in ((np.float32, RAY_F32), (np.float64, RAY_F64)):
            buf = torch.empty(5000 * rdt.itemsize, dtype=torch.uint8, device="cuda")
            RayBatch.generate(123456, 5000, tb.default_bounds(), buf, dt)
            ctx.synchronize()
            got = buf.cpu().numpy().view(rdt)
            exp = orc.create_rays(123456, 5000)
            if dt == np.float64:
                # f64 stream = f32 points widened, then Ray::new in f64
                st = 2 * 123456 * 0x9E3779B97F4A7C15 % (1 << 64)
                import ctypes as C
                o = np.zeros((5000, 3), np.float32); d = np.zeros((5000, 3), np.float32)
                seed = C.c_uint64(st)
                b = tb.default_bounds()
                for i in range(5000):
                    orc.lib().orc_next_point3(C.byref(seed), b.ctypes.data_as(C.c_void_p), o[i].ctypes.data_as(C.c_void_p))
                    orc.lib().orc_next_point3(C.byref(seed), b.ctypes.data_as(C.c_void_p), d[i].ctypes.data_as(C.c_void_p))
                exp = orc.make_rays(o.astype(np.float64), d.astype(np.float64), np.float64)
            same = got.tobytes() == exp.tobytes()
            print(f"== device ray stream {np.dtype(dt).name}:", "byte-identical" if same else "MISMATCH")
            results[f"raygen {np.dtype(dt).name}"] = same
    except Exception:
        traceback.print_exc()
        results["raygen"] = False
    for label, aabbs, rays, dt in cases:
        try:
            results[label] = run_case(aabbs, rays, label, dt)
        except Exception:
            traceback.print_exc()
            results[label] = False
    print("SUMMARY", results)
    return 0 if all(results.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
