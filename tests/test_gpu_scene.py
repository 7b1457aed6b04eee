"""BASELINE.json configs[2] and configs[3] on the stand-in for the reference's missing media/sponza.obj
(bvh_amd.scene.make_atrium_obj, ingested through the OBJ loader like load_sponza_scene, testbase.rs:619-634):
oracle parity at sizes the oracle finishes in seconds, size-independent properties at the full ray counts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import bvh_amd
    if bvh_amd.device_count() <= 0:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    return bvh_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import orc as o
    return o


@pytest.fixture(scope="module")
def atrium(eng):
    from bvh_amd import scene
    tris, aabbs, bounds = scene.parse_obj(scene.make_atrium_obj(4))
    return tris, aabbs, bounds


def _camera(bounds):
    from bvh_amd.api import camera
    c = (bounds[:3] + bounds[3:]) * 0.5                    # pinhole at the scene-bounds centre
    return camera(c, c + np.array([1.0, -0.15, 0.25]), fov_y_deg=70.0, aspect=4000 / 2500)


def test_config2_primary_rays_parity(eng, orc, atrium):
    """coherent primary rays on the stand-in: device ray generator == restatement, build / flatten / CSR /
    per-candidate triangle Intersection / closest hit == oracle, bit for bit (160 x 100 image)."""
    import torch
    from bvh_amd._lib import RAY_F32
    tris, aabbs, bounds = atrium
    cam = _camera(bounds)
    W, H = 160, 100
    ctx = eng.default_context()
    buf = torch.empty(W * H * RAY_F32.itemsize, dtype=torch.uint8, device="cuda")
    rb = eng.RayBatch.primary(cam, W, H, 0, W * H, buf, np.float32, ctx)
    ctx.synchronize()
    rays = buf.cpu().numpy().view(RAY_F32)
    assert rays.tobytes() == orc.primary_rays(cam, W, H, 0, W * H).tobytes()
    bvh = eng.Bvh.from_aabbs(aabbs)
    ot = orc.build(aabbs)
    assert bvh.nodes.tobytes() == ot.nodes.tobytes()
    flat = bvh.flatten()
    oflat = orc.flatten(ot.nodes)
    assert flat.nodes.tobytes() == oflat.tobytes()
    flat.set_triangles(tris)
    off, idx, isect, st = flat.intersect_triangles(rb, stats=True)
    ooff, oidx, _, ost = orc.traverse_flat(oflat, aabbs, rays, threads=orc.max_threads())
    oisect, oclosest, oprim = orc.triangle_stage(tris, rays, ooff, oidx)
    assert np.array_equal(off, ooff) and np.array_equal(idx, oidx) and isect.tobytes() == oisect.tobytes()
    assert st["visited"] == ost["visited"]
    cl, prim, _ = flat.closest_hits(rb)
    assert cl.tobytes() == oclosest.tobytes() and np.array_equal(prim, oprim)
    assert np.isfinite(cl[:, 0]).mean() > 0.5 and len(idx) > 3 * W * H   # several candidates per ray


def test_config2_full_size_10m_primary_rays(eng, orc, atrium):
    """4000 x 2500 = 10 M coherent primary rays: fused closest hit over the whole image; properties that do not
    depend on the size: a window of the image equals the oracle, the two walk kernels agree on a checksum,
    the fused closest hit equals the argmin over the per-candidate Intersections of the CSR pass."""
    import torch
    from bvh_amd import Context
    from bvh_amd._lib import RAY_F32, TUNE_TRAVERSE_VARIANT
    tris, aabbs, bounds = atrium
    cam = _camera(bounds)
    W, H = 4000, 2500
    ctx = Context(0)
    flat = eng.Bvh.from_aabbs(aabbs, ctx).flatten()
    flat.set_triangles(tris)
    buf = torch.empty(W * H * RAY_F32.itemsize, dtype=torch.uint8, device="cuda")
    rb = eng.RayBatch.primary(cam, W, H, 0, W * H, buf, np.float32, ctx)
    cl, prim, st = flat.closest_hits(rb, stats=True)
    assert cl.shape == (W * H, 3) and st["hits"] > 3 * W * H
    # rows 1200..1209 against the oracle (40 000 rays)
    first, n = 1200 * W, 10 * W
    rays = orc.primary_rays(cam, W, H, first, n)
    oflat = orc.flatten(orc.build(aabbs).nodes)
    ooff, oidx, _, _ = orc.traverse_flat(oflat, aabbs, rays, threads=orc.max_threads())
    _, oclosest, oprim = orc.triangle_stage(tris, rays, ooff, oidx)
    assert cl[first:first + n].tobytes() == oclosest.tobytes() and np.array_equal(prim[first:first + n], oprim)
    # the other walk kernel gives the same image
    ctx.set_tuning(TUNE_TRAVERSE_VARIANT, 0)
    cl0, prim0, st0 = flat.closest_hits(rb, stats=True)
    assert cl0.tobytes() == cl.tobytes() and np.array_equal(prim0, prim) and st0["visited"] == st["visited"]
    ctx.set_tuning(TUNE_TRAVERSE_VARIANT, 2)
    # CSR pass on a 1 M-ray band: argmin of the per-candidate distances (first on ties) == fused closest hit
    band = eng.RayBatch.primary(cam, W, H, 800 * W, 250 * W, buf, np.float32, ctx)
    off, idx, isect, _ = flat.intersect_triangles(band)
    d = isect[:, 0]
    cnt = np.diff(off.astype(np.int64))
    assert off[-1] == len(idx) and cnt.min() >= 0
    seg = np.repeat(np.arange(len(cnt)), cnt)
    best = np.full(len(cnt), np.inf, dtype=np.float32)
    np.minimum.at(best, seg, d)
    assert np.array_equal(best, cl[800 * W: 1050 * W, 0])
    is_first_min = (d == best[seg]) & np.isfinite(d)
    pos = np.full(len(cnt), len(idx), dtype=np.int64)
    np.minimum.at(pos, seg[is_first_min], np.nonzero(is_first_min)[0])
    hit = np.isfinite(best)
    assert np.array_equal(idx[pos[hit]], prim[800 * W: 1050 * W][hit])
    assert np.all(prim[800 * W: 1050 * W][~hit] == 0xFFFFFFFF)


def test_config3_incoherent_rays_shard(eng, orc, atrium):
    """configs[3]: the create_ray stream over the scene bounds, sharded 8 ways (shard i owns rays
    [i*R/8, (i+1)*R/8)).  Oracle parity on a slice of shard 5, then the whole 12.5 M-ray shard
    through the fused closest hit with chunked == whole."""
    import torch
    from bvh_amd import Context, dist as bdist
    from bvh_amd._lib import RAY_F32
    tris, aabbs, bounds = atrium
    ctx = Context(0)
    flat = eng.Bvh.from_aabbs(aabbs, ctx).flatten()
    flat.set_triangles(tris)
    R = 100_000_000 // 8
    first, count = bdist.shard_range(5, 8, R)
    assert (first, count) == (5 * R, R)
    buf = torch.empty(R * RAY_F32.itemsize, dtype=torch.uint8, device="cuda")
    rb = eng.RayBatch.generate(first, R, bounds, buf, np.float32, ctx)
    cl, prim, st = flat.closest_hits(rb, stats=True)
    # oracle on the first 60 000 rays of the shard
    n = 60_000
    rays = orc.create_rays(first, n, bounds)
    oflat = orc.flatten(orc.build(aabbs).nodes)
    ooff, oidx, _, ost = orc.traverse_flat(oflat, aabbs, rays, threads=orc.max_threads())
    _, oclosest, oprim = orc.triangle_stage(tris, rays, ooff, oidx)
    assert cl[:n].tobytes() == oclosest.tobytes() and np.array_equal(prim[:n], oprim)
    # CSR parity on the same slice
    sl = eng.RayBatch.generate(first, n, bounds, buf, np.float32, ctx)
    off, idx, _, st2 = flat.traverse_batch(sl, stats=True)
    assert np.array_equal(off, ooff) and np.array_equal(idx, oidx) and st2["visited"] == ost["visited"]
    # chunked == whole: the last 2 M rays of the shard as their own batch
    tail0 = R - 2_000_000
    tb_ = eng.RayBatch.generate(first + tail0, 2_000_000, bounds, buf, np.float32, ctx)
    cl2, prim2, _ = flat.closest_hits(tb_)
    assert cl2.tobytes() == cl[tail0:].tobytes() and np.array_equal(prim2, prim[tail0:])
    # the scene blob that travels to the peers carries everything the shard needs (slots + triangles)
    blob = torch.empty(flat.scene_nbytes(), dtype=torch.uint8, device="cuda")
    flat.scene_export(blob)
    peer = eng.FlatBvh.scene_import(blob, blob.numel(), ctx)
    cl3, prim3, _ = peer.closest_hits(tb_)
    assert cl3.tobytes() == cl2.tobytes() and np.array_equal(prim3, prim2)


# ---- round 3: EVERY ray of configs[2] / [3] diffed against the oracle, at the bench's scene size ----------
def _full_csr_diff(eng, orc, flat, aabbs, rb, oracle_rays, n, chunk=1_000_000, coherent=False):
    """ONE GPU traversal of the whole batch (the walk the bench times: the wide walk with whole rays above ~2 M rays), CSR fetched
    once; the oracle goes through the rays in 1 M-ray chunks and each chunk is compared with its slice — offsets, indices (i.e.
    the per-ray ORDER of flat_bvh.rs:396-431) — and the reference-equivalent visit counters with a STATS pass."""
    import hashlib
    off, idx, _, _ = flat.traverse_batch(rb, coherent=coherent)   # (coherent: hits handed over through per-ray slots, as bench.py's configs[2] does)
    st = flat.traverse_batch(rb, stats=True, fetch=False)[3]
    oflat = orc.flatten(orc.build(aabbs).nodes)
    V = VL = H = 0
    hg, ho = hashlib.sha256(), hashlib.sha256()
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        ooff, oidx, _, ost = orc.traverse_flat(oflat, aabbs, oracle_rays(c0, m), threads=orc.max_threads())
        base = int(off[c0])
        goff, gidx = off[c0:c0 + m + 1] - np.uint32(base), idx[base:int(off[c0 + m])]
        assert np.array_equal(goff, ooff), f"offsets differ in rays [{c0}, {c0 + m})"
        assert np.array_equal(gidx, oidx), f"indices differ in rays [{c0}, {c0 + m})"
        hg.update(goff.tobytes()); hg.update(gidx.tobytes()); ho.update(ooff.tobytes()); ho.update(oidx.tobytes())
        V += ost["visited"]; VL += ost["leaf_visits"]; H += ost["hits"]
    assert hg.hexdigest() == ho.hexdigest()
    assert len(idx) == H and (st["visited"], st["leaf_visits"], st["hits"]) == (V, VL, H)
    return H


@pytest.fixture(scope="module")
def atrium16(eng):
    from bvh_amd import scene
    return scene.parse_obj(scene.make_atrium_obj(16))      # the scene bench.py's configs[2] / [3] run on (165 k triangles)


def test_config2_all_10m_primary_rays_against_oracle(eng, orc, atrium16):
    import torch
    from bvh_amd import Context
    from bvh_amd._lib import RAY_F32
    tris, aabbs, bounds = atrium16
    cam = _camera(bounds)
    W, H = 4000, 2500
    ctx = Context(0)
    flat = eng.Bvh.from_aabbs(aabbs, ctx).flatten()
    buf = torch.empty(W * H * RAY_F32.itemsize, dtype=torch.uint8, device="cuda")
    rb = eng.RayBatch.primary(cam, W, H, 0, W * H, buf, np.float32, ctx)
    hits = _full_csr_diff(eng, orc, flat, aabbs, rb, lambda c0, m: orc.primary_rays(cam, W, H, c0, m), W * H, coherent=True)
    assert hits > 3 * W * H


def test_config3_all_12_5m_shard_rays_against_oracle(eng, orc, atrium16):
    import torch
    from bvh_amd import Context
    from bvh_amd._lib import RAY_F32
    tris, aabbs, bounds = atrium16
    ctx = Context(0)
    flat = eng.Bvh.from_aabbs(aabbs, ctx).flatten()
    R = 100_000_000 // 8
    first = 5 * R                                           # the shard rank 5 of 8 owns
    buf = torch.empty(R * RAY_F32.itemsize, dtype=torch.uint8, device="cuda")
    rb = eng.RayBatch.generate(first, R, bounds, buf, np.float32, ctx)
    hits = _full_csr_diff(eng, orc, flat, aabbs, rb, lambda c0, m: orc.create_rays(first + c0, m, bounds), R)
    assert hits > R
