"""Input contract, the asynchronous step API and the robustness fixes of round 2 — on the GPU, through the C ABI,
checked against the oracle."""
import ctypes as C

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


def _chain(n, dtype=np.float32):
    """exponentially spaced boxes: a very unbalanced SAH tree (the builder's host-synchronised continuation)"""
    x = dtype(1.004) ** np.arange(n, dtype=dtype)
    lo = np.stack([x, np.zeros(n, dtype), np.zeros(n, dtype)], axis=1)
    return np.concatenate([lo, lo + dtype(0.5)], axis=1).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [2, 70, 900, 5000])
def test_nan_and_inf_input_is_rejected(eng, n, dtype):
    """The reference panics on a NaN / infinite centroid (bvh_node.rs:214-217, to_usize().unwrap()): the engine returns
    INVALID_ARG from every builder tier, builds nothing, and the tree object stays usable."""
    from bvh_amd._lib import INVALID_ARG, BvhGpuError
    rng = np.random.default_rng(5)
    lo = rng.uniform(-10, 10, size=(n, 3)).astype(dtype)
    good = np.concatenate([lo, lo + dtype(1)], axis=1)
    bvh = eng.Bvh.from_aabbs(good)
    for bad_value, col in ((np.nan, 1), (np.inf, 4), (-np.inf, 0)):
        bad = good.copy()
        bad[n // 2, col] = bad_value
        with pytest.raises(BvhGpuError) as e:
            eng.Bvh.from_aabbs(bad)
        assert e.value.status == INVALID_ARG and "bvh_node.rs:214-217" in str(e.value)
        with pytest.raises(BvhGpuError):
            bvh.rebuild(bad, flatten=True)
    # finite boxes whose centroid extent overflows: (c - cmin) / ext is NaN for some shape
    huge = good.copy()
    m = np.finfo(dtype).max
    huge[0] = [-m, 0, 0, -m * dtype(0.99), 1, 1]
    huge[1] = [m * dtype(0.99), 0, 0, m, 1, 1]
    with pytest.raises(BvhGpuError):
        eng.Bvh.from_aabbs(huge)
    bvh.rebuild(good, flatten=True)            # the object survives a rejected rebuild
    assert bvh.info()[0] == n
    # a single shape is never bucketed: accepted like in the reference
    one = np.array([[np.nan, 0, 0, 1, 1, 1]], dtype=dtype)
    assert eng.Bvh.from_aabbs(one).info()[0] == 1


def test_rebuild_flat_on_unbalanced_tree(eng, orc):
    """bvhgpu_rebuild_flat_* enqueues the flatten before the host knows that the level queue is drained; on an unbalanced
    tree that optimistic flatten runs over an unfinished node array: it must be harmless and the final arrays
    must be the oracle's — also when a smaller / larger tree reuses the buffers."""
    for n in (12000, 3000, 20000):
        aabbs = _chain(n)
        bvh = eng.Bvh.from_aabbs(_chain(64))
        bvh.rebuild(aabbs, flatten=True)
        ot = orc.build(aabbs)
        assert bvh.nodes.tobytes() == ot.nodes.tobytes()
        flat = bvh.flatten()
        assert flat.nodes.tobytes() == orc.flatten(ot.nodes).tobytes()
        if n == 12000:
            assert bvh.build_levels >= 5


def test_async_step_matches_sync(eng, orc):
    """bvhgpu_rebuild_flat_async + bvhgpu_traverse_async + bvhgpu_hits_wait from ONE host thread on two contexts
    (two streams): same CSR as the synchronous calls and the oracle."""
    import torch
    from bvh_amd import Bvh, Context, RayBatch, testbase as tb
    from bvh_amd._lib import RAY_F32
    from bvh_amd.api import _Hits
    bounds = tb.default_bounds()
    _, aabbs_np = tb.create_n_cubes(2000)
    aabbs = torch.from_numpy(aabbs_np).cuda()
    R = 100_000
    lanes = []
    for j in range(2):
        ctx = Context(0)
        buf = torch.empty(R * RAY_F32.itemsize, dtype=torch.uint8, device="cuda")
        rays = RayBatch.generate(j * R, R, bounds, buf, np.float32, ctx)
        tree = Bvh.from_aabbs(aabbs, ctx)
        lanes.append((ctx, tree, rays, _Hits(ctx), buf))
    ot = orc.build(aabbs_np)
    oflat = orc.flatten(ot.nodes)
    for _ in range(3):                       # several steps in flight, alternating streams
        for ctx, tree, rays, hits, _ in lanes:
            tree.rebuild_async(aabbs)
            tree.traverse_async(rays, hits)
        for j, (ctx, tree, rays, hits, _) in enumerate(lanes):
            st = hits.wait()
            off, idx = hits.fetch(R)
            ooff, oidx, _, _ = orc.traverse_flat(oflat, aabbs_np, orc.create_rays(j * R, R), threads=orc.max_threads())
            assert np.array_equal(off, ooff) and np.array_equal(idx, oidx) and st["hits"] == len(oidx)
    assert lanes[0][1].nodes.tobytes() == ot.nodes.tobytes()


def test_result_object_across_batch_sizes(eng, orc):
    """One result object through batches of very different sizes: the wide walk keeps its per-ray words, the scan-block sums
    (two sets, used alternately) and the counter sets zero BETWEEN batches instead of clearing them — a batch above 2 M rays
    takes the path with a reduce pass and a publishing launch in between."""
    import torch
    from bvh_amd import Bvh, Context, RayBatch, testbase as tb
    from bvh_amd._lib import RAY_F32
    from bvh_amd.api import _Hits
    ctx = Context(0)
    bounds = tb.default_bounds()
    _, aabbs = tb.create_n_cubes(3000)
    bvh = Bvh.from_aabbs(torch.from_numpy(aabbs).cuda(), ctx)
    bvh.flatten_in_place()
    oflat = orc.flatten(orc.build(aabbs).nodes)
    hits = _Hits(ctx)
    for R in (100_000, 2_300_000, 120_000, 40_000, 100_000, 20_000, 100_000):
        buf = torch.empty(R * RAY_F32.itemsize, dtype=torch.uint8, device="cuda")
        rays = RayBatch.generate(3, R, bounds, buf, np.float32, ctx)
        st = bvh.traverse_async(rays, hits).wait()
        off, idx = hits.fetch(R)
        ooff, oidx, _, _ = orc.traverse_flat(oflat, aabbs, orc.create_rays(3, R), threads=orc.max_threads())
        assert np.array_equal(off, ooff) and np.array_equal(idx, oidx) and st["hits"] == len(oidx), R


def test_async_step_unbalanced_tree_and_bad_input(eng, orc):
    """What the optimistic asynchronous launch cannot know is settled at the wait: an unbalanced tree is finished and the
    batch replayed; invalid input surfaces as the status the synchronous call would have returned."""
    import torch
    from bvh_amd import Bvh, Context, RayBatch
    from bvh_amd._lib import INVALID_ARG, BvhGpuError
    from bvh_amd.api import _Hits
    ctx = Context(0)
    n = 12000
    chain = _chain(n)
    o = np.zeros((20000, 3), np.float32); o[:, 0] = -1; o[:, 1] = np.linspace(0.01, 0.49, 20000); o[:, 2] = 0.25
    d = np.tile(np.array([1, 0, 0], np.float32), (20000, 1)); d[::3] = [1, 0.002, 0]
    rays_np = orc.make_rays(o, d)
    rays_t = torch.from_numpy(rays_np.view(np.uint8).reshape(-1)).cuda()
    rays = RayBatch.from_device(rays_t, len(rays_np), np.float32)
    dev = torch.from_numpy(chain).cuda()
    tree = Bvh.from_aabbs(_chain(100), ctx)
    hits = _Hits(ctx)
    tree.rebuild_async(dev)
    tree.traverse_async(rays, hits)
    st = hits.wait()
    off, idx = hits.fetch(len(rays_np))
    ot = orc.build(chain)
    ooff, oidx, _, _ = orc.traverse_flat(orc.flatten(ot.nodes), chain, rays_np, threads=orc.max_threads())
    assert np.array_equal(off, ooff) and np.array_equal(idx, oidx) and st["hits"] == len(oidx)
    assert tree.nodes.tobytes() == ot.nodes.tobytes() and tree.build_levels >= 5
    bad = chain.copy(); bad[77, 3] = np.nan
    tree.rebuild_async(torch.from_numpy(bad).cuda())
    tree.traverse_async(rays, hits)
    with pytest.raises(BvhGpuError) as e:
        hits.wait()
    assert e.value.status == INVALID_ARG
    tree.rebuild_async(dev).wait()            # and the objects are still usable
    tree.traverse_async(rays, hits)
    assert hits.wait()["hits"] == len(oidx)


def test_scene_import_rejects_corrupt_header(eng):
    """bvhgpu_scene_import trusts nothing in the blob header: sizes must follow from the counts and fit."""
    from bvh_amd import Bvh, FlatBvh
    from bvh_amd._lib import BvhGpuError
    rng = np.random.default_rng(3)
    lo = rng.uniform(-10, 10, size=(500, 3)).astype(np.float32)
    bvh = Bvh.from_aabbs(np.concatenate([lo, lo + 1], axis=1))
    bvh.flatten_in_place()
    nbytes = bvh.scene_nbytes()
    blob = np.zeros(nbytes, dtype=np.uint8)
    bvh.scene_export(blob)
    ok = FlatBvh.scene_import(blob, nbytes)
    assert ok.info()[0] == 500
    hdr = blob[:256].view(np.uint64)          # magic/dtype | n | n_trav | unfolded/pad | trav_bytes | aabb_bytes | slot_bytes | tri_bytes
    for word, value in ((1, 10**9), (2, 7), (4, int(hdr[4]) + 32), (5, 24), (7, 12345), (4, 2**63)):
        bad = blob.copy()
        bad[:256].view(np.uint64)[word] = value
        with pytest.raises(BvhGpuError):
            FlatBvh.scene_import(bad, nbytes)
    with pytest.raises(BvhGpuError):
        FlatBvh.scene_import(blob[: nbytes - 4096].copy(), nbytes - 4096)


def test_from_flat_rejects_non_binary_array(eng, orc):
    """an uploaded flat array that passes the index checks but is not a flattened BINARY tree"""
    from bvh_amd import FlatBvh
    from bvh_amd._lib import BvhGpuError
    rng = np.random.default_rng(4)
    lo = rng.uniform(-10, 10, size=(3, 3)).astype(np.float32)
    aabbs = np.concatenate([lo, lo + 1], axis=1)
    flat = np.zeros(6, dtype=orc.FLAT_F32)    # three navigator + leaf pairs side by side under the (implicit) root
    for k in range(3):
        flat[2 * k] = (aabbs[k, :3], aabbs[k, 3:], 2 * k + 1, 2 * k + 2, 0xFFFFFFFF)
        flat[2 * k + 1] = ((np.inf,) * 3, (-np.inf,) * 3, 0xFFFFFFFF, 2 * k + 2, k)
    with pytest.raises(BvhGpuError):
        FlatBvh.from_flat_nodes(flat, aabbs)


def test_rebuild_with_other_shape_count_drops_triangles(eng):
    """one triangle per shape: a rebuild with a different shape count invalidates the vertex array"""
    from bvh_amd import Bvh, RayBatch, testbase as tb
    from bvh_amd._lib import BvhGpuError
    from oracle import orc
    tris, aabbs = tb.create_n_cubes(20)
    bvh = Bvh.from_aabbs(aabbs)
    bvh.flatten_in_place()
    bvh.set_triangles(tris)
    rays = orc.create_rays(0, 100)
    rb = RayBatch(len(rays), np.float32, host=rays)
    bvh.closest_hits(rb)
    tris2, aabbs2 = tb.create_n_cubes(30)
    bvh.rebuild(aabbs2, flatten=True)
    with pytest.raises(BvhGpuError):
        bvh.closest_hits(rb)
    bvh.set_triangles(tris2)
    bvh.closest_hits(rb)


def test_ctx_may_be_finalised_before_what_was_made_on_it(eng, orc):
    """The cyclic collector finalises objects of a reference cycle in arbitrary order (a pytest.raises traceback holding a test's locals
    is enough): a Context closed while trees / result objects made on it are alive hands its destruction to the last of them instead
    of leaving them with a dangling ctx (round 4: this crashed the interpreter at the end of a test session)."""
    import gc
    from bvh_amd import Bvh, Context, RayBatch, testbase as tb
    from bvh_amd.api import _Hits
    _, aabbs = tb.create_n_cubes(200)
    rays = orc.create_rays(0, 2000)
    for order in ("ctx_first", "tree_first", "cycle"):
        ctx = Context(0)
        tree = Bvh.from_aabbs(aabbs, ctx)
        flat = tree.flatten()
        extra = _Hits(ctx)
        off, idx, _, _ = flat.traverse_batch(RayBatch(len(rays), np.float32, host=rays))
        assert ctx._nchildren == 3                      # the tree, its result object, the extra one
        if order == "ctx_first":
            ctx.close()                                 # deferred: the handle stays valid
            assert ctx._h is not None and ctx._deferred
            off2, idx2, _, _ = flat.traverse_batch(RayBatch(len(rays), np.float32, host=rays))
            assert np.array_equal(idx, idx2)
            extra.close(); tree.close()
            assert ctx._h is None                       # the last child destroyed it
        elif order == "tree_first":
            tree.close(); extra.close(); ctx.close()
            assert ctx._h is None
        else:
            cyc = {"ctx": ctx, "tree": tree, "flat": flat, "extra": extra}
            cyc["self"] = cyc                           # a cycle that owns everything
            del ctx, tree, flat, extra, cyc
            gc.collect()
    gc.collect()

