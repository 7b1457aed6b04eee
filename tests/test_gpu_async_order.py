"""an asynchronous batch must be replayed when the optimistic tree it walked turned out to be unfinished
(or must not be walked wide) — whoever finalizes the build first.  Round 2 decided that from the tree's state at the moment of
bvhgpu_hits_wait; a bvhgpu_tree_wait (or another result object's wait, a flatten, a rebuild) in between made the stale batch
look final.  Round 3: every result object records the tree's generation and whether it was still unfinalized."""
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
    x = dtype(1.004) ** np.arange(n, dtype=dtype)
    lo = np.stack([x, np.zeros(n, dtype), np.zeros(n, dtype)], axis=1)
    return np.concatenate([lo, lo + dtype(0.5)], axis=1).astype(dtype)


def _chain_rays(orc, m):
    o = np.zeros((m, 3), np.float32); o[:, 0] = -1; o[:, 1] = np.linspace(0.01, 0.49, m); o[:, 2] = 0.25
    d = np.tile(np.array([1, 0, 0], np.float32), (m, 1)); d[::3] = [1, 0.002, 0]
    return orc.make_rays(o, d)


def _setup(eng, orc, n=12000, m=20000):
    import torch
    from bvh_amd import Bvh, Context, RayBatch
    ctx = Context(0)
    chain = _chain(n)
    rays_np = _chain_rays(orc, m)
    rays_t = torch.from_numpy(rays_np.view(np.uint8).reshape(-1)).cuda()
    rays = RayBatch.from_device(rays_t, len(rays_np), np.float32)
    dev = torch.from_numpy(chain).cuda()
    ooff, oidx, _, _ = orc.traverse_flat(orc.flatten(orc.build(chain).nodes), chain, rays_np, threads=orc.max_threads())
    tree = Bvh.from_aabbs(_chain(100), ctx)      # no level hint for n shapes: the first asynchronous build takes the slow path
    return ctx, tree, dev, rays, rays_t, ooff, oidx


def test_tree_wait_before_hits_wait(eng, orc):
    from bvh_amd.api import _Hits
    ctx, tree, dev, rays, keep, ooff, oidx = _setup(eng, orc)
    hits = _Hits(ctx)
    tree.rebuild_async(dev)
    tree.traverse_async(rays, hits)
    tree.wait()                                   # finalizes the build on the slow path BEFORE the batch is waited for
    assert tree.build_levels >= 5
    st = hits.wait()
    off, idx = hits.fetch(rays.n)
    assert np.array_equal(off, ooff) and np.array_equal(idx, oidx) and st["hits"] == len(oidx)


def test_two_result_objects_on_one_pending_tree(eng, orc):
    from bvh_amd.api import _Hits
    ctx, tree, dev, rays, keep, ooff, oidx = _setup(eng, orc)
    h1, h2 = _Hits(ctx), _Hits(ctx)
    tree.rebuild_async(dev)
    tree.traverse_async(rays, h1)
    tree.traverse_async(rays, h2)
    for h in (h1, h2):                            # the first wait finalizes (slow path); the second must still replay
        st = h.wait()
        off, idx = h.fetch(rays.n)
        assert np.array_equal(off, ooff) and np.array_equal(idx, oidx) and st["hits"] == len(oidx)
    # a batch enqueued AFTER the finalize is final as it is
    tree.traverse_async(rays, h1)
    assert h1.wait()["hits"] == len(oidx)


def test_rebuild_and_destroy_complete_batches_in_flight(eng, orc):
    """The tree is rebuilt (then destroyed) while a batch on it has not been waited for: the batch is completed first — on the
    tree it was enqueued on — and its own wait returns that result."""
    import torch
    from bvh_amd import testbase as tb
    from bvh_amd.api import _Hits
    ctx, tree, dev, rays, keep, ooff, oidx = _setup(eng, orc)
    hits = _Hits(ctx)
    tree.rebuild_async(dev)
    tree.traverse_async(rays, hits)
    _, other = tb.create_n_cubes(500)
    tree.rebuild_async(torch.from_numpy(other).cuda())     # another scene into the same tree object
    st = hits.wait()
    off, idx = hits.fetch(rays.n)
    assert np.array_equal(off, ooff) and np.array_equal(idx, oidx) and st["hits"] == len(oidx)
    tree.rebuild_async(dev)
    tree.traverse_async(rays, hits)
    tree.close()                                           # bvhgpu_tree_destroy with a batch in flight
    st = hits.wait()
    off, idx = hits.fetch(rays.n)
    assert np.array_equal(off, ooff) and np.array_equal(idx, oidx)


@pytest.mark.parametrize("consumer", ["tree_wait", "rebuild_good", "other_result"])
def test_failed_build_reaches_every_batch_enqueued_on_it(eng, orc, consumer):
    """a NaN build's INVALID_ARG can be consumed by bvhgpu_tree_wait, by the rebuild that replaces the tree or by
    another result object's wait before THIS batch is waited for; its own wait must still return INVALID_ARG — what the synchronous
    call would have returned — never OK with lists from an unbuilt tree."""
    import torch
    from bvh_amd._lib import INVALID_ARG, BvhGpuError
    from bvh_amd.api import _Hits
    ctx, tree, dev, rays, keep, ooff, oidx = _setup(eng, orc, n=3000, m=5000)
    tree.rebuild_async(dev).wait()
    bad = _chain(3000); bad[1234, 4] = np.nan
    bad_dev = torch.from_numpy(bad).cuda()
    h1, h2 = _Hits(ctx), _Hits(ctx)
    tree.rebuild_async(bad_dev)
    tree.traverse_async(rays, h1)
    if consumer == "tree_wait":
        with pytest.raises(BvhGpuError) as e:
            tree.wait()
        assert e.value.status == INVALID_ARG
    elif consumer == "rebuild_good":
        tree.rebuild_async(dev)                   # swallows the bad generation's outcome: everything is rebuilt
    else:
        tree.traverse_async(rays, h2)
        with pytest.raises(BvhGpuError) as e:
            h2.wait()
        assert e.value.status == INVALID_ARG
    with pytest.raises(BvhGpuError) as e:
        h1.wait()
    assert e.value.status == INVALID_ARG and "bvh_node.rs:214-217" in str(e.value)
    # and everything is still usable
    tree.rebuild_async(dev)
    tree.traverse_async(rays, h1)
    assert h1.wait()["hits"] == len(oidx)
    off, idx = h1.fetch(rays.n)
    assert np.array_equal(off, ooff) and np.array_equal(idx, oidx)


def test_fetch_before_wait_is_refused(eng, orc):
    from bvh_amd._lib import INVALID_ARG, BvhGpuError
    from bvh_amd.api import _Hits
    ctx, tree, dev, rays, keep, ooff, oidx = _setup(eng, orc, n=3000)
    hits = _Hits(ctx)
    tree.rebuild_async(dev)
    tree.traverse_async(rays, hits)
    with pytest.raises(BvhGpuError) as e:
        hits.fetch(rays.n)
    assert e.value.status == INVALID_ARG
    hits.wait()
    hits.fetch(rays.n)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_exact_only_survives_scene_blob(eng, orc, dtype):
    """A tree with a split that had no SAH winner (empty child bounds, bvh_node.rs:225-230) must not be walked wide; the
    property now travels in the scene blob (round 2 dropped it: the importer walked wide and reported leaves the reference
    never reaches)."""
    import torch
    from bvh_amd import Bvh, Context, FlatBvh, RayBatch
    ctx = Context(0)
    rng = np.random.default_rng(5)
    big = dtype(1e19 if dtype == np.float32 else 1e154)
    lo = (rng.uniform(-1, 1, size=(500, 3)) * big).astype(dtype)
    far = np.concatenate([lo, lo + big * dtype(0.01)], axis=1)
    m = 30_000                                             # above the large-batch threshold: the wide walk would be chosen
    o = (rng.uniform(-1, 1, size=(m, 3)) * big).astype(dtype)
    d = rng.normal(size=(m, 3)).astype(dtype)
    rays_np = orc.make_rays(o, d, dtype)
    ooff, oidx, _, _ = orc.traverse_flat(orc.flatten(orc.build(far).nodes), far, rays_np, threads=orc.max_threads())
    assert len(oidx) > 0
    bvh = Bvh.from_aabbs(far, ctx)
    bvh.flatten_in_place()
    rb = RayBatch(m, dtype, host=rays_np)
    off, idx, _, _ = bvh.traverse_batch(rb)
    assert np.array_equal(off, ooff) and np.array_equal(idx, oidx)
    blob = torch.empty(bvh.scene_nbytes(), dtype=torch.uint8, device="cuda")
    bvh.scene_export(blob)
    imp = FlatBvh.scene_import(blob, blob.numel(), ctx)
    off2, idx2, _, _ = imp.traverse_batch(rb)
    assert np.array_equal(off2, ooff) and np.array_equal(idx2, oidx)
    host_blob = blob.cpu().numpy()
    imp2 = FlatBvh.scene_import(host_blob, len(host_blob), ctx)
    off3, idx3, _, _ = imp2.traverse_batch(rb)
    assert np.array_equal(off3, ooff) and np.array_equal(idx3, oidx)


@pytest.mark.parametrize("cubes,R,early", [(10_000, 1_000_000, 1), (10_000, 1_000_000, 0), (3000, 300_000, 1), (400, 100_000, 1)])
def test_early_item_filter_same_result(eng, orc, cubes, R, early):
    """BVHGPU_TRAVERSE_RAYS_READY on a tree that is being rebuilt: the wide walk's item filter runs on a side stream beside the build
    (k_wide_items) — same CSR as the oracle, step after step, whether the top of the tree qualifies (120 k / 36 k triangles) or
    the kernel hands the filter back to the walk (4 800 triangles: tree level 4 is below the level tier)."""
    import torch
    from bvh_amd import Bvh, Context, RayBatch, testbase as tb
    from bvh_amd._lib import RAY_F32, TRAVERSE_RAYS_READY, TUNE_WIDE_EARLY_ITEMS
    from bvh_amd.api import _Hits
    ctx = Context(0)
    ctx.set_tuning(TUNE_WIDE_EARLY_ITEMS, early)
    bounds = tb.default_bounds()
    _, aabbs_np = tb.create_n_cubes(cubes)
    aabbs = torch.from_numpy(aabbs_np).cuda()
    buf = torch.empty(R * RAY_F32.itemsize, dtype=torch.uint8, device="cuda")
    rays = RayBatch.generate(7, R, bounds, buf, np.float32, ctx)
    tree = Bvh.from_aabbs(aabbs, ctx)
    hits = _Hits(ctx)
    oflat = orc.flatten(orc.build(aabbs_np).nodes)
    ooff, oidx, _, _ = orc.traverse_flat(oflat, aabbs_np, orc.create_rays(7, R), threads=orc.max_threads())
    for step in range(4):
        tree.rebuild_async(aabbs)
        st = tree.traverse_async(rays, hits, flags=TRAVERSE_RAYS_READY).wait()
        off, idx = hits.fetch(R)
        assert np.array_equal(off, ooff) and np.array_equal(idx, oidx) and st["hits"] == len(oidx), step
    # another scene into the same objects (the filter must follow the NEW tree's top, not the previous build's records)
    _, other_np = tb.create_n_cubes(cubes, tb.default_bounds() * np.float32(0.5))
    other = torch.from_numpy(other_np).cuda()
    o2 = orc.flatten(orc.build(other_np).nodes)
    ooff2, oidx2, _, _ = orc.traverse_flat(o2, other_np, orc.create_rays(7, R), threads=orc.max_threads())
    for step in range(2):
        tree.rebuild_async(other)
        tree.traverse_async(rays, hits, flags=TRAVERSE_RAYS_READY).wait()
        off, idx = hits.fetch(R)
        assert np.array_equal(off, ooff2) and np.array_equal(idx, oidx2), step


@pytest.mark.parametrize("shift", [-1, 0, 2, 4, 5])
def test_staged_hit_output_growth_and_order(eng, orc, shift):
    """Whole-ray wide walk with staged output (round 3): the first 2^shift shapes of a ray go to its own slot, later ones through
    pool records — on a scene where EVERY ray hits 300 boxes in a row both halves, the growth of the index array (sized by the hit
    total, no longer by the pool) and of the pool, and the per-ray order must all come out right, first batch and replayed batch."""
    from bvh_amd import Bvh, Context, RayBatch
    from bvh_amd._lib import TUNE_WIDE_ITEMS_LOG4, TUNE_WIDE_STAGE_SHIFT
    ctx = Context(0)
    ctx.set_tuning(TUNE_WIDE_ITEMS_LOG4, 0)                # whole rays also for this small batch
    ctx.set_tuning(TUNE_WIDE_STAGE_SHIFT, shift)
    m = 300
    x = np.arange(m, dtype=np.float32) * np.float32(0.25)
    lo = np.stack([x, np.zeros(m, np.float32), np.zeros(m, np.float32)], axis=1)
    row = np.concatenate([lo, lo + np.float32(1.0)], axis=1)
    R = 30_000
    o = np.tile(np.array([-5, 0.5, 0.5], np.float32), (R, 1)); o[:, 1] += np.linspace(0, 0.4, R).astype(np.float32)
    d = np.tile(np.array([1, 0, 0], np.float32), (R, 1))
    d[::5] = [0, 1, 0]                                      # a fifth of the rays misses everything
    o[1::5, 0] = 40.0                                       # another fifth starts half-way: fewer hits
    rays = orc.make_rays(o, d)
    flat = Bvh.from_aabbs(row, ctx).flatten()
    ooff, oidx, _, _ = orc.traverse_flat(orc.flatten(orc.build(row).nodes), row, rays, threads=orc.max_threads())
    assert len(oidx) > 4_000_000
    rb = RayBatch(R, np.float32, host=rays)
    for _ in range(2):
        off, idx, _, _ = flat.traverse_batch(rb, coherent=True)   # (the default stages only batches flagged COHERENT)
        assert np.array_equal(off, ooff) and np.array_equal(idx, oidx)
