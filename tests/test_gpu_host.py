"""ABI 7 — the host-resident batch (bvhgpu_traverse_host_*: what GpuBvh::traverse_batch of the Rust shim calls).  Rays start in host memory
as origins + directions (Ray::new on the device, ray_impl.rs:70-80) or as Ray structs, the CSR ends in host memory; the batch is walked in
chunks on three streams.  The result must be the CSR of `for ray in rays { flat.traverse(&ray, shapes) }` (flat_bvh.rs:396-431) byte for
byte: against the oracle, and against the device-resident path of the same library, for every chunk count, for pinned and pageable
buffers, for a tree that is still building, for batches whose index lists outgrow the caller's buffer, in f32 and f64."""
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


def _od(orc, first, n, dtype, bounds=None):
    """origins and un-normalised directions of the create_ray stream (testbase.rs:687-691: ray k = Ray::new(draw 2k+1, draw 2k+2 as a
    vector); f64: the f32 points widened before Ray::new, as the engine defines the configs[4] stream) + the oracle's rays of the stream"""
    from bvh_amd import testbase as tb
    b = tb.default_bounds() if bounds is None else bounds
    k = np.arange(first, first + n, dtype=np.uint64)
    o = tb.next_point3_at(2 * k + 1, b).astype(dtype)
    d = tb.next_point3_at(2 * k + 2, b).astype(dtype)
    return np.ascontiguousarray(o), np.ascontiguousarray(d), orc.create_rays(first, n, b, dtype)


def _oracle_csr(orc, aabbs, rays):
    flat = orc.flatten(orc.build(aabbs).nodes)
    off, idx, _, _ = orc.traverse_flat(flat, aabbs, rays, threads=orc.max_threads())
    return off, idx


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("chunks,zero_copy", [(0, 3), (0, 0), (1, 3), (3, 1), (3, 2), (16, 3), (16, 0)])
def test_host_batch_equals_oracle_and_device_path(eng, orc, dtype, chunks, zero_copy):
    """zero_copy (BVHGPU_TUNE_HOST_ZERO_COPY): bit 0 the device reads the pinned ray arrays itself, bit 1 it writes offsets / indices itself;
    0 = copy engines both ways (what pageable buffers always get)"""
    from bvh_amd import Bvh, Context, HostStep, RayBatch, testbase as tb
    from bvh_amd._lib import TUNE_HOST_CHUNKS, TUNE_HOST_ZERO_COPY
    ctx = Context(0)
    ctx.set_tuning(TUNE_HOST_CHUNKS, chunks)
    ctx.set_tuning(TUNE_HOST_ZERO_COPY, zero_copy)
    _, aabbs = tb.create_n_cubes(3000)
    aabbs = aabbs.astype(dtype)
    n = 131072 * 3 + 17                                         # ragged: the last chunk is not a multiple of anything
    o, d, rays = _od(orc, 0, n, dtype)
    assert orc.make_rays(o[:5000], d[:5000], dtype).tobytes() == rays[:5000].tobytes()      # Ray::new of the inputs = the stream's rays
    ooff, oidx = _oracle_csr(orc, aabbs, rays)
    bvh = Bvh.from_aabbs(aabbs, ctx)
    bvh.flatten_in_place()
    hs = HostStep(bvh, len(aabbs), n, dtype)
    hs.aabbs[:] = aabbs; hs.origins[:] = o; hs.directions[:] = d
    for rep in range(3):                                        # first call: pools grow (replay path); then the steady state
        hs.offsets[:] = 0xDEADBEEF; hs.indices[:] = 0xDEADBEEF
        off, idx = hs.run(fused=rep != 1)
        assert np.array_equal(off, ooff) and np.array_equal(idx, oidx), (rep, chunks)
    # origin and direction side by side in one pinned array (BVHGPU_TRAVERSE_RAYS_OD6), one call and two
    hs6 = HostStep(bvh, len(aabbs), n, dtype, od6=True)
    hs6.aabbs[:] = aabbs; hs6.origins[:] = o; hs6.directions[:] = d
    for fused in (True, False):
        hs6.offsets[:] = 0; hs6.indices[:] = 0
        off, idx = hs6.run(fused=fused)
        assert np.array_equal(off, ooff) and np.array_equal(idx, oidx), (fused, chunks)
    od_page = np.ascontiguousarray(np.concatenate([o, d], axis=1))     # ... and out of pageable memory
    off6, idx6 = np.zeros(n + 1, np.uint32), np.zeros(max(len(oidx), 1), np.uint32)
    assert bvh.traverse_host(od_page, None, off6, idx6, od6=True) == len(oidx)
    assert np.array_equal(off6, ooff) and np.array_equal(idx6[:len(oidx)], oidx)
    hs6.close()
    # result arrays of two kinds: pinned offsets with pageable indices (the device writes the one, the download stream carries the other) and the reverse
    from bvh_amd.api import pinned_array
    for rep in range(2):
        p_off, g_idx = pinned_array(ctx, (n + 1,), np.uint32), np.zeros(max(len(oidx), 1), np.uint32)
        p_off[:] = 0
        assert bvh.traverse_host(hs.origins, hs.directions, p_off, g_idx) == len(oidx)
        assert np.array_equal(p_off, ooff) and np.array_equal(g_idx[:len(oidx)], oidx), rep
        g_off, p_idx = np.zeros(n + 1, np.uint32), pinned_array(ctx, (max(len(oidx), 1),), np.uint32)
        p_idx[:] = 0
        assert bvh.traverse_host(hs.origins, hs.directions, g_off, p_idx) == len(oidx)
        assert np.array_equal(g_off, ooff) and np.array_equal(p_idx[:len(oidx)], oidx), rep
    # the caller's own Ray structs out of pinned memory
    pr = pinned_array(ctx, (n,), rays.dtype)
    pr[:] = rays
    hs.offsets[:] = 0; hs.indices[:] = 0
    assert bvh.traverse_host(pr, None, hs.offsets, hs.indices) == len(oidx)
    assert np.array_equal(hs.offsets, ooff) and np.array_equal(hs.indices[:len(oidx)], oidx)
    # the device-resident path of the same library on Ray::new'd rays
    off2, idx2, _, _ = bvh.traverse_batch(RayBatch(n, dtype, host=rays))
    assert np.array_equal(off2, ooff) and np.array_equal(idx2, oidx)
    # pageable buffers, and the caller's own Ray structs (directions = NULL)
    off3, idx3 = np.zeros(n + 1, np.uint32), np.zeros(max(len(oidx), 1), np.uint32)
    bvh.rebuild(aabbs, flatten=True)
    assert bvh.traverse_host(o.copy(), d.copy(), off3, idx3) == len(oidx)
    assert np.array_equal(off3, ooff) and np.array_equal(idx3[:len(oidx)], oidx)
    off3[:] = 0; idx3[:] = 0
    assert bvh.traverse_host(rays, None, off3, idx3) == len(oidx)
    assert np.array_equal(off3, ooff) and np.array_equal(idx3[:len(oidx)], oidx)
    hs.close(); bvh.close()


def test_small_and_empty_batches(eng, orc):
    from bvh_amd import Bvh, Context, testbase as tb
    ctx = Context(0)
    _, aabbs = tb.create_n_cubes(500)
    bvh = Bvh.from_aabbs(aabbs, ctx); bvh.flatten_in_place()
    for n in (0, 1, 100, 5000, 20000):                         # below 16 384 rays the binary walk runs; above, the wide walk
        o, d, rays = _od(orc, 0, max(n, 1), np.float32)
        o, d, rays = o[:n], d[:n], rays[:n]
        off, idx = np.full(n + 1, 7, np.uint32), np.zeros(4096, np.uint32)
        total = bvh.traverse_host(np.ascontiguousarray(o), np.ascontiguousarray(d), off, idx)
        ooff, oidx = _oracle_csr(orc, aabbs, rays) if n else (np.zeros(1, np.uint32), np.zeros(0, np.uint32))
        assert total == len(oidx) and np.array_equal(off, ooff) and np.array_equal(idx[:total], oidx), n
    bvh.close()
    # an empty hierarchy (bvh_impl.rs:57-59: no nodes, every traversal returns nothing) and one with a single shape (flat_bvh.rs:129-141)
    o, d, rays = _od(orc, 0, 40_000, np.float32)
    for a in (np.zeros((0, 6), np.float32), aabbs[:1]):
        tree = Bvh.from_aabbs(a, ctx); tree.flatten_in_place()
        off, idx = np.full(len(o) + 1, 7, np.uint32), np.zeros(4096, np.uint32)
        total = tree.traverse_host(o, d, off, idx)
        if len(a):
            ooff, oidx = _oracle_csr(orc, a, rays)
        else:
            ooff, oidx = np.zeros(len(o) + 1, np.uint32), np.zeros(0, np.uint32)
        assert total == len(oidx) and np.array_equal(off, ooff) and np.array_equal(idx[:total], oidx), len(a)
        tree.close()


@pytest.mark.parametrize("pinned", [False, True])
def test_index_lists_larger_than_the_callers_buffer(eng, orc, pinned):
    """hit-heavy batch (the stand-in scene): total > indices_cap → offsets written, total returned, indices fetched afterwards without a
    second traversal (bvhgpu_traverse_host_indices)"""
    from bvh_amd import Bvh, Context, scene
    ctx = Context(0)
    _, aabbs, bounds = scene.parse_obj(scene.make_atrium_obj(4))
    n = 300_000
    o, d, rays = _od(orc, 0, n, np.float32, bounds)
    ooff, oidx = _oracle_csr(orc, aabbs, rays)
    assert len(oidx) > 2 * n
    bvh = Bvh.from_aabbs(aabbs, ctx); bvh.flatten_in_place()
    off, small = np.zeros(n + 1, np.uint32), np.zeros(1000, np.uint32)
    if pinned:      # (the device writes into the caller's arrays itself: what does not fit must not be written past the end)
        from bvh_amd.api import pinned_array
        off, blk = pinned_array(ctx, (n + 1,), np.uint32), pinned_array(ctx, (3000,), np.uint32)
        blk[:] = 0x5A5A5A5A
        small = blk[1000:2000]
        po, pd = pinned_array(ctx, (n, 3), np.float32), pinned_array(ctx, (n, 3), np.float32)
        po[:] = o; pd[:] = d
        o, d = po, pd
    total = bvh.traverse_host(o, d, off, small)
    assert total == len(oidx) and np.array_equal(off, ooff)
    if pinned:
        assert np.array_equal(small, oidx[:1000]) and (blk[:1000] == 0x5A5A5A5A).all() and (blk[2000:] == 0x5A5A5A5A).all()
    else:
        assert not small.any()
    with pytest.raises(eng.BvhGpuError):
        bvh.traverse_host_indices(small)
    idx = np.zeros(total, np.uint32)
    bvh.traverse_host_indices(idx)
    assert np.array_equal(idx, oidx)
    bvh.close()


def test_host_batch_on_a_tree_that_is_still_building(eng, orc):
    """rebuild_async from pinned AABBs, the batch enqueued at once: the ray upload runs beside the build; a build that fails (NaN input)
    is reported by the batch's call and leaves the ctx usable"""
    from bvh_amd import Bvh, Context, HostStep, testbase as tb
    from bvh_amd._lib import TUNE_FLATTEN_INLINE
    ctx = Context(0)
    ctx.set_tuning(TUNE_FLATTEN_INLINE, 1)   # (the wave tier flattens its own subtrees: a failed build runs none, the walk behind it must still find a consistent tree)
    _, a1 = tb.create_n_cubes(2000)
    _, a2 = tb.create_n_cubes(2500)
    a2 = a2[:len(a1)] * np.float32(0.5)
    n = 400_000
    o, d, rays = _od(orc, 0, n, np.float32)
    bvh = Bvh.from_aabbs(a1, ctx); bvh.flatten_in_place()
    hs = HostStep(bvh, len(a1), n, np.float32)
    hs.origins[:] = o; hs.directions[:] = d
    for a, fused in ((a1, True), (a2, False), (a1, True), (a2, False)):     # one call (upload enqueued before the build) / two calls
        hs.aabbs[:] = a
        off, idx = hs.run(fused=fused)
        ooff, oidx = _oracle_csr(orc, a, rays)
        assert np.array_equal(off, ooff) and np.array_equal(idx, oidx)
        assert bvh.nodes.tobytes() == orc.build(a).nodes.tobytes()
    hs.aabbs[:] = a1
    hs.aabbs[7, 2] = np.nan
    for fused in (True, False):
        with pytest.raises(eng.BvhGpuError):
            hs.run(fused=fused)
    hs.aabbs[:] = a2
    off, idx = hs.run()
    ooff, oidx = _oracle_csr(orc, a2, rays)
    assert np.array_equal(off, ooff) and np.array_equal(idx, oidx)
    hs.close(); bvh.close()


def test_registered_caller_memory(eng, orc):
    """bvhgpu_host_register: memory the caller already owns (ordinary numpy arrays here, a Vec's allocation in Rust) announced once — the batch then
    runs at pinned speed and the device writes offsets / indices straight into those arrays.  Same CSR as ever."""
    import ctypes as C
    from bvh_amd import Bvh, Context, _lib, testbase as tb
    ctx = Context(0)
    lib = _lib.load()
    _, aabbs = tb.create_n_cubes(3000)
    n = 300_000
    o, d, rays = _od(orc, 0, n, np.float32)
    ooff, oidx = _oracle_csr(orc, aabbs, rays)
    od = np.ascontiguousarray(np.concatenate([o, d], axis=1))
    off, idx = np.full(n + 1, 0xABABABAB, np.uint32), np.full(max(len(oidx), 1) + 100, 0xABABABAB, np.uint32)
    regs = [od, off, idx]
    for a in regs:
        assert lib.bvhgpu_host_register(ctx._h, a.ctypes.data_as(C.c_void_p), a.nbytes) == 0
    try:
        bvh = Bvh.from_aabbs(aabbs, ctx); bvh.flatten_in_place()
        for _ in range(2):
            off[:] = 0xABABABAB; idx[:] = 0xABABABAB
            assert bvh.traverse_host(od, None, off, idx, od6=True) == len(oidx)
            assert np.array_equal(off, ooff) and np.array_equal(idx[:len(oidx)], oidx) and (idx[len(oidx):] == 0xABABABAB).all()
        bvh.close()
    finally:
        for a in regs:
            assert lib.bvhgpu_host_unregister(ctx._h, a.ctypes.data_as(C.c_void_p)) == 0


def test_pinned_memory_entry_points(eng):
    import ctypes as C
    from bvh_amd import Context, _lib
    from bvh_amd.api import pinned_array
    ctx = Context(0)
    a = pinned_array(ctx, (1000, 6), np.float32)
    a[:] = 3.5
    assert a.shape == (1000, 6) and float(a.sum()) == 3.5 * 6000 and a._bvhgpu_pinned
    lib = _lib.load()
    buf = np.zeros(1 << 20, np.uint8)
    assert lib.bvhgpu_host_register(ctx._h, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
    assert lib.bvhgpu_host_unregister(ctx._h, buf.ctypes.data_as(C.c_void_p)) == 0
    assert lib.bvhgpu_host_register(ctx._h, None, 16) != 0 and lib.bvhgpu_host_free(ctx._h, None) == 0
    del a
