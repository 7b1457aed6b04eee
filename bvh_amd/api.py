"""Host-side mirror of the reference's operator surface for the hot path, over the C ABI.

Names, argument meaning and error behaviour follow the Rust crate (paths under /root/reference/src):
  Bounded.aabb()                     aabb/aabb_impl.rs:28-56
  BHShape.set_bh_node_index / bh_node_index   bounding_hierarchy.rs:53-65
  BoundingHierarchy.build / build_par / build_with_executor / traverse   bounding_hierarchy.rs:89-336
  Bvh, Bvh.flatten, FlatBvh          bvh/bvh_impl.rs:27-119, flat_bvh.rs:240-431
  Ray.new / intersects_aabb / intersection_slice_for_aabb   ray/ray_impl.rs:70-145
Conventions kept: empty input → empty structure, no error (bvh_impl.rs:57-59); `traverse` returns the
hit shapes themselves, in the reference's order; shapes are only borrowed.
Everything that computes runs on the GPU through libbvh_mi355x.so; this file only marshals.
"""
from __future__ import annotations

import atexit
import ctypes as C
import weakref
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import (DEVICE, F32, F64, FLAT_F32, FLAT_F64, HOST, NODE_F32, NODE_F64, NONE, RAY_F32, RAY_F64,
                   TRAVERSE_CLOSEST, TRAVERSE_COHERENT, TRAVERSE_FARTHEST_FIRST, TRAVERSE_NEAREST_FIRST, TRAVERSE_STATS, TRAVERSE_T_SLICE, TRAVERSE_TRIANGLES, BvhGpuError, check, ptr)


def _sfx(dtype) -> str:
    dt = np.dtype(dtype)
    if dt == np.float32:
        return "f32"
    if dt == np.float64:
        return "f64"
    raise TypeError(f"BHValue must be f32 or f64, got {dt}")


def _ray_dtype(s):
    return RAY_F32 if s == "f32" else RAY_F64


# Native handles must be released while the HIP runtime is still alive: at interpreter shutdown the
# runtime's own static destructors may already have run when Python finalises leftover objects.
_live = weakref.WeakSet()
_closing = False


def _shutdown():
    global _closing
    for kind in ("_Hits", "tree", "Context"):
        for obj in list(_live):
            k = type(obj).__name__
            if (kind == "tree" and k not in ("_Hits", "Context")) or k == kind:
                try:
                    obj.close()
                except Exception:
                    pass
    _closing = True


atexit.register(_shutdown)


def _is_device_tensor(x) -> bool:
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda") and bool(x.is_cuda)


# ------------------------------------------------------------------------------------------------
class Context:
    """One GPU + one HIP stream + scratch (bvhgpu_ctx).  `stream` may be a raw hipStream_t
    (e.g. torch.cuda.current_stream().cuda_stream) so that work is ordered with the caller's."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        lib = _lib.load()
        if _lib.device_count() <= 0:
            raise BvhGpuError(_lib.NO_DEVICE, "no MI355X / HIP device visible: the engine has no CPU fallback")
        h = C.c_void_p()
        check(lib.bvhgpu_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h = h
        self.device = int(device)
        # The C ABI wants a ctx to outlive what was made on it (bvhgpu_tree_destroy / bvhgpu_hits_destroy / bvhgpu_comm_destroy look at
        # their ctx).  Reference counts normally give that order, but objects caught in a reference cycle (a pytest.raises traceback is
        # enough) are finalised by the cyclic collector in ARBITRARY order — so the ctx counts its children and, if it is closed while
        # some are alive, hands its destruction to the last of them.
        self._nchildren = 0
        self._deferred = False
        _live.add(self)

    def _child_add(self):
        self._nchildren += 1

    def _child_drop(self):
        self._nchildren -= 1
        if self._nchildren <= 0 and self._deferred:
            self._destroy()

    def _destroy(self):
        if getattr(self, "_h", None) and not _closing:
            _lib.load().bvhgpu_destroy(self._h)
        self._h = None
        self._deferred = False

    def synchronize(self):
        check(_lib.load().bvhgpu_synchronize(self._h), self._h)

    def enable_timing(self, on: bool = True):
        check(_lib.load().bvhgpu_enable_timing(self._h, int(on)), self._h)

    def last_timings(self) -> dict:
        t = _lib.Timings()
        check(_lib.load().bvhgpu_last_timings(self._h, C.byref(t)), self._h)
        return dict(build_ms=t.build_ms, flatten_ms=t.flatten_ms, traverse_kernel_ms=t.traverse_kernel_ms,
                    traverse_total_ms=t.traverse_total_ms)

    def set_tuning(self, knob: int, value: int):
        """performance knobs (include/bvh_mi355x.h bvhgpu_tune); results never change."""
        check(_lib.load().bvhgpu_set_tuning(self._h, int(knob), int(value)), self._h)

    def get_tuning(self, knob: int) -> int:
        v = C.c_int(0)
        check(_lib.load().bvhgpu_get_tuning(self._h, int(knob), C.byref(v)), self._h)
        return int(v.value)

    def close(self):
        if getattr(self, "_h", None) is None:
            return
        if getattr(self, "_nchildren", 0) > 0 and not _closing:
            self._deferred = True      # the last tree / result object / communicator made on it destroys it
            return
        self._destroy()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: Optional[Context] = None


class PinnedArray(np.ndarray):
    """numpy view of pinned host memory (bvhgpu_host_alloc): the DMA engines read and write it directly"""
    _bvhgpu_pinned = True


def pinned_array(ctx: "Context", shape, dtype) -> np.ndarray:
    """an uninitialised array in pinned host memory; freed (bvhgpu_host_free) when the last view of it is gone"""
    lib = _lib.load()
    dt = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dt.itemsize
    p = C.c_void_p()
    check(lib.bvhgpu_host_alloc(ctx._h, max(nbytes, 1), C.byref(p)), ctx._h)
    buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
    ctx._child_add()

    def release(addr=p.value, c=ctx):
        if not _closing and getattr(c, "_h", None):
            lib.bvhgpu_host_free(c._h, C.c_void_p(addr))
        c._child_drop()
    weakref.finalize(buf, release)
    a = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape).view(PinnedArray)
    return a


class HostStep:
    """What a host-resident caller of the drop-in boundary does per frame, with everything in pinned memory: shapes' AABBs, ray origins
    and directions in; CSR offsets / indices out.  run() = GpuBvh::build + traverse_batch of the Rust shim in one call
    (bvhgpu_build_traverse_host_*): the ray upload overlaps the build, Ray::new runs on the device."""

    def __init__(self, bvh: "Bvh", n_shapes: int, n_rays: int, dtype=np.float32, index_cap: Optional[int] = None, od6: bool = False):
        """od6: origin and direction of a ray side by side in ONE pinned array (`od`, n x 6; `origins` / `directions` are views of its
        halves) — BVHGPU_TRAVERSE_RAYS_OD6: one transfer per chunk instead of two"""
        self.bvh, ctx = bvh, bvh.ctx
        self.aabbs = pinned_array(ctx, (n_shapes, 6), dtype)
        self.od = pinned_array(ctx, (n_rays, 6), dtype) if od6 else None
        self.origins = self.od[:, :3] if od6 else pinned_array(ctx, (n_rays, 3), dtype)
        self.directions = self.od[:, 3:] if od6 else pinned_array(ctx, (n_rays, 3), dtype)
        self.offsets = pinned_array(ctx, (n_rays + 1,), np.uint32)
        self.indices = pinned_array(ctx, (index_cap or max(n_rays, 1 << 16),), np.uint32)
        self.total = 0

    def run(self, coherent: bool = False, fused: bool = True):
        """fused: bvhgpu_build_traverse_host_* (one call, the ray upload enqueued before the build); else the two calls
        bvhgpu_rebuild_flat_async_*(HOST) + bvhgpu_traverse_host_*"""
        if not fused:
            self.bvh.rebuild_async(self.aabbs)
        if self.od is not None:
            self.total = self.bvh.traverse_host(self.od, None, self.offsets, self.indices, coherent=coherent, aabbs=self.aabbs if fused else None,
                                                od6=True)
        else:
            self.total = self.bvh.traverse_host(self.origins, self.directions, self.offsets, self.indices, coherent=coherent,
                                                aabbs=self.aabbs if fused else None)
        if self.total > self.indices.size:
            self.indices = pinned_array(self.bvh.ctx, (self.total + self.total // 8,), np.uint32)
            self.bvh.traverse_host_indices(self.indices)
        return self.offsets, self.indices[:self.total]

    def close(self):
        self.aabbs = self.origins = self.directions = self.od = self.offsets = self.indices = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


# ------------------------------------------------------------------------------------------------
class Aabb:
    """Aabb<T,3> {min, max} (aabb_impl.rs:10-16) as a host value type.  Only construction helpers
    live here (a shape needs them to answer Bounded.aabb()); all hot-path arithmetic is on the GPU."""
    __slots__ = ("min", "max")

    def __init__(self, mn, mx, dtype=np.float32):
        self.min = np.asarray(mn, dtype=dtype).reshape(3)
        self.max = np.asarray(mx, dtype=dtype).reshape(3)

    @staticmethod
    def with_bounds(mn, mx, dtype=np.float32) -> "Aabb":  # aabb_impl.rs:97-99
        return Aabb(mn, mx, dtype)

    @staticmethod
    def empty(dtype=np.float32) -> "Aabb":  # aabb_impl.rs:119-124
        return Aabb([np.inf] * 3, [-np.inf] * 3, dtype)

    def join(self, other: "Aabb") -> "Aabb":  # aabb_impl.rs:303-308
        return Aabb(np.minimum(self.min, other.min), np.maximum(self.max, other.max), self.min.dtype)

    def grow(self, p) -> "Aabb":  # aabb_impl.rs:375-380
        p = np.asarray(p, dtype=self.min.dtype)
        return Aabb(np.minimum(self.min, p), np.maximum(self.max, p), self.min.dtype)

    def as6(self) -> np.ndarray:
        return np.concatenate([self.min, self.max])

    def __repr__(self):
        return f"Aabb(min={self.min.tolist()}, max={self.max.tolist()})"


class Bounded:
    """trait Bounded (aabb_impl.rs:28-56)."""

    def aabb(self) -> Aabb:
        raise NotImplementedError


class BHShape(Bounded):
    """trait BHShape (bounding_hierarchy.rs:53-65)."""

    def set_bh_node_index(self, index: int) -> None:
        raise NotImplementedError

    def bh_node_index(self) -> int:
        raise NotImplementedError


# ------------------------------------------------------------------------------------------------
class RayBatch:
    """A batch of Ray<T,3> (ray_impl.rs:17-29) in host memory (structured array) or in HBM."""

    def __init__(self, n: int, dtype, host: Optional[np.ndarray] = None, device=None, device_ptr: int = 0):
        self.n = int(n)
        self.sfx = _sfx(dtype)
        self.host = host          # numpy structured array or None
        self.device = device      # object that owns the device memory (e.g. torch tensor) or None
        self.device_ptr = int(device_ptr)

    @property
    def mem(self) -> int:
        return DEVICE if self.host is None else HOST

    def _ptr(self):
        return ptr(self.device_ptr) if self.host is None else ptr(self.host)

    @staticmethod
    def new(origins, directions, dtype=np.float32, ctx: Optional[Context] = None) -> "RayBatch":
        """Ray::new for every row (ray_impl.rs:70-80): normalise the direction, cache 1/d — on the GPU."""
        ctx = ctx or default_context()
        s = _sfx(dtype)
        o = np.ascontiguousarray(origins, dtype=dtype).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=dtype).reshape(-1, 3)
        if len(o) != len(d):
            raise ValueError("origins and directions differ in length")
        out = np.zeros(len(o), dtype=_ray_dtype(s))
        fn = getattr(_lib.load(), f"bvhgpu_rays_new_{s}")
        check(fn(ctx._h, ptr(o), ptr(d), len(o), HOST, ptr(out), HOST), ctx._h)
        return RayBatch(len(o), dtype, host=out)

    @staticmethod
    def from_device(device_obj, n: int, dtype=np.float32) -> "RayBatch":
        """Wrap rays already resident in HBM (torch uint8/float tensor holding n Ray structs)."""
        return RayBatch(n, dtype, host=None, device=device_obj, device_ptr=device_obj.data_ptr())

    @staticmethod
    def generate(first: int, n: int, bounds, device_obj, dtype=np.float32, ctx: Optional[Context] = None) -> "RayBatch":
        """The bench ray stream create_ray(seed 0) (testbase.rs:687-691, :825), rays [first, first+n),
        written straight into `device_obj` (a torch tensor of >= n*sizeof(Ray) bytes on the GPU)."""
        ctx = ctx or default_context()
        s = _sfx(dtype)
        b = np.ascontiguousarray(bounds, dtype=np.float32).reshape(6)
        fn = getattr(_lib.load(), f"bvhgpu_gen_rays_{s}")
        check(fn(ctx._h, C.c_uint64(first), n, ptr(b), ptr(device_obj.data_ptr())), ctx._h)
        return RayBatch(n, dtype, host=None, device=device_obj, device_ptr=device_obj.data_ptr())


def intersect_triangle_pairs(rays: "RayBatch", tris, ctx: Optional["Context"] = None) -> np.ndarray:
    """Ray::intersects_triangle (ray_impl.rs:154-213) for ray i against triangle i, on the GPU.
    tris: (n,3,3) or (n,9).  returns (n,3) = Intersection{distance,u,v}."""
    ctx = ctx or default_context()
    ft = np.float32 if rays.sfx == "f32" else np.float64
    t = np.ascontiguousarray(tris, dtype=ft).reshape(-1, 9)
    if len(t) != rays.n or rays.host is None:
        raise ValueError("one host triangle per host ray is required")
    out = np.zeros((rays.n, 3), dtype=ft)
    fn = getattr(_lib.load(), f"bvhgpu_ray_triangle_pairs_{rays.sfx}")
    check(fn(ctx._h, rays._ptr(), ptr(t), rays.n, HOST, ptr(out)), ctx._h)
    return out


def camera(eye, look_at, up=(0.0, 1.0, 0.0), fov_y_deg: float = 60.0, aspect: float = 1.6) -> np.ndarray:
    """cam[14] for RayBatch.primary: eye, right, up, forward (orthonormal, f32), tan_x, tan_y."""
    e = np.asarray(eye, dtype=np.float64); f = np.asarray(look_at, dtype=np.float64) - e
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, dtype=np.float64)); r /= np.linalg.norm(r)
    u = np.cross(r, f)
    ty = np.tan(np.radians(fov_y_deg) / 2)
    return np.concatenate([e, r, u, f, [ty * aspect, ty]]).astype(np.float32)


def _primary(cam, width, height, first, n, device_obj, dtype, ctx):
    s = _sfx(dtype)
    c = np.ascontiguousarray(cam, dtype=np.float32).reshape(14)
    fn = getattr(_lib.load(), f"bvhgpu_gen_primary_rays_{s}")
    check(fn(ctx._h, ptr(c), width, height, C.c_uint64(first), n, ptr(device_obj.data_ptr())), ctx._h)
    return RayBatch(n, dtype, host=None, device=device_obj, device_ptr=device_obj.data_ptr())


RayBatch.primary = staticmethod(lambda cam, width, height, first, n, device_obj, dtype=np.float32, ctx=None:
                               _primary(cam, width, height, first, n, device_obj, dtype, ctx or default_context()))


class Ray:
    """struct Ray (ray_impl.rs:17-29).  Ray(origin, direction) == Ray::new."""

    def __init__(self, origin, direction, dtype=np.float32, ctx: Optional[Context] = None):
        self._batch = RayBatch.new([origin], [direction], dtype, ctx)
        r = self._batch.host[0]
        self.origin, self.direction, self.inv_direction = r["o"].copy(), r["d"].copy(), r["inv"].copy()
        self.dtype = np.dtype(dtype)

    @staticmethod
    def new(origin, direction, dtype=np.float32) -> "Ray":
        return Ray(origin, direction, dtype)

    def _probe(self, aabb: Aabb, want_t: bool):
        # one-box scene: a single-shape Bvh traversed by this ray is exactly one
        # Ray::intersects_aabb (flat_bvh.rs:411-418 / bvh_node.rs:314)
        bvh = Bvh.from_aabbs(aabb.as6().reshape(1, 6).astype(self.dtype))
        off, idx, ts, _ = bvh.flatten().traverse_batch(self._batch, want_t=want_t)
        return (len(idx) == 1), (ts[0] if want_t and len(idx) else None)

    def intersects_triangle(self, a, b, c):  # ray_impl.rs:154-213 → Intersection(distance, u, v)
        t = np.concatenate([np.asarray(v, dtype=self.dtype).reshape(3) for v in (a, b, c)]).reshape(1, 9)
        r = intersect_triangle_pairs(self._batch, t)[0]
        return r[0], r[1], r[2]

    def intersects_aabb(self, aabb: Aabb) -> bool:  # ray_impl.rs:105-110, intersect_default.rs:16-37
        return self._probe(aabb, False)[0]

    def intersection_slice_for_aabb(self, aabb: Aabb):  # ray_impl.rs:118-145
        hit, ts = self._probe(aabb, True)
        return (ts[0], ts[1]) if hit else None


# ------------------------------------------------------------------------------------------------
class _Hits:
    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.h = C.c_void_p()
        self._counted = True
        ctx._child_add()
        _live.add(self)

    def wait(self) -> dict:
        """bvhgpu_hits_wait: complete an asynchronous batch; returns the stats dict of traverse_batch."""
        lib = _lib.load()
        check(lib.bvhgpu_hits_wait(self.h), self.ctx._h)
        total = C.c_uint64()
        st = _lib.TraverseStats()
        check(lib.bvhgpu_hits_info(self.h, None, C.byref(total), C.byref(st)), self.ctx._h)
        return dict(hits=int(st.hits), visited=int(st.visited), leaf_visits=int(st.leaf_visits),
                    device_steps=int(st.device_steps), wave_steps=int(st.wave_steps), total=int(total.value))

    def info(self) -> dict:
        """stats of the completed batch (bvhgpu_hits_info)"""
        lib = _lib.load()
        total = C.c_uint64()
        st = _lib.TraverseStats()
        check(lib.bvhgpu_hits_info(self.h, None, C.byref(total), C.byref(st)), self.ctx._h)
        return dict(hits=int(st.hits), visited=int(st.visited), leaf_visits=int(st.leaf_visits),
                    device_steps=int(st.device_steps), wave_steps=int(st.wave_steps), total=int(total.value))

    def walk_kernel(self) -> str:
        """bvhgpu_hits_walk_kernel: the kernel the completed batch was walked by, spelled as rocprofv3 prints it"""
        buf = C.create_string_buffer(160)
        check(_lib.load().bvhgpu_hits_walk_kernel(self.h, buf, 160), self.ctx._h)
        return buf.value.decode()

    def walk_flags(self) -> int:
        f = C.c_uint(0)
        check(_lib.load().bvhgpu_hits_walk_info(self.h, C.byref(f)), self.ctx._h)
        return int(f.value)

    def fetch_closest(self, n_rays: int, dtype=np.float32):
        """CLOSEST batches: (Intersection{distance,u,v}[n,3], shape[n]) of the completed batch, copied to the host"""
        isect = np.zeros((n_rays, 3), dtype=dtype)
        shape = np.zeros(n_rays, dtype=np.uint32)
        check(_lib.load().bvhgpu_hits_fetch_closest(self.h, ptr(isect), ptr(shape), HOST), self.ctx._h)
        return isect, shape

    def fetch_triangles(self, dtype=np.float32):
        """TRIANGLES batches: Intersection{distance,u,v} of every candidate, CSR order"""
        total = C.c_uint64()
        check(_lib.load().bvhgpu_hits_info(self.h, None, C.byref(total), None), self.ctx._h)
        isect = np.zeros((total.value, 3), dtype=dtype)
        check(_lib.load().bvhgpu_hits_fetch_triangles(self.h, ptr(isect), HOST), self.ctx._h)
        return isect

    def fetch(self, n_rays: int):
        """(offsets, indices) of the completed batch, copied to the host."""
        lib = _lib.load()
        total = C.c_uint64()
        check(lib.bvhgpu_hits_info(self.h, None, C.byref(total), None), self.ctx._h)
        offsets = np.zeros(n_rays + 1, dtype=np.uint32)
        indices = np.zeros(total.value, dtype=np.uint32)
        check(lib.bvhgpu_hits_fetch(self.h, ptr(offsets), ptr(indices), None, HOST), self.ctx._h)
        return offsets, indices

    def close(self):
        if getattr(self, "h", None) and not _closing:
            _lib.load().bvhgpu_hits_destroy(self.h)
        self.h = None
        if getattr(self, "_counted", False):
            self._counted = False
            self.ctx._child_drop()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _TreeBase:
    def __init__(self, ctx: Context, handle, sfx: str):
        self.ctx = ctx
        self._t = handle
        self.sfx = sfx
        self._hits = _Hits(ctx)
        self._counted = True
        ctx._child_add()
        _live.add(self)

    def close(self):
        if getattr(self, "_t", None):
            if getattr(self, "_hits", None) is not None:
                self._hits.close()
            if not _closing:
                _lib.load().bvhgpu_tree_destroy(self._t)
        self._t = None
        if getattr(self, "_counted", False):
            self._counted = False
            self.ctx._child_drop()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> Tuple[int, int, int]:
        n, nn, nf = C.c_size_t(), C.c_size_t(), C.c_size_t()
        dt = C.c_int()
        check(_lib.load().bvhgpu_tree_info(self._t, C.byref(dt), C.byref(n), C.byref(nn), C.byref(nf)), self.ctx._h)
        return int(n.value), int(nn.value), int(nf.value)

    # ---- traversal -------------------------------------------------------------------------
    def traverse_batch(self, rays: RayBatch, want_t: bool = False, stats: bool = False, fetch: bool = True,
                       coherent: bool = False, order: Optional[str] = None):
        """<FlatBvh as BoundingHierarchy>::traverse for a batch (flat_bvh.rs:396-431).
        returns (offsets[n+1], indices[total], tslice[total,2]|None, stats dict)."""
        if rays.sfx != self.sfx:
            raise BvhGpuError(_lib.DTYPE_MISMATCH, "ray dtype differs from tree dtype")
        lib = _lib.load()
        flags = (TRAVERSE_T_SLICE if want_t else 0) | (TRAVERSE_STATS if stats else 0) | (TRAVERSE_COHERENT if coherent else 0)
        flags |= _lib.ORDER_FLAGS[order]
        fn = getattr(lib, f"bvhgpu_traverse_{self.sfx}")
        check(fn(self._t, rays._ptr(), rays.n, rays.mem, flags, C.byref(self._hits.h)), self.ctx._h)
        total = C.c_uint64()
        st = _lib.TraverseStats()
        check(lib.bvhgpu_hits_info(self._hits.h, None, C.byref(total), C.byref(st)), self.ctx._h)
        wk = C.c_uint()
        check(lib.bvhgpu_hits_walk_info(self._hits.h, C.byref(wk)), self.ctx._h)
        sd = dict(hits=int(st.hits), visited=int(st.visited), leaf_visits=int(st.leaf_visits),
                  device_steps=int(st.device_steps), wave_steps=int(st.wave_steps), walk=int(wk.value),   # walk: _lib.WALK_* bits (diagnostic)
                  kernel=self._hits.walk_kernel())                                                          # ... and the kernel's name
        if not fetch:
            return None, None, None, sd
        ft = np.float32 if self.sfx == "f32" else np.float64
        offsets = np.zeros(rays.n + 1, dtype=np.uint32)
        indices = np.zeros(total.value, dtype=np.uint32)
        ts = np.zeros((total.value, 2), dtype=ft) if want_t else None
        check(lib.bvhgpu_hits_fetch(self._hits.h, ptr(offsets), ptr(indices), ptr(ts), HOST), self.ctx._h)
        return offsets, indices, ts, sd

    def traverse_async(self, rays: RayBatch, hits: Optional["_Hits"] = None, flags: int = 0) -> "_Hits":
        """bvhgpu_traverse_async_*: enqueue the batch (rays resident in HBM) and return at once; the tree may still be
        building on the same stream.  `hits.wait()` completes it (and returns the stats dict)."""
        if rays.sfx != self.sfx:
            raise BvhGpuError(_lib.DTYPE_MISMATCH, "ray dtype differs from tree dtype")
        hits = hits or self._hits
        fn = getattr(_lib.load(), f"bvhgpu_traverse_async_{self.sfx}")
        check(fn(self._t, rays._ptr(), rays.n, rays.mem, flags, C.byref(hits.h)), self.ctx._h)
        return hits

    # ---- triangle stage -------------------------------------------------------------------
    def set_triangles(self, tris) -> None:
        """vertices of the shapes, (n,3,3) or (n,9) [a, b, c] (testbase.rs Triangle :316-323); numpy or torch GPU tensor."""
        lib = _lib.load()
        fn = getattr(lib, f"bvhgpu_tree_set_triangles_{self.sfx}")
        if _is_device_tensor(tris):
            check(fn(self._t, ptr(tris.data_ptr()), tris.numel() // 9, DEVICE), self.ctx._h)
        else:
            ft = np.float32 if self.sfx == "f32" else np.float64
            a = np.ascontiguousarray(tris, dtype=ft).reshape(-1, 9)
            check(fn(self._t, ptr(a), len(a), HOST), self.ctx._h)

    def intersect_triangles(self, rays: RayBatch, stats: bool = False, coherent: bool = False, order: Optional[str] = None):
        """traverse + Ray::intersects_triangle on every returned shape (testbase.rs:826-836).
        returns (offsets, indices, isect[total,3] = Intersection{distance,u,v}, stats)"""
        lib = _lib.load()
        flags = TRAVERSE_TRIANGLES | (TRAVERSE_STATS if stats else 0) | (TRAVERSE_COHERENT if coherent else 0)
        flags |= _lib.ORDER_FLAGS[order]
        check(getattr(lib, f"bvhgpu_traverse_{self.sfx}")(self._t, rays._ptr(), rays.n, rays.mem, flags,
                                                           C.byref(self._hits.h)), self.ctx._h)
        total = C.c_uint64()
        st = _lib.TraverseStats()
        check(lib.bvhgpu_hits_info(self._hits.h, None, C.byref(total), C.byref(st)), self.ctx._h)
        ft = np.float32 if self.sfx == "f32" else np.float64
        offsets = np.zeros(rays.n + 1, dtype=np.uint32)
        indices = np.zeros(total.value, dtype=np.uint32)
        isect = np.zeros((total.value, 3), dtype=ft)
        check(lib.bvhgpu_hits_fetch(self._hits.h, ptr(offsets), ptr(indices), None, HOST), self.ctx._h)
        check(lib.bvhgpu_hits_fetch_triangles(self._hits.h, ptr(isect), HOST), self.ctx._h)
        sd = dict(hits=int(st.hits), visited=int(st.visited), leaf_visits=int(st.leaf_visits),
                  device_steps=int(st.device_steps), wave_steps=int(st.wave_steps))
        return offsets, indices, isect, sd

    def closest_hits(self, rays: RayBatch, stats: bool = False, fetch: bool = True, coherent: bool = False,
                     order: Optional[str] = None):
        """triangle stage fused into the walk: per ray the nearest Intersection and its shape
        (distance +inf / shape NONE when nothing is hit).  returns (isect[n,3], shape[n], stats)"""
        lib = _lib.load()
        flags = TRAVERSE_CLOSEST | (TRAVERSE_STATS if stats else 0) | (TRAVERSE_COHERENT if coherent else 0)
        flags |= _lib.ORDER_FLAGS[order]
        check(getattr(lib, f"bvhgpu_traverse_{self.sfx}")(self._t, rays._ptr(), rays.n, rays.mem, flags,
                                                           C.byref(self._hits.h)), self.ctx._h)
        st = _lib.TraverseStats()
        check(lib.bvhgpu_hits_info(self._hits.h, None, None, C.byref(st)), self.ctx._h)
        sd = dict(hits=int(st.hits), visited=int(st.visited), leaf_visits=int(st.leaf_visits),
                  device_steps=int(st.device_steps), wave_steps=int(st.wave_steps))
        if not fetch:
            return None, None, sd
        ft = np.float32 if self.sfx == "f32" else np.float64
        isect = np.zeros((rays.n, 3), dtype=ft)
        shape = np.zeros(rays.n, dtype=np.uint32)
        check(lib.bvhgpu_hits_fetch_closest(self._hits.h, ptr(isect), ptr(shape), HOST), self.ctx._h)
        return isect, shape, sd

    # ---- point query ---------------------------------------------------------------------
    def nearest_batch(self, points, triangles: bool = False):
        """<FlatBvh as BoundingHierarchy>::nearest_to (flat_bvh.rs:513-562) for many points.  Shape distance: the
        shape's own AABB (UnitBox, testbase.rs:101-105) or, with triangles=True, the closest point on the triangle
        (testbase.rs:436-443; needs set_triangles).  returns (shape[n] u32 — NONE for an empty hierarchy, dist[n])"""
        ft = np.float32 if self.sfx == "f32" else np.float64
        p = np.ascontiguousarray(points, dtype=ft).reshape(-1, 3)
        shape = np.zeros(len(p), dtype=np.uint32)
        dist = np.zeros(len(p), dtype=ft)
        fn = getattr(_lib.load(), f"bvhgpu_nearest_{self.sfx}")
        check(fn(self._t, ptr(p), len(p), HOST, 1 if triangles else 0, ptr(shape), ptr(dist)), self.ctx._h)
        return shape, dist

    def nearest_to(self, query, shapes: Sequence, triangles: bool = False):
        """BoundingHierarchy::nearest_to (bounding_hierarchy.rs:262-336): Option<(&Shape, distance)>."""
        s, d = self.nearest_batch([query], triangles)
        return None if s[0] == NONE else (shapes[int(s[0])], d[0])

    def nearest_traverse(self, ray: "Ray", shapes: Sequence) -> List:
        """Bvh::nearest_traverse_iterator (bvh_impl.rs:145-151; DistanceTraverseIterator) collected into a list."""
        _, idx, _, _ = self.traverse_batch(ray._batch, order="nearest_heap")
        return [shapes[int(i)] for i in idx]

    def farthest_traverse(self, ray: "Ray", shapes: Sequence) -> List:
        """Bvh::farthest_traverse_iterator (bvh_impl.rs:162-176) collected into a list."""
        _, idx, _, _ = self.traverse_batch(ray._batch, order="farthest_heap")
        return [shapes[int(i)] for i in idx]

    def nearest_child_traverse(self, ray: "Ray", shapes: Sequence) -> List:
        """Bvh::nearest_child_traverse_iterator (bvh_impl.rs:184-190) collected into a list."""
        _, idx, _, _ = self.traverse_batch(ray._batch, order="nearest")
        return [shapes[i] for i in idx.tolist()]

    def farthest_child_traverse(self, ray: "Ray", shapes: Sequence) -> List:
        """Bvh::farthest_child_traverse_iterator (bvh_impl.rs:206-212) collected into a list."""
        _, idx, _, _ = self.traverse_batch(ray._batch, order="farthest")
        return [shapes[i] for i in idx.tolist()]

    def hits_device(self) -> Tuple[int, int]:
        """device addresses of the last result's (offsets, indices) — valid until the next traverse."""
        o, i, t = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(_lib.load().bvhgpu_hits_device(self._hits.h, C.byref(o), C.byref(i), C.byref(t)), self.ctx._h)
        return int(o.value or 0), int(i.value or 0)

    def traverse(self, query, shapes: Sequence) -> List:
        """BoundingHierarchy::traverse (bounding_hierarchy.rs:246-250): the shapes whose AABB `query`
        (a Ray) intersects, in the reference's order."""
        if not isinstance(query, Ray):
            raise NotImplementedError("the MI355X engine implements traverse for Ray queries "
                                      "(IntersectsAabb for Aabb/Point/Ball are out of scope, SURVEY §2)")
        _, idx, _, _ = self.traverse_batch(query._batch)
        return [shapes[i] for i in idx.tolist()]

    # ---- multi-GPU scene transport ---------------------------------------------------------
    def scene_nbytes(self) -> int:
        n = C.c_size_t()
        check(_lib.load().bvhgpu_scene_nbytes(self._t, C.byref(n)), self.ctx._h)
        return int(n.value)

    def scene_export(self, dst) -> None:
        """dst: torch uint8 tensor on the GPU (DEVICE) or a numpy uint8 array (HOST)."""
        if _is_device_tensor(dst):
            check(_lib.load().bvhgpu_scene_export(self._t, ptr(dst.data_ptr()), DEVICE), self.ctx._h)
        else:
            check(_lib.load().bvhgpu_scene_export(self._t, ptr(dst), HOST), self.ctx._h)


class FlatBvh(_TreeBase):
    """type FlatBvh = Vec<FlatNode> (flat_bvh.rs:254) resident in HBM."""

    @staticmethod
    def build(shapes: Sequence, dtype=np.float32, ctx: Optional[Context] = None) -> "FlatBvh":
        return Bvh.build(shapes, dtype, ctx).flatten()  # flat_bvh.rs:328-331

    @staticmethod
    def build_par(shapes: Sequence, dtype=np.float32, ctx: Optional[Context] = None) -> "FlatBvh":
        return Bvh.build_par(shapes, dtype, ctx).flatten()

    @property
    def nodes(self) -> np.ndarray:
        """the FlatNode array in the reference's layout (flat_bvh.rs:17-46)."""
        _, _, nf = self.info()
        out = np.zeros(nf, dtype=FLAT_F32 if self.sfx == "f32" else FLAT_F64)
        check(_lib.load().bvhgpu_flat_nodes(self._t, ptr(out), HOST), self.ctx._h)
        return out

    def __len__(self):
        return self.info()[2]

    @staticmethod
    def from_flat_nodes(flat: np.ndarray, shape_aabbs: np.ndarray, ctx: Optional[Context] = None) -> "FlatBvh":
        """Upload a FlatBvh produced elsewhere (e.g. by the Rust crate via flatten_custom)."""
        ctx = ctx or default_context()
        s = "f32" if flat.dtype == FLAT_F32 else "f64"
        ft = np.float32 if s == "f32" else np.float64
        sa = np.ascontiguousarray(shape_aabbs, dtype=ft).reshape(-1, 6)
        flat = np.ascontiguousarray(flat)
        h = C.c_void_p()
        fn = getattr(_lib.load(), f"bvhgpu_tree_from_flat_{s}")
        check(fn(ctx._h, ptr(flat), len(flat), ptr(sa), len(sa), C.byref(h)), ctx._h)
        return FlatBvh(ctx, h, s)

    @staticmethod
    def scene_import(src, nbytes: int, ctx: Optional[Context] = None, reuse: Optional["FlatBvh"] = None) -> "FlatBvh":
        """Import a scene blob (scene_export) — used on the peers after the RCCL broadcast.
        `reuse`: a FlatBvh from an earlier scene_import whose HBM is overwritten in place."""
        ctx = ctx or default_context()
        h = reuse._t if reuse is not None else C.c_void_p()
        if _is_device_tensor(src):
            check(_lib.load().bvhgpu_scene_import(ctx._h, ptr(src.data_ptr()), nbytes, DEVICE, C.byref(h)), ctx._h)
        else:
            check(_lib.load().bvhgpu_scene_import(ctx._h, ptr(src), nbytes, HOST, C.byref(h)), ctx._h)
        dt = C.c_int()
        check(_lib.load().bvhgpu_tree_info(h, C.byref(dt), None, None, None), ctx._h)
        if reuse is not None:
            reuse.sfx = "f32" if dt.value == F32 else "f64"
            return reuse
        return FlatBvh(ctx, h, "f32" if dt.value == F32 else "f64")


class Bvh(_TreeBase):
    """struct Bvh {nodes: Vec<BvhNode>} (bvh_impl.rs:27-33), built and resident on the GPU."""

    # ---- construction ----------------------------------------------------------------------
    @staticmethod
    def from_aabbs(aabbs, ctx: Optional[Context] = None) -> "Bvh":
        """Build from an (n,6) array [min xyz, max xyz] (numpy → uploaded; torch GPU tensor → used in place)."""
        ctx = ctx or default_context()
        lib = _lib.load()
        h = C.c_void_p()
        if _is_device_tensor(aabbs):
            import torch  # only needed when the caller already uses torch
            s = "f32" if aabbs.dtype == torch.float32 else "f64"
            n = aabbs.numel() // 6
            check(getattr(lib, f"bvhgpu_build_{s}")(ctx._h, ptr(aabbs.data_ptr()), n, DEVICE, C.byref(h)), ctx._h)
        else:
            a = np.asarray(aabbs)
            if a.dtype not in (np.float32, np.float64):
                a = a.astype(np.float32)
            s = _sfx(a.dtype)
            a = np.ascontiguousarray(a).reshape(-1, 6)
            check(getattr(lib, f"bvhgpu_build_{s}")(ctx._h, ptr(a), len(a), HOST, C.byref(h)), ctx._h)
        return Bvh(ctx, h, s)

    def rebuild(self, aabbs, flatten: bool = False) -> "Bvh":
        """Bvh::build again into the same device buffers (no allocation when n fits).  flatten=True: FlatBvh::build
        (flat_bvh.rs:328-331) — build + flatten in one call."""
        lib = _lib.load()
        fn = getattr(lib, f"bvhgpu_rebuild_flat_{self.sfx}" if flatten else f"bvhgpu_rebuild_{self.sfx}")
        if _is_device_tensor(aabbs):
            n = aabbs.numel() // 6
            check(fn(self._t, ptr(aabbs.data_ptr()), n, DEVICE), self.ctx._h)
        else:
            ft = np.float32 if self.sfx == "f32" else np.float64
            a = np.ascontiguousarray(aabbs, dtype=ft).reshape(-1, 6)
            check(fn(self._t, ptr(a), len(a), HOST), self.ctx._h)
        return self

    def rebuild_async(self, aabbs) -> "Bvh":
        """bvhgpu_rebuild_flat_async_*: FlatBvh::build enqueued on the context's stream, no wait (`aabbs`: a GPU tensor, or a
        PinnedArray — either stays valid until wait()).  Everything that looks at the tree afterwards completes the build first."""
        fn = getattr(_lib.load(), f"bvhgpu_rebuild_flat_async_{self.sfx}")
        if isinstance(aabbs, np.ndarray) and getattr(aabbs, "_bvhgpu_pinned", False):
            check(fn(self._t, ptr(aabbs), aabbs.size // 6, HOST), self.ctx._h)
            return self
        if not _is_device_tensor(aabbs):
            raise BvhGpuError(_lib.INVALID_ARG, "rebuild_async takes shape AABBs that are resident in HBM (or in pinned host memory: pinned_array)")
        check(fn(self._t, ptr(aabbs.data_ptr()), aabbs.numel() // 6, DEVICE), self.ctx._h)
        return self

    def traverse_host(self, origins, directions, offsets: np.ndarray, indices: np.ndarray, coherent: bool = False, aabbs=None,
                      od6: bool = False) -> int:
        """bvhgpu_traverse_host_*: `for (o, d) in rays { flat.traverse(&Ray::new(o, d), shapes) }` for rays in host memory (n x 3 each;
        directions None: `origins` is an array of Ray structs), CSR written into the caller's `offsets` (n + 1) / `indices`; returns the
        hit total (indices are written when they fit: fetch with traverse_host_indices otherwise).  The tree may still be building.
        aabbs (host array, n x 6): bvhgpu_build_traverse_host_* — the tree is rebuilt from them first, underneath the ray upload.
        od6: `origins` is ONE array n x 6 = [o xyz, d xyz] per ray (BVHGPU_TRAVERSE_RAYS_OD6: one transfer per chunk), directions None."""
        lib = _lib.load()
        ft = np.float32 if self.sfx == "f32" else np.float64
        if od6:
            assert directions is None and origins.dtype == ft and origins.flags.c_contiguous and origins.size % 6 == 0
            n = origins.size // 6
        elif directions is None:
            assert origins.dtype == _ray_dtype(self.sfx) and origins.flags.c_contiguous
            n = len(origins)
        else:
            assert origins.dtype == ft and directions.dtype == ft and origins.flags.c_contiguous and directions.flags.c_contiguous
            n = origins.size // 3
            assert directions.size == 3 * n
        assert offsets.dtype == np.uint32 and offsets.size >= n + 1 and indices.dtype == np.uint32
        total = C.c_uint64(0)
        tail = (ptr(origins), ptr(directions) if directions is not None else None, n,
                (TRAVERSE_COHERENT if coherent else 0) | (_lib.TRAVERSE_RAYS_OD6 if od6 else 0),
                ptr(offsets), ptr(indices), indices.size, C.byref(total))
        if aabbs is not None:
            assert aabbs.dtype == ft and aabbs.flags.c_contiguous
            check(getattr(lib, f"bvhgpu_build_traverse_host_{self.sfx}")(self._t, ptr(aabbs), aabbs.size // 6, *tail), self.ctx._h)
        else:
            check(getattr(lib, f"bvhgpu_traverse_host_{self.sfx}")(self._t, *tail), self.ctx._h)
        return int(total.value)

    def traverse_host_indices(self, indices: np.ndarray) -> None:
        check(_lib.load().bvhgpu_traverse_host_indices(self.ctx._h, ptr(indices), indices.size), self.ctx._h)

    def wait(self) -> "Bvh":
        check(_lib.load().bvhgpu_tree_wait(self._t), self.ctx._h)
        return self

    def refit(self, aabbs) -> "Bvh":
        """The shapes moved: same topology, every child AABB recomputed from the new shape AABBs
        (Bvh::fix_aabbs_ascending, optimization.rs:355-391, applied to the whole tree); a flattened tree is
        re-flattened.  ~10x cheaper than rebuild(); update_shapes' re-insertion (optimization.rs:337-352) is not
        reproduced — rebuild() when the topology should follow the motion."""
        lib = _lib.load()
        fn = getattr(lib, f"bvhgpu_refit_{self.sfx}")
        if _is_device_tensor(aabbs):
            check(fn(self._t, ptr(aabbs.data_ptr()), aabbs.numel() // 6, DEVICE), self.ctx._h)
        else:
            ft = np.float32 if self.sfx == "f32" else np.float64
            a = np.ascontiguousarray(aabbs, dtype=ft).reshape(-1, 6)
            check(fn(self._t, ptr(a), len(a), HOST), self.ctx._h)
        return self

    @staticmethod
    def build(shapes: Sequence, dtype=np.float32, ctx: Optional[Context] = None) -> "Bvh":
        """Bvh::build (bvh_impl.rs:40-45).  Calls shape.aabb() ONCE per shape (the reference calls it
        once per level, only observable for impure aabb()), builds on the GPU, then calls
        shape.set_bh_node_index(leaf) for every shape (bvh_node.rs:102)."""
        n = len(shapes)
        a = np.empty((n, 6), dtype=dtype)
        for i, s in enumerate(shapes):
            a[i] = s.aabb().as6()
        bvh = Bvh.from_aabbs(a, ctx)
        if n:
            for s, ni in zip(shapes, bvh.shape_nodes.tolist()):
                s.set_bh_node_index(ni)
        return bvh

    @staticmethod
    def build_par(shapes: Sequence, dtype=np.float32, ctx: Optional[Context] = None) -> "Bvh":
        """BoundingHierarchy::build_par (bounding_hierarchy.rs:170-177): same tree; the GPU build is
        the parallel build."""
        return Bvh.build(shapes, dtype, ctx)

    @staticmethod
    def build_with_executor(shapes: Sequence, executor=None, dtype=np.float32, ctx: Optional[Context] = None) -> "Bvh":
        """bvh_impl.rs:53-96.  The executor only chooses how sub-builds are scheduled on the CPU and
        cannot change the result (node placement is arithmetic); the GPU schedules its own."""
        return Bvh.build(shapes, dtype, ctx)

    # ---- inspection ------------------------------------------------------------------------
    @property
    def nodes(self) -> np.ndarray:
        """Vec<BvhNode> as a structured array (include/bvh_mi355x.h bvhgpu_node_*)."""
        _, nn, _ = self.info()
        out = np.zeros(nn, dtype=NODE_F32 if self.sfx == "f32" else NODE_F64)
        check(_lib.load().bvhgpu_tree_nodes(self._t, ptr(out), HOST), self.ctx._h)
        return out

    @property
    def shape_nodes(self) -> np.ndarray:
        n, _, _ = self.info()
        out = np.zeros(n, dtype=np.uint32)
        check(_lib.load().bvhgpu_tree_shape_nodes(self._t, ptr(out), HOST), self.ctx._h)
        return out

    @property
    def build_levels(self) -> int:
        lv = C.c_int()
        check(_lib.load().bvhgpu_tree_build_levels(self._t, C.byref(lv)), self.ctx._h)
        return int(lv.value)

    # ---- flatten ---------------------------------------------------------------------------
    def flatten(self) -> "FlatBvh":
        """Bvh::flatten (flat_bvh.rs:312-319).  The flat arrays live in the same device object."""
        check(_lib.load().bvhgpu_flatten(self._t), self.ctx._h)
        f = FlatBvh.__new__(FlatBvh)
        f.ctx, f._t, f.sfx, f._hits = self.ctx, self._t, self.sfx, self._hits
        f._owner = self          # shares the handle; the Bvh keeps ownership
        f.__class__ = _FlatView
        return f

    def flatten_in_place(self) -> None:
        check(_lib.load().bvhgpu_flatten(self._t), self.ctx._h)

    def traverse(self, query, shapes: Sequence) -> List:
        """Bvh::traverse (bvh_impl.rs:104-119): same hit list, same order as the flat traversal
        (bvh_node.rs:288-319 visits the same boxes in the same order)."""
        self.flatten_in_place()
        return super().traverse(query, shapes)

    def nearest_batch(self, points, triangles: bool = False):
        """Bvh::nearest_to answered by the flat loop (flat_bvh.rs:513-562): the distance is the same as the
        recursive form's (bvh_node.rs:327-374); on exact ties the two reference forms may name different shapes."""
        self.flatten_in_place()
        return super().nearest_batch(points, triangles)


class _FlatView(FlatBvh):
    """FlatBvh that borrows a Bvh's device object (so flatten() does not copy the tree)."""

    def close(self):  # the owning Bvh destroys the handle
        self._t = None

    def __del__(self):
        self._t = None
