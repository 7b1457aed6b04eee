"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from the product package bvh_amd/.
Parity pin status: see the header of oracle/bvh_oracle.h ("partially pinned").
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

NONE = 0xFFFFFFFF

NODE_F32 = np.dtype([("l_min", "<f4", 3), ("l_max", "<f4", 3), ("r_min", "<f4", 3), ("r_max", "<f4", 3),
                     ("parent", "<u4"), ("l", "<u4"), ("r", "<u4"), ("shape", "<u4")])
NODE_F64 = np.dtype([("l_min", "<f8", 3), ("l_max", "<f8", 3), ("r_min", "<f8", 3), ("r_max", "<f8", 3),
                     ("parent", "<u4"), ("l", "<u4"), ("r", "<u4"), ("shape", "<u4")])
FLAT_F32 = np.dtype([("min", "<f4", 3), ("max", "<f4", 3), ("entry", "<u4"), ("exit", "<u4"), ("shape", "<u4")])
FLAT_F64 = np.dtype([("min", "<f8", 3), ("max", "<f8", 3), ("entry", "<u4"), ("exit", "<u4"), ("shape", "<u4"),
                     ("_pad", "<u4")])
RAY_F32 = np.dtype([("o", "<f4", 3), ("d", "<f4", 3), ("inv", "<f4", 3)])
RAY_F64 = np.dtype([("o", "<f8", 3), ("d", "<f8", 3), ("inv", "<f8", 3)])
assert NODE_F32.itemsize == 64 and NODE_F64.itemsize == 112
assert FLAT_F32.itemsize == 36 and FLAT_F64.itemsize == 64
assert RAY_F32.itemsize == 36 and RAY_F64.itemsize == 72

DEFAULT_BOUNDS = np.array([-100000.0] * 3 + [100000.0] * 3, dtype=np.float32)  # testbase.rs:598-603


class TravStats(C.Structure):
    _fields_ = [("visited", C.c_uint64), ("leaf_visits", C.c_uint64), ("hits", C.c_uint64),
                ("max_visited", C.c_uint64)]


def build_library(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("bvh_oracle.c", "oracle_impl.inc", "bvh_oracle.h")]
    stale = force or not os.path.exists(_SO) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if stale and all(os.path.exists(s) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None
_native = False


def _load(path):
    l = C.CDLL(path)
    l.orc_splitmix64.restype = C.c_uint64
    l.orc_max_threads.restype = C.c_int
    for s, ct in (("f32", C.c_float), ("f64", C.c_double)):
        getattr(l, f"orc_surface_area_{s}").restype = ct
        getattr(l, f"orc_ray_triangle_{s}").restype = ct
        getattr(l, f"orc_flatten_{s}").restype = C.c_size_t
        getattr(l, f"orc_traverse_flat_{s}").restype = C.c_uint64
        getattr(l, f"orc_traverse_tree_{s}").restype = C.c_uint64
        getattr(l, f"orc_traverse_flat_once_{s}").restype = C.c_uint64
    return l


def lib():
    global _lib
    if _lib is None:
        build_library()
        _lib = _load(_SO)
    return _lib


def use_native() -> bool:
    """bench.py's cpu_baseline leg only: rebuild the oracle ON THIS BOX with -O3 -march=native (still
    -ffp-contract=off, so results do not change) and route every call of this module through it.  False (and the portable
    -O2 build stays in use) if the box has no compiler."""
    global _lib, _native
    so = os.path.join(_HERE, "liboracle_native.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_native.so"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
        _lib = _load(so)
        _native = True
    except Exception:
        _native = False
    return _native


def is_native() -> bool:
    return _native


def library_path() -> str:
    """the shared object every call of this module goes through right now (the portable -O2 build, or the native one after use_native())"""
    lib()
    return os.path.join(_HERE, "liboracle_native.so") if _native else _SO


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _sfx(dtype):
    return "f32" if np.dtype(dtype) == np.float32 else "f64"


def _types(s):
    return (np.float32, NODE_F32, FLAT_F32, RAY_F32) if s == "f32" else (np.float64, NODE_F64, FLAT_F64, RAY_F64)


def max_threads() -> int:
    return int(lib().orc_max_threads())


# ---------------------------------------------------------------- generators
def create_n_cubes(n_cubes: int, bounds=DEFAULT_BOUNDS):
    tris = np.empty((n_cubes * 12, 3, 3), dtype=np.float32)
    aabbs = np.empty((n_cubes * 12, 6), dtype=np.float32)
    b = np.ascontiguousarray(bounds, dtype=np.float32)
    lib().orc_create_n_cubes(C.c_size_t(n_cubes), _p(b), _p(tris), _p(aabbs))
    return tris, aabbs


def create_rays(first: int, n: int, bounds=DEFAULT_BOUNDS, dtype=np.float32):
    """create_ray stream (testbase.rs:687-691); dtype f64: the same f32 points widened before Ray::new (the engine's
    definition of the configs[4] stream, bvhgpu_gen_rays_f64)"""
    b = np.ascontiguousarray(bounds, dtype=np.float32)
    if np.dtype(dtype) == np.float64:
        rays = np.empty(n, dtype=RAY_F64)
        lib().orc_create_rays_f64(C.c_uint64(first), C.c_size_t(n), _p(b), _p(rays))
        return rays
    rays = np.empty(n, dtype=RAY_F32)
    lib().orc_create_rays(C.c_uint64(first), C.c_size_t(n), _p(b), _p(rays))
    return rays


def primary_rays(cam, width: int, height: int, first: int, n: int, dtype=np.float32):
    """coherent primary rays (bvh_oracle.c orc_primary_rays): cam = eye3, right3, up3, forward3, tan_x, tan_y"""
    c = np.ascontiguousarray(cam, dtype=np.float32).reshape(14)
    f64 = np.dtype(dtype) == np.float64
    rays = np.zeros(n, dtype=RAY_F64 if f64 else RAY_F32)
    fn = lib().orc_primary_rays_f64 if f64 else lib().orc_primary_rays
    fn(_p(c), C.c_uint32(width), C.c_uint32(height), C.c_uint64(first), C.c_size_t(n), _p(rays))
    return rays


def aligned_boxes():
    a = np.empty((21, 6), dtype=np.float32)
    lib().orc_aligned_boxes(_p(a))
    return a


def make_rays(origins, dirs, dtype=np.float32):
    s = _sfx(dtype)
    ft, _, _, rt = _types(s)
    o = np.ascontiguousarray(origins, dtype=ft).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, dtype=ft).reshape(-1, 3)
    rays = np.empty(len(o), dtype=rt)
    f = getattr(lib(), f"orc_ray_new_{s}")
    for i in range(len(o)):
        f(_p(o[i]), _p(d[i]), C.c_void_p(rays.ctypes.data + i * rt.itemsize))
    return rays


# ---------------------------------------------------------------- primitives
def ray_intersects_aabb(ray, box) -> bool:
    s = "f32" if ray.dtype == RAY_F32 else "f64"
    ft = _types(s)[0]
    b = np.ascontiguousarray(box, dtype=ft).reshape(6)
    r = np.ascontiguousarray(ray).reshape(1)
    return bool(getattr(lib(), f"orc_ray_intersects_aabb_{s}")(_p(r), _p(b)))


def ray_slice(ray, box):
    s = "f32" if ray.dtype == RAY_F32 else "f64"
    ft = _types(s)[0]
    b = np.ascontiguousarray(box, dtype=ft).reshape(6)
    r = np.ascontiguousarray(ray).reshape(1)
    out = np.zeros(2, dtype=ft)
    ok = getattr(lib(), f"orc_ray_slice_{s}")(_p(r), _p(b), _p(out))
    return (out[0], out[1]) if ok else None


def ray_triangle(ray, a, b, c):
    s = "f32" if ray.dtype == RAY_F32 else "f64"
    ft = _types(s)[0]
    r = np.ascontiguousarray(ray).reshape(1)
    a, b, c = (np.ascontiguousarray(v, dtype=ft).reshape(3) for v in (a, b, c))
    uv = np.zeros(2, dtype=ft)
    d = getattr(lib(), f"orc_ray_triangle_{s}")(_p(r), _p(a), _p(b), _p(c), _p(uv))
    return ft(d), uv[0], uv[1]


def surface_area(box, dtype=np.float32):
    s = _sfx(dtype)
    b = np.ascontiguousarray(box, dtype=dtype).reshape(6)
    return getattr(lib(), f"orc_surface_area_{s}")(_p(b))


def center(box, dtype=np.float32):
    s = _sfx(dtype)
    b = np.ascontiguousarray(box, dtype=dtype).reshape(6)
    out = np.zeros(3, dtype=dtype)
    getattr(lib(), f"orc_center_{s}")(_p(b), _p(out))
    return out


def largest_axis(box, dtype=np.float32):
    s = _sfx(dtype)
    b = np.ascontiguousarray(box, dtype=dtype).reshape(6)
    return int(getattr(lib(), f"orc_largest_axis_{s}")(_p(b)))


# ---------------------------------------------------------------- build / flatten / traverse
@dataclass
class Tree:
    nodes: np.ndarray
    shape_node: np.ndarray


def build(aabbs, parallel: bool = False, threads: int = 0, schedule: str = "tasks") -> Tree:
    """threads > 0: a parallel build on a team of that size — schedule "tasks": the rayon_executor restatement (OpenMP task
    recursion, bvh_impl.rs:527-543); "fast": the big nodes split by the whole team, then one parallel for over the subtrees
    (oracle_impl.inc build_fast: the same arithmetic per node, byte-equal arrays, scales with the cores)"""
    s = _sfx(aabbs.dtype)
    ft, nt, _, _ = _types(s)
    a = np.ascontiguousarray(aabbs, dtype=ft).reshape(-1, 6)
    n = len(a)
    nodes = np.zeros(max(2 * n - 1, 0), dtype=nt)
    shape_node = np.zeros(n, dtype=np.uint32)
    if threads > 0:
        fn = getattr(lib(), f"orc_build_fast_{s}" if schedule == "fast" else f"orc_build_threads_{s}")
        rc = fn(_p(a), C.c_size_t(n), _p(nodes), _p(shape_node), C.c_int(threads))
    else:
        fn = getattr(lib(), f"orc_build_par_{s}" if parallel else f"orc_build_{s}")
        rc = fn(_p(a), C.c_size_t(n), _p(nodes), _p(shape_node))
    if rc != 0:
        raise MemoryError("oracle build failed")
    return Tree(nodes, shape_node)


def flatten(nodes) -> np.ndarray:
    s = "f32" if nodes.dtype == NODE_F32 else "f64"
    _, _, flt, _ = _types(s)
    n_nodes = len(nodes)
    n = (n_nodes + 1) // 2
    cap = 3 * n - 2 if n >= 2 else n
    out = np.zeros(max(cap, 0), dtype=flt)
    got = getattr(lib(), f"orc_flatten_{s}")(_p(nodes), C.c_size_t(n_nodes), _p(out))
    assert got == len(out), (got, len(out))
    return out


def traverse_flat(flat, shape_aabbs, rays, want_t: bool = False, threads: int = 1):
    """returns (offsets[r+1], indices, tslice|None, stats dict)"""
    s = "f32" if flat.dtype == FLAT_F32 else "f64"
    ft = _types(s)[0]
    sa = np.ascontiguousarray(shape_aabbs, dtype=ft).reshape(-1, 6)
    rays = np.ascontiguousarray(rays)
    nr = len(rays)
    offsets = np.zeros(nr + 1, dtype=np.uint32)
    st = TravStats()
    fn = getattr(lib(), f"orc_traverse_flat_{s}")
    total = fn(_p(flat), C.c_size_t(len(flat)), _p(sa), _p(rays), C.c_size_t(nr), _p(offsets), None,
               C.c_uint64(0), None, C.byref(st), C.c_int(threads))
    indices = np.zeros(total, dtype=np.uint32)
    ts = np.zeros((total, 2), dtype=ft) if want_t else None
    fn(_p(flat), C.c_size_t(len(flat)), _p(sa), _p(rays), C.c_size_t(nr), _p(offsets), _p(indices),
       C.c_uint64(total), _p(ts), C.byref(st), C.c_int(threads))
    stats = dict(visited=st.visited, leaf_visits=st.leaf_visits, hits=st.hits, max_visited=st.max_visited)
    return offsets, indices, ts, stats


def traverse_flat_once(flat, shape_aabbs, rays, threads: int = 1):
    """the reference harness' loop (testbase.rs:826-836): ONE walk per ray into a growable per-ray list, lists dropped; returns
    (total hits, checksum of the shape indices).  bench.py's cpu_baseline times this; the CSR form above walks twice."""
    s = "f32" if flat.dtype == FLAT_F32 else "f64"
    ft = _types(s)[0]
    sa = np.ascontiguousarray(shape_aabbs, dtype=ft).reshape(-1, 6)
    rays = np.ascontiguousarray(rays)
    ck = C.c_uint64(0)
    total = getattr(lib(), f"orc_traverse_flat_once_{s}")(_p(flat), C.c_size_t(len(flat)), _p(sa), _p(rays), C.c_size_t(len(rays)),
                                                          C.c_int(threads), C.byref(ck))
    return int(total), int(ck.value)


def harness_loop(flat, shape_aabbs, tris, first: int, n_rays: int, bounds=DEFAULT_BOUNDS, cam=None, width: int = 0, height: int = 0,
                 threads: int = 1):
    """intersect_bh (testbase.rs:819-837) whole, f32: create_ray (or a primary ray of `cam`) → FlatBvh::traverse into a growable list →
    Ray::intersects_triangle on every candidate; rays [first, first + n) of the stream, rays-parallel.  returns (candidates, checksum)"""
    sa = np.ascontiguousarray(shape_aabbs, dtype=np.float32).reshape(-1, 6)
    t = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)
    b = np.ascontiguousarray(bounds, dtype=np.float32).reshape(6)
    c = None if cam is None else np.ascontiguousarray(cam, dtype=np.float32).reshape(14)
    ck = C.c_uint64(0)
    fn = lib().orc_harness_loop_f32
    fn.restype = C.c_uint64
    total = fn(_p(np.ascontiguousarray(flat)), C.c_size_t(len(flat)), _p(sa), _p(t), C.c_uint64(first), C.c_size_t(n_rays), _p(b), _p(c),
               C.c_uint32(width), C.c_uint32(height), C.c_int(threads), C.byref(ck))
    return int(total), int(ck.value)


def traverse_tree(nodes, shape_aabbs, rays):
    s = "f32" if nodes.dtype == NODE_F32 else "f64"
    ft = _types(s)[0]
    sa = np.ascontiguousarray(shape_aabbs, dtype=ft).reshape(-1, 6)
    rays = np.ascontiguousarray(rays)
    nr = len(rays)
    offsets = np.zeros(nr + 1, dtype=np.uint32)
    fn = getattr(lib(), f"orc_traverse_tree_{s}")
    total = fn(_p(nodes), C.c_size_t(len(nodes)), _p(sa), _p(rays), C.c_size_t(nr), _p(offsets), None, C.c_uint64(0))
    indices = np.zeros(total, dtype=np.uint32)
    fn(_p(nodes), C.c_size_t(len(nodes)), _p(sa), _p(rays), C.c_size_t(nr), _p(offsets), _p(indices),
       C.c_uint64(total))
    return offsets, indices


def triangle_stage(tris, rays, offsets, indices):
    """testbase.rs:826-836 after traversal: Intersection{distance,u,v} of every candidate (CSR order) and
    the closest candidate per ray.  tris: (n,3,3) or (n,9).  returns (isect[total,3], closest[r,3], prim[r])"""
    s = "f32" if rays.dtype == RAY_F32 else "f64"
    ft = _types(s)[0]
    t = np.ascontiguousarray(tris, dtype=ft).reshape(-1, 9)
    rays = np.ascontiguousarray(rays)
    off = np.ascontiguousarray(offsets, dtype=np.uint32)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    isect = np.zeros((len(idx), 3), dtype=ft)
    closest = np.zeros((len(rays), 3), dtype=ft)
    prim = np.zeros(len(rays), dtype=np.uint32)
    fn = getattr(lib(), f"orc_triangle_stage_{s}")
    fn.restype = None
    fn(_p(t), _p(rays), C.c_size_t(len(rays)), _p(off), _p(idx), _p(isect), _p(closest), _p(prim))
    return isect, closest, prim


def aabb_min_dist2(box, p, dtype=np.float32):
    s = _sfx(dtype)
    fn = getattr(lib(), f"orc_aabb_min_dist2_{s}")
    fn.restype = C.c_float if s == "f32" else C.c_double
    return dtype(fn(_p(np.ascontiguousarray(box, dtype=dtype).reshape(6)), _p(np.ascontiguousarray(p, dtype=dtype).reshape(3))))


def triangle_dist2(tri, p, dtype=np.float32):
    s = _sfx(dtype)
    fn = getattr(lib(), f"orc_triangle_dist2_{s}")
    fn.restype = C.c_float if s == "f32" else C.c_double
    return dtype(fn(_p(np.ascontiguousarray(tri, dtype=dtype).reshape(9)), _p(np.ascontiguousarray(p, dtype=dtype).reshape(3))))


def nearest(tree_or_flat, shape_aabbs, points, tris=None):
    """BoundingHierarchy::nearest_to for n points: FlatBvh loop (flat_bvh.rs:513-562) if given a flat array,
    Bvh recursion (bvh_node.rs:327-374) if given a node array.  Shape distance: triangles if `tris` is given,
    else the shape's own AABB (UnitBox, testbase.rs:101-105).  returns (shape[n] u32, dist[n])"""
    is_flat = tree_or_flat.dtype in (FLAT_F32, FLAT_F64)
    s = "f32" if tree_or_flat.dtype in (FLAT_F32, NODE_F32) else "f64"
    ft = _types(s)[0]
    sa = np.ascontiguousarray(shape_aabbs, dtype=ft).reshape(-1, 6)
    pts = np.ascontiguousarray(points, dtype=ft).reshape(-1, 3)
    t = None if tris is None else np.ascontiguousarray(tris, dtype=ft).reshape(-1, 9)
    shape = np.zeros(len(pts), dtype=np.uint32)
    dist = np.zeros(len(pts), dtype=ft)
    fn = getattr(lib(), f"orc_nearest_{'flat' if is_flat else 'tree'}_{s}")
    fn.restype = None
    fn(_p(np.ascontiguousarray(tree_or_flat)), C.c_size_t(len(tree_or_flat)), _p(sa), _p(t), C.c_int(0 if t is None else 1),
       _p(pts), C.c_size_t(len(pts)), _p(shape), _p(dist))
    return shape, dist


def traverse_child_ordered(nodes, shape_aabbs, rays, ascending: bool = True):
    """Bvh::nearest_child_traverse_iterator / farthest_child_traverse_iterator collected per ray → (offsets, indices)"""
    s = "f32" if nodes.dtype == NODE_F32 else "f64"
    ft = _types(s)[0]
    sa = np.ascontiguousarray(shape_aabbs, dtype=ft).reshape(-1, 6)
    rays = np.ascontiguousarray(rays)
    nr = len(rays)
    offsets = np.zeros(nr + 1, dtype=np.uint32)
    fn = getattr(lib(), f"orc_traverse_child_ordered_{s}")
    fn.restype = C.c_uint64
    total = fn(_p(nodes), C.c_size_t(len(nodes)), _p(sa), _p(rays), C.c_size_t(nr), C.c_int(int(ascending)), _p(offsets),
               None, C.c_uint64(0))
    if total == 0xFFFFFFFFFFFFFFFF:
        raise OverflowError("tree deeper than the iterator's 32-entry stack (the reference panics)")
    indices = np.zeros(total, dtype=np.uint32)
    fn(_p(nodes), C.c_size_t(len(nodes)), _p(sa), _p(rays), C.c_size_t(nr), C.c_int(int(ascending)), _p(offsets),
       _p(indices), C.c_uint64(total))
    return offsets, indices


def traverse_distance(nodes, shape_aabbs, rays, ascending: bool = True, want_peak: bool = False):
    """Bvh::nearest_traverse_iterator / farthest_traverse_iterator (DistanceTraverseIterator, BinaryHeap driven)
    collected per ray → (offsets, indices[, largest heap length])"""
    s = "f32" if nodes.dtype == NODE_F32 else "f64"
    ft = _types(s)[0]
    sa = np.ascontiguousarray(shape_aabbs, dtype=ft).reshape(-1, 6)
    rays = np.ascontiguousarray(rays)
    nr = len(rays)
    offsets = np.zeros(nr + 1, dtype=np.uint32)
    fn = getattr(lib(), f"orc_traverse_distance_{s}")
    fn.restype = C.c_uint64
    peak = C.c_uint32(0)
    total = fn(_p(nodes), C.c_size_t(len(nodes)), _p(sa), _p(rays), C.c_size_t(nr), C.c_int(int(ascending)), _p(offsets),
               None, C.c_uint64(0), C.byref(peak))
    indices = np.zeros(total, dtype=np.uint32)
    fn(_p(nodes), C.c_size_t(len(nodes)), _p(sa), _p(rays), C.c_size_t(nr), C.c_int(int(ascending)), _p(offsets),
       _p(indices), C.c_uint64(total), C.byref(peak))
    return (offsets, indices, int(peak.value)) if want_peak else (offsets, indices)


def refit(nodes, aabbs):
    """same topology, child AABBs recomputed bottom-up from the (moved) shapes → new node array"""
    s = "f32" if nodes.dtype == NODE_F32 else "f64"
    ft = _types(s)[0]
    a = np.ascontiguousarray(aabbs, dtype=ft).reshape(-1, 6)
    out = np.ascontiguousarray(nodes).copy()
    fn = getattr(lib(), f"orc_refit_{s}")
    fn.restype = None
    fn(_p(out), C.c_size_t(len(out)), _p(a))
    return out


def check_tree(nodes, aabbs) -> int:
    s = "f32" if nodes.dtype == NODE_F32 else "f64"
    ft = _types(s)[0]
    a = np.ascontiguousarray(aabbs, dtype=ft).reshape(-1, 6)
    return int(getattr(lib(), f"orc_check_tree_{s}")(_p(nodes), C.c_size_t(len(nodes)), _p(a), C.c_size_t(len(a))))


def tree_stats(nodes, aabbs):
    s = "f32" if nodes.dtype == NODE_F32 else "f64"
    ft = _types(s)[0]
    a = np.ascontiguousarray(aabbs, dtype=ft).reshape(-1, 6)
    out = np.zeros(3, dtype=np.uint64)
    getattr(lib(), f"orc_tree_stats_{s}")(_p(nodes), C.c_size_t(len(nodes)), _p(a), _p(out))
    n = len(a)
    return dict(max_depth=int(out[0]), mean_leaf_depth=float(out[1]) / max(n, 1), degenerate_splits=int(out[2]))
