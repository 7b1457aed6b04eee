"""ctypes binding of libbvh_mi355x.so (include/bvh_mi355x.h).

The HIP engine is the only implementation: if the shared library is missing or a GPU call is made
without a device, this module raises — there is no CPU fallback anywhere in the product package.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("BVH_AMD_SO") or os.path.join(HERE, "libbvh_mi355x.so")  # BVH_AMD_SO: developer profiling builds only

OK, INVALID_ARG, HIP_ERROR, OOM, OVERFLOW, NO_DEVICE, DTYPE_MISMATCH, NOT_FLATTENED, RCCL_ERROR, REBROADCAST = range(10)
COMM_ID_BYTES = 128
BCAST_TRIANGLES = 1
F32, F64 = 0, 1
HOST, DEVICE = 0, 1
NONE = 0xFFFFFFFF
TRAVERSE_T_SLICE = 1
TRAVERSE_STATS = 2
TRAVERSE_TRIANGLES = 4
TRAVERSE_CLOSEST = 8
TRAVERSE_COHERENT = 16
TRAVERSE_NEAREST_FIRST = 32
TRAVERSE_FARTHEST_FIRST = 64
TRAVERSE_BEST_FIRST = 128
TRAVERSE_RAYS_READY = 256
# `order=` of the batch calls → flags: the child-ordered depth-first iterators and the heap-driven best-first ones
ORDER_FLAGS = {None: 0, "nearest": TRAVERSE_NEAREST_FIRST, "farthest": TRAVERSE_FARTHEST_FIRST,
               "nearest_heap": TRAVERSE_NEAREST_FIRST | TRAVERSE_BEST_FIRST,
               "farthest_heap": TRAVERSE_FARTHEST_FIRST | TRAVERSE_BEST_FIRST}

NODE_F32 = np.dtype([("l_min", "<f4", 3), ("l_max", "<f4", 3), ("r_min", "<f4", 3), ("r_max", "<f4", 3),
                     ("parent", "<u4"), ("l", "<u4"), ("r", "<u4"), ("shape", "<u4")])
NODE_F64 = np.dtype([("l_min", "<f8", 3), ("l_max", "<f8", 3), ("r_min", "<f8", 3), ("r_max", "<f8", 3),
                     ("parent", "<u4"), ("l", "<u4"), ("r", "<u4"), ("shape", "<u4")])
FLAT_F32 = np.dtype([("min", "<f4", 3), ("max", "<f4", 3), ("entry", "<u4"), ("exit", "<u4"), ("shape", "<u4")])
FLAT_F64 = np.dtype([("min", "<f8", 3), ("max", "<f8", 3), ("entry", "<u4"), ("exit", "<u4"), ("shape", "<u4"),
                     ("_pad", "<u4")])
RAY_F32 = np.dtype([("o", "<f4", 3), ("d", "<f4", 3), ("inv", "<f4", 3)])
RAY_F64 = np.dtype([("o", "<f8", 3), ("d", "<f8", 3), ("inv", "<f8", 3)])


class BvhGpuError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"bvhgpu status {status}: {message}")
        self.status = status


class TraverseStats(C.Structure):
    _fields_ = [("hits", C.c_uint64), ("visited", C.c_uint64), ("leaf_visits", C.c_uint64),
                ("device_steps", C.c_uint64), ("wave_steps", C.c_uint64)]


class Timings(C.Structure):
    _fields_ = [("build_ms", C.c_float), ("flatten_ms", C.c_float), ("traverse_kernel_ms", C.c_float),
                ("traverse_total_ms", C.c_float)]


# every symbol include/bvh_mi355x.h declares: (name, restype, argtypes)
_vp, _sz, _i, _u = C.c_void_p, C.c_size_t, C.c_int, C.c_uint
_pp = C.POINTER(C.c_void_p)
SYMBOLS = [
    ("bvhgpu_abi_version", _i, []),
    ("bvhgpu_device_count", _i, [C.POINTER(_i)]),
    ("bvhgpu_status_string", C.c_char_p, [_i]),
    ("bvhgpu_create", _i, [_i, _vp, _pp]),
    ("bvhgpu_destroy", None, [_vp]),
    ("bvhgpu_last_error", C.c_char_p, [_vp]),
    ("bvhgpu_synchronize", _i, [_vp]),
    ("bvhgpu_stream", _vp, [_vp]),
    ("bvhgpu_device_alloc", _i, [_vp, _sz, _pp]),
    ("bvhgpu_device_free", _i, [_vp, _vp]),
    ("bvhgpu_device_copy", _i, [_vp, _vp, _i, _vp, _i, _sz]),
    ("bvhgpu_host_alloc", _i, [_vp, _sz, _pp]),
    ("bvhgpu_host_free", _i, [_vp, _vp]),
    ("bvhgpu_host_register", _i, [_vp, _vp, _sz]),
    ("bvhgpu_host_unregister", _i, [_vp, _vp]),
    ("bvhgpu_build_f32", _i, [_vp, _vp, _sz, _i, _pp]),
    ("bvhgpu_build_f64", _i, [_vp, _vp, _sz, _i, _pp]),
    ("bvhgpu_rebuild_f32", _i, [_vp, _vp, _sz, _i]),
    ("bvhgpu_rebuild_f64", _i, [_vp, _vp, _sz, _i]),
    ("bvhgpu_build_flat_f32", _i, [_vp, _vp, _sz, _i, _pp]),
    ("bvhgpu_build_flat_f64", _i, [_vp, _vp, _sz, _i, _pp]),
    ("bvhgpu_refit_f32", _i, [_vp, _vp, _sz, _i]),
    ("bvhgpu_refit_f64", _i, [_vp, _vp, _sz, _i]),
    ("bvhgpu_rebuild_flat_f32", _i, [_vp, _vp, _sz, _i]),
    ("bvhgpu_rebuild_flat_f64", _i, [_vp, _vp, _sz, _i]),
    ("bvhgpu_rebuild_flat_async_f32", _i, [_vp, _vp, _sz, _i]),
    ("bvhgpu_rebuild_flat_async_f64", _i, [_vp, _vp, _sz, _i]),
    ("bvhgpu_tree_wait", _i, [_vp]),
    ("bvhgpu_traverse_async_f32", _i, [_vp, _vp, _sz, _i, _u, _pp]),
    ("bvhgpu_traverse_async_f64", _i, [_vp, _vp, _sz, _i, _u, _pp]),
    ("bvhgpu_hits_wait", _i, [_vp]),
    ("bvhgpu_tree_destroy", None, [_vp]),
    ("bvhgpu_tree_info", _i, [_vp, C.POINTER(_i), C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz)]),
    ("bvhgpu_tree_nodes", _i, [_vp, _vp, _i]),
    ("bvhgpu_tree_shape_nodes", _i, [_vp, _vp, _i]),
    ("bvhgpu_tree_build_levels", _i, [_vp, C.POINTER(_i)]),
    ("bvhgpu_flatten", _i, [_vp]),
    ("bvhgpu_flat_nodes", _i, [_vp, _vp, _i]),
    ("bvhgpu_tree_from_flat_f32", _i, [_vp, _vp, _sz, _vp, _sz, _pp]),
    ("bvhgpu_tree_from_flat_f64", _i, [_vp, _vp, _sz, _vp, _sz, _pp]),
    ("bvhgpu_scene_nbytes", _i, [_vp, C.POINTER(_sz)]),
    ("bvhgpu_scene_export", _i, [_vp, _vp, _i]),
    ("bvhgpu_scene_import", _i, [_vp, _vp, _sz, _i, _pp]),
    ("bvhgpu_comm_unique_id", _i, [_vp]),
    ("bvhgpu_comm_init_rank", _i, [_vp, _i, _i, _vp, _pp]),
    ("bvhgpu_comm_init_all", _i, [_pp, _i, _pp]),
    ("bvhgpu_comm_info", _i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    ("bvhgpu_comm_destroy", None, [_vp]),
    ("bvhgpu_rccl_info", _i, [C.POINTER(_i), C.POINTER(_i), C.c_char_p, _sz]),
    ("bvhgpu_bcast", _i, [_vp, _pp, _i]),
    ("bvhgpu_bcast_known", _i, [_vp, _pp, _i, _i, _sz, _u]),
    ("bvhgpu_rays_new_f32", _i, [_vp, _vp, _vp, _sz, _i, _vp, _i]),
    ("bvhgpu_rays_new_f64", _i, [_vp, _vp, _vp, _sz, _i, _vp, _i]),
    ("bvhgpu_gen_rays_f32", _i, [_vp, C.c_uint64, _sz, _vp, _vp]),
    ("bvhgpu_gen_rays_f64", _i, [_vp, C.c_uint64, _sz, _vp, _vp]),
    ("bvhgpu_gen_primary_rays_f32", _i, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint64, _sz, _vp]),
    ("bvhgpu_gen_primary_rays_f64", _i, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint64, _sz, _vp]),
    ("bvhgpu_nearest_f32", _i, [_vp, _vp, _sz, _i, _i, _vp, _vp]),
    ("bvhgpu_nearest_f64", _i, [_vp, _vp, _sz, _i, _i, _vp, _vp]),
    ("bvhgpu_ray_triangle_pairs_f32", _i, [_vp, _vp, _vp, _sz, _i, _vp]),
    ("bvhgpu_ray_triangle_pairs_f64", _i, [_vp, _vp, _vp, _sz, _i, _vp]),
    ("bvhgpu_traverse_f32", _i, [_vp, _vp, _sz, _i, _u, _pp]),
    ("bvhgpu_traverse_f64", _i, [_vp, _vp, _sz, _i, _u, _pp]),
    ("bvhgpu_traverse_host_f32", _i, [_vp, _vp, _vp, _sz, _u, _vp, _vp, _sz, C.POINTER(C.c_uint64)]),
    ("bvhgpu_traverse_host_f64", _i, [_vp, _vp, _vp, _sz, _u, _vp, _vp, _sz, C.POINTER(C.c_uint64)]),
    ("bvhgpu_traverse_host_indices", _i, [_vp, _vp, _sz]),
    ("bvhgpu_build_traverse_host_f32", _i, [_vp, _vp, _sz, _vp, _vp, _sz, _u, _vp, _vp, _sz, C.POINTER(C.c_uint64)]),
    ("bvhgpu_build_traverse_host_f64", _i, [_vp, _vp, _sz, _vp, _vp, _sz, _u, _vp, _vp, _sz, C.POINTER(C.c_uint64)]),
    ("bvhgpu_tree_set_triangles_f32", _i, [_vp, _vp, _sz, _i]),
    ("bvhgpu_tree_set_triangles_f64", _i, [_vp, _vp, _sz, _i]),
    ("bvhgpu_hits_fetch_triangles", _i, [_vp, _vp, _i]),
    ("bvhgpu_hits_fetch_closest", _i, [_vp, _vp, _vp, _i]),
    ("bvhgpu_hits_info", _i, [_vp, C.POINTER(_sz), C.POINTER(C.c_uint64), C.POINTER(TraverseStats)]),
    ("bvhgpu_hits_walk_info", _i, [_vp, C.POINTER(C.c_uint)]),
    ("bvhgpu_hits_walk_kernel", _i, [_vp, C.c_char_p, _sz]),
    ("bvhgpu_hits_fetch", _i, [_vp, _vp, _vp, _vp, _i]),
    ("bvhgpu_hits_device", _i, [_vp, _pp, _pp, _pp]),
    ("bvhgpu_hits_destroy", None, [_vp]),
    ("bvhgpu_enable_timing", _i, [_vp, _i]),
    ("bvhgpu_last_timings", _i, [_vp, C.POINTER(Timings)]),
    ("bvhgpu_obj_parse", _i, [C.c_char_p, _sz, C.POINTER(C.POINTER(C.c_float)), C.POINTER(_sz), _vp]),
    ("bvhgpu_obj_free", None, [C.POINTER(C.c_float)]),
    ("bvhgpu_obj_last_error", C.c_char_p, []),
    ("bvhgpu_triangles_aabbs_f32", _i, [_vp, _sz, _vp]),
    ("bvhgpu_set_tuning", _i, [_vp, _i, _i]),
    ("bvhgpu_get_tuning", _i, [_vp, _i, C.POINTER(_i)]),
]
TUNE_TRAVERSE_VARIANT = 0
TUNE_TRAVERSE_LDS_MIN_RAYS, TUNE_TRAVERSE_LDS_SLOTS, TUNE_TRAVERSE_LDS_THREADS, TUNE_TRAVERSE_SPLIT = 3, 4, 5, 6
TUNE_WIDE_ITEMS_LOG4, TUNE_WIDE_STACK_LDS, TUNE_WIDE_WG_PER_CU, TUNE_WIDE_THREADS, TUNE_WIDE_SLOTS = 1, 2, 7, 8, 9
TUNE_WIDE_EARLY_ITEMS = 11       # wide walk on a tree being rebuilt, RAYS_READY batches: item filter beside the build (1) or in the walk's prologue (0, default)
TUNE_WIDE_STAGE_SHIFT = 12       # wide walk, whole rays, indices only: 2^v shapes per ray staged without pool records (-1 default = 3, 0 off)
TUNE_WIDE_REC8 = 13              # wide walk, whole rays, indices only: pair records, 8 bytes per hit (1, default) or the 12-byte HitRec (0)
TUNE_BUILD_LEVEL_LAUNCHES = 10   # builder, level tier: 1 one launch per level, 2 k_bin + k_split per level, 0 (default) by scene size
TUNE_WIDE_F64_GUIDE = 14        # wide walk, f64 trees, indices only: walk the f32 guide boxes, test leaf candidates in f64 (1, default) or the f64 walk (0)
TUNE_FLATTEN_LAZY = 15           # the flatten behind a build writes the wide walk's arrays; FlatNode / binary arrays follow on first use (1, default), at once (0), at once on the side stream beside the walk (2), or the FlatNode array at once and the binary array on first use (3)
TUNE_WIDE_MIN_RAYS_PER_WG = 18   # wide walk over items, batches below 512 K rays: spread over all workgroup slots down to this many rays each (256 default, 0 off)
TUNE_HOST_ZERO_COPY = 19         # host batches on pinned buffers: bit 0 the device reads the ray arrays itself, bit 1 it writes offsets / indices itself (2 default)
TUNE_BUILD_LEVEL_TILE = 20      # builder, two launches per level: positions per tile (0 default = by scene size, 512 .. 4096)
TUNE_FLATTEN_INLINE = 21        # f32: the builder's wave tier writes the flatten's FLAT / WIDE parts for its subtrees itself (1 default, 0 = the flatten kernel writes everything)
TUNE_HOST_CHUNKS = 17            # bvhgpu_traverse_host_*: chunks the batch is walked in (0, default = by batch size)
TUNE_BUILD_LEVEL_PERSIST = 16    # builder, level tier: tree levels 3.. of the tier as ONE persistent launch, one level-3 subtree per XCD (1) or a launch per level (0)
TRAVERSE_RAYS_OD6 = 512
WALK_WIDE, WALK_STAGED, WALK_REC8, WALK_F64_GUIDE = 1, 2, 4, 8   # bvhgpu_hits_walk_info
ABI_VERSION = 7

_lib = None


def load():
    """Load the engine.  Raises ImportError with build instructions if the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if os.environ.get("BVH_AMD_NO_TORCH") != "1":
        # PyTorch-ROCm wheels bundle their own HIP/HSA runtime (torch/lib/libamdhip64.so, soname
        # libamdhip64.so.7).  Two HIP runtimes in one process cannot both own the GPU, so when torch is
        # installed it must be loaded FIRST; our library then binds to the already-loaded runtime by soname.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"{SO_PATH} is missing: the MI355X HIP engine has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `python bvh_amd/build_ext.py`). "
            "There is no CPU fallback.")
    lib = C.CDLL(SO_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.bvhgpu_abi_version() != ABI_VERSION:
        raise ImportError("libbvh_mi355x.so ABI version mismatch")
    _lib = lib
    return lib


def device_count() -> int:
    n = C.c_int(0)
    load().bvhgpu_device_count(C.byref(n))
    return int(n.value)


def check(rc: int, ctx=None):
    if rc == OK:
        return
    lib = load()
    msg = lib.bvhgpu_last_error(ctx).decode() if True else ""
    if not msg:
        msg = lib.bvhgpu_status_string(rc).decode()
    raise BvhGpuError(rc, msg)


def ptr(a):
    """c_void_p of a numpy array (host) or an int device address; None → NULL."""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return C.c_void_p(int(a))
    return a.ctypes.data_as(C.c_void_p)
