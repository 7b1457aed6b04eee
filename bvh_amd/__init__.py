"""bvh_amd — MI355X-native (gfx950) engine for the hot path of the Rust crate `bvh`:
SAH build → flatten → batched ray traversal, behind the C ABI in include/bvh_mi355x.h.

The package holds only what that path needs: csrc/ (HIP kernels + C ABI), _lib (ctypes binding),
api (mirror of the reference's Bounded / BHShape / BoundingHierarchy / Bvh / FlatBvh / Ray surface)
and testbase (the reference's scene and ray generators, used as synthetic inputs).
"""
from ._lib import BvhGpuError, NONE, device_count  # noqa: F401
from .api import (Aabb, BHShape, Bounded, Bvh, Context, FlatBvh, HostStep, Ray, RayBatch,  # noqa: F401
                  default_context, pinned_array)

__all__ = ["Aabb", "BHShape", "Bounded", "Bvh", "Context", "FlatBvh", "HostStep", "pinned_array", "Ray", "RayBatch", "BvhGpuError",
           "NONE", "device_count", "default_context"]
