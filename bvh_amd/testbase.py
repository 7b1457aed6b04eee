"""Synthetic inputs of the reference's test/bench harness (src/testbase.rs), as numpy arrays.

These are INPUT GENERATORS, not part of the measured path: integer splitmix64 plus element-wise
IEEE f32 operations, which numpy evaluates exactly like the Rust code, so the arrays are
bit-identical to the reference's scenes (cross-checked against the oracle in tests/).
"""
from __future__ import annotations

import numpy as np

from .api import Aabb, BHShape

GAMMA = np.uint64(0x9E3779B97F4A7C15)


def default_bounds() -> np.ndarray:  # testbase.rs:598-603
    return np.array([-100000.0] * 3 + [100000.0] * 3, dtype=np.float32)


def _splitmix64_at(draw_index: np.ndarray) -> np.ndarray:
    """value of the j-th draw (1-based) of splitmix64 seeded with 0 (testbase.rs:558-564):
    the state before mixing is j*GAMMA (mod 2^64)."""
    with np.errstate(over="ignore"):
        z = draw_index.astype(np.uint64) * GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def next_point3_at(draw_index: np.ndarray, bounds: np.ndarray) -> np.ndarray:
    """next_point3 (testbase.rs:567-595) for the given 1-based draw indices → (n,3) float32."""
    u = _splitmix64_at(np.asarray(draw_index, dtype=np.uint64))
    a = ((u >> np.uint64(32)) & np.uint64(0xFFFFFFFF)).astype(np.int64) - np.int64(0x80000000)
    b = (u & np.uint64(0xFFFFFFFF)).astype(np.int64) - np.int64(0x80000000)
    ub = b.astype(np.uint64)
    rot = (ub << np.uint64(6)) | (ub >> np.uint64(58))
    c = a.astype(np.uint64) ^ rot
    raw = np.stack([a.astype(np.int32), b.astype(np.int32), (c & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)],
                   axis=1)
    b32 = np.asarray(bounds, dtype=np.float32)
    fv = (raw.astype(np.float32) / np.float32(2147483648.0) + np.float32(1.0)) * np.float32(0.5)
    size = b32[3:] - b32[:3]
    return (b32[:3] + fv * size).astype(np.float32)


_CORNERS = np.array([[0.5, 0.5, -0.5], [0.5, 0.5, 0.5], [-0.5, 0.5, 0.5], [-0.5, 0.5, -0.5],
                     [0.5, -0.5, -0.5], [0.5, -0.5, 0.5], [-0.5, -0.5, 0.5], [-0.5, -0.5, -0.5]], dtype=np.float32)
_TFR, _TBR, _TBL, _TFL, _BFR, _BBR, _BBL, _BFL = range(8)
_CUBE_TRIS = np.array([  # push_cube, testbase.rs:490-554 (vertex order preserved)
    [_TBR, _TFR, _TFL], [_TFL, _TBL, _TBR], [_BFL, _BFR, _BBR], [_BBR, _BBL, _BFL],
    [_TBL, _TFL, _BFL], [_BFL, _BBL, _TBL], [_BFR, _TFR, _TBR], [_TBR, _BBR, _BFR],
    [_TFL, _TFR, _BFR], [_BFR, _BFL, _TFL], [_BBR, _TBR, _TBL], [_TBL, _BBL, _BBR]])


def triangles_aabbs(tris: np.ndarray) -> np.ndarray:
    """Triangle::new's aabb = empty.grow(a).grow(b).grow(c) (testbase.rs:325-333) → (n,6)."""
    return np.concatenate([tris.min(axis=1), tris.max(axis=1)], axis=1).astype(tris.dtype)


def create_n_cubes(n_cubes: int, bounds=None):
    """create_n_cubes (testbase.rs:608-615, seed 0) → (tris (12n,3,3) f32, aabbs (12n,6) f32)."""
    bounds = default_bounds() if bounds is None else bounds
    pos = next_point3_at(np.arange(1, n_cubes + 1, dtype=np.uint64), bounds)      # (n,3)
    verts = pos[:, None, :] + _CORNERS[None, :, :]                                  # (n,8,3)
    tris = verts[:, _CUBE_TRIS, :].reshape(n_cubes * 12, 3, 3).astype(np.float32)
    return tris, triangles_aabbs(tris)


def generate_aligned_boxes_aabbs() -> np.ndarray:
    """generate_aligned_boxes + UnitBox::aabb (testbase.rs:109-116, 84-89) → (21,6) f32."""
    pos = np.zeros((21, 3), dtype=np.float32)
    pos[:, 0] = np.arange(-10, 11, dtype=np.float32)
    return np.concatenate([pos + np.float32(-0.5), pos + np.float32(0.5)], axis=1)


class UnitBox(BHShape):
    """struct UnitBox (testbase.rs:65-100)."""

    def __init__(self, id: int, pos):
        self.id = id
        self.pos = np.asarray(pos, dtype=np.float32)
        self.node_index = 0

    def aabb(self) -> Aabb:
        return Aabb(self.pos + np.float32(-0.5), self.pos + np.float32(0.5))

    def set_bh_node_index(self, index: int) -> None:
        self.node_index = index

    def bh_node_index(self) -> int:
        return self.node_index


def generate_aligned_boxes():
    return [UnitBox(x, (float(x), 0.0, 0.0)) for x in range(-10, 11)]


class Triangle(BHShape):
    """struct Triangle (testbase.rs:316-356)."""

    def __init__(self, a, b, c):
        self.a, self.b, self.c = (np.asarray(v, dtype=np.float32) for v in (a, b, c))
        self._aabb = Aabb.empty().grow(self.a).grow(self.b).grow(self.c)
        self.node_index = 0

    def aabb(self) -> Aabb:
        return self._aabb

    def set_bh_node_index(self, index: int) -> None:
        self.node_index = index

    def bh_node_index(self) -> int:
        return self.node_index
