"""Scene ingest for the reference's OBJ-based benches (src/testbase.rs:445-487, 619-634), over the C ABI.

load_obj / parse_obj  — `obj::load_obj::<Triangle>`: OBJ text → triangles by fan triangulation (C++ parser in
                        libbvh_mi355x.so, bvh_amd/csrc/obj.cpp); scene bounds = join of the triangle AABBs.
make_atrium_obj       — a procedural stand-in for media/sponza.obj, which the reference repository does not ship
                        (SURVEY §8d): an atrium with a floor, walls, two storeys of colonnades with polygonal
                        columns, arches made of n-gons and a roof lattice, emitted as OBJ TEXT so that the same
                        ingest path is exercised.  It is NOT Sponza; every result on it is labelled "stand-in".
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Tuple

import numpy as np

from . import _lib
from ._lib import check


def parse_obj(text) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """OBJ text (str or bytes) → (tris (n,3,3) f32, aabbs (n,6) f32, bounds (6,) f32)."""
    lib = _lib.load()
    data = text.encode() if isinstance(text, str) else bytes(text)
    out = C.POINTER(C.c_float)()
    n = C.c_size_t()
    bounds = np.zeros(6, dtype=np.float32)
    rc = lib.bvhgpu_obj_parse(data, len(data), C.byref(out), C.byref(n), bounds.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise _lib.BvhGpuError(rc, lib.bvhgpu_obj_last_error().decode())
    try:
        tris = np.ctypeslib.as_array(out, shape=(n.value * 9,)).copy().reshape(-1, 3, 3) if n.value else \
            np.zeros((0, 3, 3), dtype=np.float32)
    finally:
        lib.bvhgpu_obj_free(out)
    aabbs = np.zeros((len(tris), 6), dtype=np.float32)
    if len(tris):
        check(lib.bvhgpu_triangles_aabbs_f32(tris.ctypes.data_as(C.c_void_p), len(tris), aabbs.ctypes.data_as(C.c_void_p)))
    return tris, aabbs, bounds


def load_obj(path: str):
    """load_sponza_scene for any file (testbase.rs:619-634)."""
    with open(path, "rb") as f:
        return parse_obj(f.read())


# ------------------------------------------------------------------------------------------------
class _ObjWriter:
    def __init__(self):
        self.lines = ["# procedural atrium — stand-in for media/sponza.obj (not shipped with the reference)"]
        self.nv = 0

    def v(self, x, y, z):
        self.lines.append(f"v {x:.6f} {y:.6f} {z:.6f}")
        self.nv += 1
        return self.nv  # 1-based

    def f(self, idx, style=0):
        if style == 0:
            self.lines.append("f " + " ".join(str(i) for i in idx))
        elif style == 1:
            self.lines.append("f " + " ".join(f"{i}/1" for i in idx))
        elif style == 2:
            self.lines.append("f " + " ".join(f"{i}//1" for i in idx))
        else:
            self.lines.append("f " + " ".join(f"{i - self.nv - 1}/1/1" for i in idx))  # negative (relative) indices

    def quad_grid(self, origin, du, dv, nu, nv, style=0):
        """(nu x nv) quads spanning origin + i*du + j*dv."""
        o = np.asarray(origin, float); du = np.asarray(du, float); dv = np.asarray(dv, float)
        ids = [[self.v(*(o + i * du + j * dv)) for j in range(nv + 1)] for i in range(nu + 1)]
        for i in range(nu):
            for j in range(nv):
                self.f([ids[i][j], ids[i + 1][j], ids[i + 1][j + 1], ids[i][j + 1]], style)

    def prism(self, cx, cz, y0, y1, radius, sides, rings, style=0):
        """a column: `sides`-gon cross-section, `rings` stacked bands of quads, n-gon caps."""
        ring_ids = []
        for r in range(rings + 1):
            y = y0 + (y1 - y0) * r / rings
            ring_ids.append([self.v(cx + radius * math.cos(2 * math.pi * s / sides), y,
                                    cz + radius * math.sin(2 * math.pi * s / sides)) for s in range(sides)])
        for r in range(rings):
            for s in range(sides):
                t = (s + 1) % sides
                self.f([ring_ids[r][s], ring_ids[r][t], ring_ids[r + 1][t], ring_ids[r + 1][s]], style)
        self.f(list(reversed(ring_ids[0])), style)   # n-gon caps: fan-triangulated by the loader
        self.f(ring_ids[-1], style)

    def arch(self, x0, x1, z, y_spring, thickness, segs, style=0):
        """a semicircular arch between two columns: `segs` quads on the intrados + two n-gon side faces."""
        cx, rad = 0.5 * (x0 + x1), 0.5 * (x1 - x0)
        front, back = [], []
        for s in range(segs + 1):
            a = math.pi * s / segs
            x, y = cx - rad * math.cos(a), y_spring + rad * math.sin(a)
            front.append(self.v(x, y, z - thickness)); back.append(self.v(x, y, z + thickness))
        for s in range(segs):
            self.f([front[s], front[s + 1], back[s + 1], back[s]], style)
        top_l_f = self.v(x0, y_spring + rad * 1.15, z - thickness); top_r_f = self.v(x1, y_spring + rad * 1.15, z - thickness)
        self.f([top_l_f] + front + [top_r_f], style)
        top_l_b = self.v(x0, y_spring + rad * 1.15, z + thickness); top_r_b = self.v(x1, y_spring + rad * 1.15, z + thickness)
        self.f([top_r_b] + list(reversed(back)) + [top_l_b], style)


def make_atrium_obj(detail: int = 4) -> str:
    """OBJ text of the stand-in atrium.  detail 1 → ~4 k triangles (tests), 4 → ~70 k, 8 → ~270 k (bench)."""
    w = _ObjWriter()
    L, W, H = 36.0, 14.0, 16.0            # length (x), width (z), height (y): Sponza-like proportions
    g = 4 * detail
    w.quad_grid((-L / 2, 0, -W / 2), (L / g / 1.0, 0, 0), (0, 0, W / g), g, g, 0)                 # floor
    w.quad_grid((-L / 2, H, -W / 2), (L / g, 0, 0), (0, 0, W / g), g, g, 1)                        # ceiling
    w.quad_grid((-L / 2, 0, -W / 2), (L / g, 0, 0), (0, H / g, 0), g, g, 2)                        # back wall
    w.quad_grid((-L / 2, 0, W / 2), (L / g, 0, 0), (0, H / g, 0), g, g, 3)                         # front wall
    w.quad_grid((-L / 2, 0, -W / 2), (0, 0, W / g), (0, H / g, 0), g, g, 0)                        # end walls
    w.quad_grid((L / 2, 0, -W / 2), (0, 0, W / g), (0, H / g, 0), g, g, 1)
    ncol = 10
    sides, rings, segs = 6 + 2 * detail, 2 + 2 * detail, 4 + 2 * detail
    for storey, (y0, y1) in enumerate(((0.0, 5.0), (7.0, 12.0))):
        for side_z in (-W / 2 + 2.5, W / 2 - 2.5):
            xs = [-L / 2 + 3.0 + i * (L - 6.0) / (ncol - 1) for i in range(ncol)]
            for i, x in enumerate(xs):
                w.prism(x, side_z, y0, y1, 0.45, sides, rings, (i + storey) % 4)
            for i in range(ncol - 1):
                w.arch(xs[i] + 0.45, xs[i + 1] - 0.45, side_z, y1, 0.3, segs, (i + storey) % 4)
        # gallery floor slabs between the storeys
        w.quad_grid((-L / 2, y1 + 1.7, -W / 2), (L / g, 0, 0), (0, 0, 2.5 / max(detail // 2, 1)), g, max(detail // 2, 1), storey)
        w.quad_grid((-L / 2, y1 + 1.7, W / 2 - 2.5), (L / g, 0, 0), (0, 0, 2.5 / max(detail // 2, 1)), g, max(detail // 2, 1), storey + 2)
    # roof lattice: thin beams crossing the void
    nb = 3 * detail
    for i in range(nb):
        x = -L / 2 + (i + 0.5) * L / nb
        w.prism(x, 0.0, H - 1.2, H - 0.9, 0.12, 4, 1, i % 4)
    return "\n".join(w.lines) + "\n"
