"""TEST INFRASTRUCTURE ONLY (see oracle/bvh_oracle.h) — CPU restatement of the reference's OBJ ingest.

`obj::load_obj::<Triangle>` (crate obj-rs 0.7, un-vendored) followed by `FromRawVertex::process`
(/root/reference/src/testbase.rs:445-487): positions from `v x y z [w]`, faces `f` in the four polygon formats
P (`i`), PT (`i/t`), PN (`i//n`), PTN (`i/t/n`), 1-based indices, negative = relative to the vertices read so far;
each polygon → triangle FAN (anchor, second, third), second = third (:461-469).  Scene bounds = join of the
triangle AABBs (load_sponza_scene, :628-631).  Plain Python, small inputs only.
"""
from __future__ import annotations

import numpy as np


class ObjError(ValueError):
    pass


def parse_obj(text: str):
    """→ (tris (n,3,3) float32, aabbs (n,6) float32, bounds (6,) float32)"""
    pts = []
    tris = []
    logical = text.replace("\\\r\n", " ").replace("\\\n", " ")   # statement continuation
    for ln, raw in enumerate(logical.split("\n"), 1):
        line = raw.split("#", 1)[0].strip()
        if not line:
            continue
        tok = line.split()
        kw, args = tok[0], tok[1:]
        if kw == "v":
            if len(args) not in (3, 4):
                raise ObjError(f"line {ln}: vertex needs 3 or 4 numbers")
            try:
                pts.append([np.float32(a) for a in args[:3]])    # w dropped (testbase.rs:455-458)
                [np.float32(a) for a in args[3:]]
            except ValueError:
                raise ObjError(f"line {ln}: vertex needs 3 or 4 numbers")
        elif kw == "f":
            kinds = set()
            idx = []
            for a in args:
                parts = a.split("/")
                if len(parts) > 3 or parts[0] == "":
                    raise ObjError(f"line {ln}: malformed face vertex")
                kinds.add((len(parts) > 1 and parts[1] != "", len(parts) > 2))
                try:
                    i = int(parts[0])
                    [int(q) for q in parts[1:] if q != ""]
                except ValueError:
                    raise ObjError(f"line {ln}: malformed face vertex")
                j = i - 1 if i > 0 else (len(pts) + i if i < 0 else -1)
                if not (0 <= j < len(pts)):
                    raise ObjError(f"line {ln}: face index out of range")
                idx.append(j)
            if len(kinds) > 1:
                raise ObjError(f"line {ln}: face mixes vertex formats")
            if len(idx) < 2:
                raise ObjError(f"line {ln}: a face needs at least two vertices")
            anchor, second = pts[idx[0]], pts[idx[1]]
            for k in idx[2:]:
                third = pts[k]
                tris.append([anchor, second, third])
                second = third
        elif kw in ("vt", "vn", "vp", "g", "o", "s", "usemtl", "mtllib", "l", "p"):
            continue
        else:
            raise ObjError(f"line {ln}: unexpected statement")
    t = np.array(tris, dtype=np.float32).reshape(-1, 3, 3)

    # Triangle::new: empty.grow(a).grow(b).grow(c) (testbase.rs:325-333).  f32::min/max leave the sign of a zero
    # result open; like the rest of the oracle (bvh_oracle.h) the join orders -0 < +0.
    def jmin(x, y):
        return np.where((x < y) | ((x == y) & np.signbit(x)), x, y)

    def jmax(x, y):
        return np.where((x > y) | ((x == y) & ~np.signbit(x)), x, y)
    if len(t):
        mn = jmin(jmin(t[:, 0], t[:, 1]), t[:, 2])
        mx = jmax(jmax(t[:, 0], t[:, 1]), t[:, 2])
        aabbs = np.concatenate([mn, mx], axis=1).astype(np.float32)
    else:
        aabbs = np.zeros((0, 6), np.float32)
    bounds = np.array([np.inf] * 3 + [-np.inf] * 3, dtype=np.float32)
    for row in aabbs:   # TAabb3::empty().join_mut(triangle.aabb()) in file order (testbase.rs:628-631)
        bounds[:3] = jmin(bounds[:3], row[:3])
        bounds[3:] = jmax(bounds[3:], row[3:])
    return t, aabbs, bounds
