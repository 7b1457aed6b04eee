"""Second, independent restatement of the reference hot path in pure Python + numpy scalars.

TEST INFRASTRUCTURE ONLY.  Written directly from /root/reference/src (not from
oracle/bvh_oracle.c) with a different structure — Python lists for the bucket
assignment vectors exactly like bvh_node.rs:195-222, recursion like :54-59 — so
that a transcription slip in either restatement shows up as a disagreement in
tests/test_oracle_golden.py.  Slow: use on <= a few thousand shapes.

Every numpy-scalar operation below is one correctly rounded IEEE operation in
the array's dtype (np.float32 / np.float64); there is no FMA in numpy scalars.
"""
from __future__ import annotations

import numpy as np

NONE = 0xFFFFFFFF
NUM_BUCKETS = 6  # bvh/bucket.rs:5


def _min(a, b):  # IEEE-754-2019 minimum on NaN-free data
    return a if (a < b or (a == b and np.signbit(a))) else b


def _max(a, b):
    return a if (a > b or (a == b and not np.signbit(a))) else b


class Aabb:
    """aabb/aabb_impl.rs:10-16"""
    __slots__ = ("min", "max", "ft")

    def __init__(self, mn, mx, ft):
        self.ft = ft
        self.min = [ft(v) for v in mn]
        self.max = [ft(v) for v in mx]

    @staticmethod
    def empty(ft):  # :119-124
        return Aabb([np.inf] * 3, [-np.inf] * 3, ft)

    def join(self, o):  # :303-308
        return Aabb([_min(a, b) for a, b in zip(self.min, o.min)], [_max(a, b) for a, b in zip(self.max, o.max)],
                    self.ft)

    def grow(self, p):  # :375-380
        return Aabb([_min(a, b) for a, b in zip(self.min, p)], [_max(a, b) for a, b in zip(self.max, p)], self.ft)

    def size(self):  # :459-461
        return [b - a for a, b in zip(self.min, self.max)]

    def center(self):  # :501-504
        h = self.ft(0.5)
        return [a * h + b * h for a, b in zip(self.min, self.max)]

    def surface_area(self):  # :551-554
        s = self.size()
        return self.ft(2.0) * ((s[0] * s[0] + s[1] * s[1]) + s[2] * s[2])

    def largest_axis(self):  # :594-596 (imax: first strict maximum)
        s = self.size()
        a, m = 0, s[0]
        for i in (1, 2):
            if s[i] > m:
                a, m = i, s[i]
        return a

    def as6(self):
        return list(self.min) + list(self.max)


def joint_aabb_of_shapes(indices, boxes, ft):  # utils.rs:97-109
    a, c = Aabb.empty(ft), Aabb.empty(ft)
    for i in indices:
        a = a.join(boxes[i])
        c = c.grow(boxes[i].center())
    return a, c


def build(aabbs):
    """Bvh::build — bvh_impl.rs:40-96.  returns (nodes list of dict, shape_node list)."""
    aabbs = np.asarray(aabbs)
    ft = aabbs.dtype.type
    n = len(aabbs)
    if n == 0:
        return [], []
    boxes = [Aabb(b[:3], b[3:], ft) for b in aabbs]
    nodes = [None] * (2 * n - 1)
    shape_node = [None] * n
    indices = list(range(n))
    eps = np.finfo(ft).eps
    with np.errstate(all="ignore"):
        A, Cb = joint_aabb_of_shapes(indices, boxes, ft)
        _build_node(boxes, indices, nodes, shape_node, 0, 0, A, Cb, ft, eps)
    return nodes, shape_node


def _build_node(boxes, indices, nodes, shape_node, ni, parent, A, Cb, ft, eps):
    # prep_build — bvh_node.rs:81-180
    if len(indices) == 1:
        nodes[ni] = dict(leaf=True, parent=parent, shape=indices[0])
        shape_node[indices[0]] = ni
        return
    ax = Cb.largest_axis()
    ext = Cb.max[ax] - Cb.min[ax]
    if ext < eps:
        half = len(indices) // 2
        l_idx, r_idx = indices[:half], indices[half:]
        AL, CL = joint_aabb_of_shapes(l_idx, boxes, ft)
        AR, CR = joint_aabb_of_shapes(r_idx, boxes, ft)
    else:
        # build_buckets — bvh_node.rs:183-279
        cnt = [0] * NUM_BUCKETS
        baabb = [Aabb.empty(ft) for _ in range(NUM_BUCKETS)]
        bcen = [Aabb.empty(ft) for _ in range(NUM_BUCKETS)]
        assign = [[] for _ in range(NUM_BUCKETS)]
        K = ft(NUM_BUCKETS) - ft(0.01)
        for idx in indices:
            box = boxes[idx]
            c = box.center()
            rel = (c[ax] - Cb.min[ax]) / ext
            b = int(rel * K)  # to_usize: truncation
            cnt[b] += 1
            baabb[b] = baabb[b].join(box)
            bcen[b] = bcen[b].grow(c)
            assign[b].append(idx)
        best, min_cost = 0, ft(np.inf)
        AL = CL = AR = CR = Aabb.empty(ft)
        for i in range(NUM_BUCKETS - 1):
            ln = rn = 0
            la, lc, ra, rc = (Aabb.empty(ft) for _ in range(4))
            for b in range(0, i + 1):
                ln += cnt[b]; la = la.join(baabb[b]); lc = lc.join(bcen[b])
            for b in range(i + 1, NUM_BUCKETS):
                rn += cnt[b]; ra = ra.join(baabb[b]); rc = rc.join(bcen[b])
            cost = (ft(ln) * la.surface_area() + ft(rn) * ra.surface_area()) / A.surface_area()
            if cost < min_cost:
                best, min_cost = i, cost
                AL, CL, AR, CR = la, lc, ra, rc
        l_idx = [s for g in assign[:best + 1] for s in g]
        r_idx = [s for g in assign[best + 1:] for s in g]
    li = ni + 1
    ri = li + (2 * len(l_idx) - 1)
    nodes[ni] = dict(leaf=False, parent=parent, l=li, r=ri, l_aabb=AL.as6(), r_aabb=AR.as6())
    _build_node(boxes, l_idx, nodes, shape_node, li, ni, AL, CL, ft, eps)
    _build_node(boxes, r_idx, nodes, shape_node, ri, ni, AR, CR, ft, eps)


def flatten(nodes, ft=np.float32):
    """Bvh::flatten — flat_bvh.rs:60-143,240-251.  returns list of (aabb6, entry, exit, shape)."""
    vec = []
    if not nodes:
        return vec
    emp = Aabb.empty(ft).as6()

    def branch(ni, box, next_free):  # create_flat_branch :60-89
        vec.append(None)
        assert len(vec) - 1 == next_free
        after = node(ni, next_free + 1)
        vec[next_free] = (box, next_free + 1, after, NONE)
        return after

    def node(ni, next_free):  # flatten_custom :96-143
        nd = nodes[ni]
        if not nd["leaf"]:
            after_l = branch(nd["l"], nd["l_aabb"], next_free)
            return branch(nd["r"], nd["r_aabb"], after_l)
        vec.append((emp, NONE, next_free + 1, nd["shape"]))
        return next_free + 1

    node(0, 0)
    return vec


def ray_new(o, d, ft=np.float32):  # ray_impl.rs:70-80
    o = [ft(v) for v in o]
    d = [ft(v) for v in d]
    with np.errstate(all="ignore"):
        nrm = np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
        d = [v / nrm for v in d]
        inv = [ft(1.0) / v for v in d]
    return o, d, inv


def ray_hit(ray, box):  # intersect_default.rs:16-37
    o, _, inv = ray
    with np.errstate(all="ignore"):
        lbr = [(box[k] - o[k]) * inv[k] for k in range(3)]
        rtr = [(box[3 + k] - o[k]) * inv[k] for k in range(3)]
    if any(np.isnan(v) for v in lbr) or any(np.isnan(v) for v in rtr):
        return False
    inf = [_min(a, b) for a, b in zip(lbr, rtr)]
    sup = [_max(a, b) for a, b in zip(lbr, rtr)]
    tmin = max(inf)
    tmax = min(sup)
    z = tmin if tmin > 0 else type(tmin)(0)
    return bool(tmax >= z)


def traverse_flat(flat, shape_aabbs, ray):  # flat_bvh.rs:396-431
    hits, i = [], 0
    while i < len(flat):
        box, entry, exit_, shape = flat[i]
        if entry == NONE:
            if ray_hit(ray, list(shape_aabbs[shape])):
                hits.append(shape)
            i = exit_
        elif ray_hit(ray, box):
            i = entry
        else:
            i = exit_
    return hits


def traverse_tree(nodes, shape_aabbs, ray):  # bvh_impl.rs:104-119, bvh_node.rs:288-319
    out = []
    if not nodes:
        return out

    def rec(ni):
        nd = nodes[ni]
        if not nd["leaf"]:
            if ray_hit(ray, nd["l_aabb"]):
                rec(nd["l"])
            if ray_hit(ray, nd["r_aabb"]):
                rec(nd["r"])
        elif ni != 0 or ray_hit(ray, list(shape_aabbs[nd["shape"]])):
            out.append(nd["shape"])

    rec(0)
    return out


# ---- testbase.rs generators, independently restated with Python ints / numpy f32 ----
MASK64 = (1 << 64) - 1


def splitmix64(state):  # testbase.rs:558-564 ; returns (new_state, value)
    state = (state + 0x9E3779B97F4A7C15) & MASK64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return state, z ^ (z >> 31)


def _as_i32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v


def next_point3(state, bounds):  # testbase.rs:567-595
    state, u = splitmix64(state)
    a = ((u >> 32) & 0xFFFFFFFF) - 0x80000000
    b = (u & 0xFFFFFFFF) - 0x80000000
    b64 = b & MASK64
    rot = ((b64 << 6) | (b64 >> 58)) & MASK64
    c = (a & MASK64) ^ rot
    raw = (_as_i32(a), _as_i32(b), _as_i32(c))
    f = np.float32
    out = []
    for k in range(3):
        fv = (f(raw[k]) / f(2147483647) + f(1.0)) * f(0.5)  # i32::MAX as f32 rounds to 2^31
        size = f(bounds[3 + k]) - f(bounds[k])
        out.append(f(bounds[k]) + fv * size)
    return state, out


def ray_triangle(ray, a, b, c):  # ray_impl.rs:154-213, written independently of oracle_impl.inc
    """Möller–Trumbore with back-face culling on numpy scalars of the ray's dtype.
    ray = (origin, direction, inv_direction); returns (distance, u, v) as Intersection::new receives them."""
    o, d = np.asarray(ray[0]), np.asarray(ray[1])
    ft = o.dtype.type
    a, b, c = (np.asarray(p, dtype=ft) for p in (a, b, c))

    def cross(x, y):  # nalgebra cross
        return np.array([ft(x[1] * y[2]) - ft(x[2] * y[1]), ft(x[2] * y[0]) - ft(x[0] * y[2]),
                         ft(x[0] * y[1]) - ft(x[1] * y[0])], dtype=ft)

    def dot(x, y):  # (x0*y0 + x1*y1) + x2*y2
        return ft(ft(ft(x[0] * y[0]) + ft(x[1] * y[1])) + ft(x[2] * y[2]))
    with np.errstate(all="ignore"):
        ab, ac = b - a, c - a
        uvec = cross(d, ac)
        det = dot(ab, uvec)
        if det < np.finfo(ft).eps:
            return ft(np.inf), ft(0), ft(0)
        inv_det = ft(ft(1) / det)
        ao = o - a
        u = ft(dot(ao, uvec) * inv_det)
        if not (ft(0) <= u <= ft(1)):
            return ft(np.inf), u, ft(0)
        vvec = cross(ao, ab)
        v = ft(dot(d, vvec) * inv_det)
        if v < 0 or ft(u + v) > ft(1):
            return ft(np.inf), u, v
        dist = ft(dot(ac, vvec) * inv_det)
        return (dist if dist > np.finfo(ft).eps else ft(np.inf)), u, v


def aabb_min_dist2(box, p, ft=np.float32):  # aabb_impl.rs:618-629, independent restatement
    box = [ft(v) for v in box]
    p = [ft(v) for v in p]
    out = []
    with np.errstate(all="ignore"):
        for k in range(3):
            half = ft((box[3 + k] - box[k]) * ft(0.5))
            centre = ft(box[k] + half)
            q = ft(abs(ft(p[k] - centre)) - half)
            out.append(q if q > 0 else ft(0))
        return ft(ft(out[0] * out[0] + out[1] * out[1]) + out[2] * out[2])


def triangle_dist2(tri, p, ft=np.float32):  # testbase.rs:353-443, independent restatement
    a, b, c = (np.asarray(tri, dtype=ft).reshape(3, 3)[i] for i in range(3))
    p = np.asarray(p, dtype=ft)

    def dot(x, y):
        return ft(ft(ft(x[0] * y[0]) + ft(x[1] * y[1])) + ft(x[2] * y[2]))

    def segment(a, b):
        ab = b - a
        s = ft(dot(ab, p - a) / dot(ab, ab))
        s = ft(0) if s < 0 else (ft(1) if s > 1 else s)
        return a + s * ab

    def closest():
        e_ab, e_bc, e_ac = bool((a == b).all()), bool((b == c).all()), bool((a == c).all())
        if e_ab and e_bc and e_ac:
            return a
        if e_ab:
            return segment(a, c)
        if e_bc or e_ac:
            return segment(a, b)
        ab, ac, ap = b - a, c - a, p - a
        d1, d2 = dot(ab, ap), dot(ac, ap)
        if d1 <= 0 and d2 <= 0:
            return a
        bp = p - b
        d3, d4 = dot(ab, bp), dot(ac, bp)
        if d3 >= 0 and d4 <= d3:
            return b
        cp = p - c
        d5, d6 = dot(ab, cp), dot(ac, cp)
        if d6 >= 0 and d5 <= d6:
            return c
        vc = ft(ft(d1 * d4) - ft(d3 * d2))
        if vc <= 0 and d1 >= 0 and d3 <= 0:
            return a + ft(d1 / ft(d1 - d3)) * ab
        vb = ft(ft(d5 * d2) - ft(d1 * d6))
        if vb <= 0 and d2 >= 0 and d6 <= 0:
            return a + ft(d2 / ft(d2 - d6)) * ac
        va = ft(ft(d3 * d6) - ft(d5 * d4))
        if va <= 0 and ft(d4 - d3) >= 0 and ft(d5 - d6) >= 0:
            return b + ft(ft(d4 - d3) / ft(ft(d4 - d3) + ft(d5 - d6))) * (c - b)
        denom = ft(ft(1) / ft(ft(va + vb) + vc))
        return (a + ft(vb * denom) * ab) + ft(vc * denom) * ac
    with np.errstate(all="ignore"):
        d = p - closest()
        return dot(d, d)


def ray_slice(ray, box):  # ray_impl.rs:118-145 → (tmin, tmax) or None
    o, _, inv = ray
    with np.errstate(all="ignore"):
        lbr = [(box[k] - o[k]) * inv[k] for k in range(3)]
        rtr = [(box[3 + k] - o[k]) * inv[k] for k in range(3)]
    if any(np.isnan(v) for v in lbr) or any(np.isnan(v) for v in rtr):
        return None
    inf = [_min(a, b) for a, b in zip(lbr, rtr)]
    sup = [_max(a, b) for a, b in zip(lbr, rtr)]
    tmin = max(max(inf), type(inf[0])(0))
    tmax = min(sup)
    return None if tmin > tmax else (tmin, tmax)


def traverse_child_ordered(nodes, shape_aabbs, ray, ascending=True):
    """ChildDistanceTraverseIterator (child_distance_traverse.rs) as the RECURSION it unrolls: at every inner node
    test both child boxes, visit the higher-priority hit child first, then the other; a leaf yields its shape."""
    out = []
    if len(nodes) == 0:
        return out
    if nodes[0]["shape"] != 0xFFFFFFFF and not ray_hit(ray, shape_aabbs[nodes[0]["shape"]]):
        return out   # iter_initially_has_node (iter.rs:164-182)

    def visit(ni):
        nd = nodes[ni]
        if nd["shape"] != 0xFFFFFFFF:
            out.append(int(nd["shape"]))
            return
        ls = ray_slice(ray, list(nd["l_min"]) + list(nd["l_max"]))
        rs = ray_slice(ray, list(nd["r_min"]) + list(nd["r_max"]))
        order = []
        if ls and rs:
            right_first = (ls[0] > rs[0]) != (not ascending)
            order = [nd["r"], nd["l"]] if right_first else [nd["l"], nd["r"]]
        elif ls:
            order = [nd["l"]]
        elif rs:
            order = [nd["r"]]
        for c in order:
            visit(int(c))
    visit(0)
    return out


class RustBinaryHeap:
    """alloc::collections::BinaryHeap (Rust std) written with swaps instead of the std's moving hole — the two
    leave the same array behind.  Max-heap on `key`; ties keep whatever the sifts leave (that IS the order
    DistanceTraverseIterator yields equal-distance nodes in)."""

    def __init__(self):
        self.d = []

    def __len__(self):
        return len(self.d)

    def _sift_up(self, pos):
        d = self.d
        while pos > 0:
            parent = (pos - 1) // 2
            if d[pos][0] <= d[parent][0]:
                break
            d[pos], d[parent] = d[parent], d[pos]
            pos = parent

    def push(self, key, val):
        self.d.append((key, val))
        self._sift_up(len(self.d) - 1)

    def pop(self):
        d = self.d
        item = d.pop()
        if not d:
            return item
        item, d[0] = d[0], item          # the former last element now sits at the root
        end, pos = len(d), 0
        while 2 * pos + 2 < end:         # two children: follow the greater one, the right one on ties
            child = 2 * pos + 1
            if d[child][0] <= d[child + 1][0]:
                child += 1
            d[pos], d[child] = d[child], d[pos]
            pos = child
        if 2 * pos + 1 == end - 1:       # a lone left child at the bottom
            d[pos], d[end - 1] = d[end - 1], d[pos]
            pos = end - 1
        self._sift_up(pos)
        return item


def traverse_distance(nodes, shape_aabbs, ray, ascending=True):
    """DistanceTraverseIterator (distance_traverse.rs:40-158): best-first over the BvhNode array."""
    out = []
    if len(nodes) == 0:
        return out
    if nodes[0]["shape"] != 0xFFFFFFFF and not ray_hit(ray, shape_aabbs[nodes[0]["shape"]]):
        return out
    ft = np.asarray(ray[0]).dtype.type
    heap = RustBinaryHeap()
    heap.push(-ft(0) if ascending else ft(0), 0)
    while len(heap):
        _, ni = heap.pop()
        nd = nodes[ni]
        if nd["shape"] != 0xFFFFFFFF:
            out.append(int(nd["shape"]))
            continue
        for child, lo, hi in ((nd["l"], nd["l_min"], nd["l_max"]), (nd["r"], nd["r_min"], nd["r_max"])):
            sl = ray_slice(ray, list(lo) + list(hi))
            if sl is None:
                continue
            heap.push(-sl[0] if ascending else sl[1], int(child))
    return out
