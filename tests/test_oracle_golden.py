"""Pin the CPU oracle against every known-answer vector the reference's own tests hold for the
hot path (tests/golden/reference_known_answers.json, transcribed from /root/reference/src) and
against the independent second restatement oracle/pyref.py.  CPU only.
"""
import json
import os

import numpy as np
import pytest

from oracle import orc, pyref

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


def unit_box(center, dtype=np.float32):
    c = np.asarray(center, dtype=dtype)
    return np.concatenate([c + dtype(-0.5), c + dtype(0.5)]).astype(dtype)


# ---------------------------------------------------------------- 21-box golden hit sets
@pytest.mark.parametrize("parallel", [False, True])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_aligned_boxes_golden_sets(parallel, dtype):
    g = GOLD["aligned_boxes"]
    aabbs = orc.aligned_boxes().astype(dtype)
    tree = orc.build(aabbs, parallel=parallel)
    assert orc.check_tree(tree.nodes, aabbs) == 0
    flat = orc.flatten(tree.nodes)
    assert len(flat) == 3 * 21 - 2
    ids = np.array(g["ids"])
    for case in g["rays"]:
        rays = orc.make_rays([case["origin"]], [case["direction"]], dtype)
        off, idx, _, _ = orc.traverse_flat(flat, aabbs, rays)
        assert sorted(ids[idx].tolist()) == sorted(case["hit_ids"])
        assert len(idx) == len(case["hit_ids"])
        off2, idx2 = orc.traverse_tree(tree.nodes, aabbs, rays)
        assert idx2.tolist() == idx.tolist()  # recursive and flat agree, same DFS order


def test_shape_indices_cover_all_leaves():
    # bvh_impl.rs:590-614
    aabbs = orc.aligned_boxes()
    tree = orc.build(aabbs)
    leaves = tree.nodes[tree.nodes["shape"] != orc.NONE]
    assert sorted(leaves["shape"].tolist()) == list(range(21))
    inner = tree.nodes[tree.nodes["shape"] == orc.NONE]
    assert len(inner) == 20
    # set_bh_node_index: shape_node[i] is the leaf that holds shape i (bvh_node.rs:102)
    for s, ni in enumerate(tree.shape_node):
        assert tree.nodes[ni]["shape"] == s


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_one_node_bvh(dtype):
    for case in GOLD["one_node"]["cases"]:
        aabbs = unit_box(case["box_center"], dtype).reshape(1, 6)
        tree = orc.build(aabbs)
        assert len(tree.nodes) == 1 and tree.nodes[0]["shape"] == 0
        flat = orc.flatten(tree.nodes)
        assert len(flat) == 1 and flat[0]["entry"] == orc.NONE and flat[0]["exit"] == 1
        rays = orc.make_rays([case["origin"]], [case["direction"]], dtype)
        off, idx, _, _ = orc.traverse_flat(flat, aabbs, rays)
        assert len(idx) == case["hits"]
        off2, idx2 = orc.traverse_tree(tree.nodes, aabbs, rays)
        assert len(idx2) == case["hits"]


def test_empty_bvh():
    # bvh_impl.rs:57-59, :563-574 ; flat_bvh.rs:245-248, :620-625
    aabbs = np.zeros((0, 6), dtype=np.float32)
    tree = orc.build(aabbs)
    assert len(tree.nodes) == 0
    assert orc.check_tree(tree.nodes, aabbs) == 0
    flat = orc.flatten(tree.nodes)
    assert len(flat) == 0
    rays = orc.make_rays([[0, 0, 0]], [[1, 0, 0]])
    off, idx, _, _ = orc.traverse_flat(flat, aabbs, rays)
    assert off.tolist() == [0, 0] and len(idx) == 0


# ---------------------------------------------------------------- slab edge cases
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_slab_edge_cases(dtype):
    s = GOLD["slab"]
    z = s["zero_depth"]
    ray = orc.make_rays([z["origin"]], [z["direction"]], dtype)[0]
    assert orc.ray_intersects_aabb(ray, z["aabb"]) is z["hit"]

    a = s["slice_accuracy"]
    pts = np.array(a["grow_points"], dtype=dtype)
    box = np.concatenate([pts.min(axis=0), pts.max(axis=0)])
    ray = orc.make_rays([a["origin"]], [a["direction"]], dtype)[0]
    tmin, tmax = orc.ray_slice(ray, box)
    assert abs(tmin - a["tmin"]) < a["tol"] and abs(tmax - a["tmax"]) < a["tol"]

    p = s["parallel_miss"]
    ray = orc.make_rays([p["origin"]], [p["direction"]], dtype)[0]
    assert orc.ray_slice(ray, unit_box(p["box_center"], dtype)) is None

    for c in s["in_plane"]:
        ray = orc.make_rays([c["origin"]], [c["direction"]], dtype)[0]
        box = unit_box(c["box_center"], dtype)
        assert orc.ray_intersects_aabb(ray, box) is c["hit"]
        assert orc.ray_slice(ray, box) is None


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_aabb_doc_tests(dtype):
    d = GOLD["aabb_doc_tests"]
    assert orc.surface_area(d["box"], dtype) == d["surface_area"]
    assert orc.center(d["box"], dtype).tolist() == d["center"]
    assert orc.largest_axis(d["largest_axis_box"], dtype) == d["largest_axis"]
    # aabb_impl.rs:730-746: center must not overflow for huge boxes (min*0.5 + max*0.5 form)
    big = np.finfo(dtype).max
    c = orc.center([-big, -big, -big, big, big, big], dtype)
    assert np.all(np.isfinite(c))


def test_triangle_intersection_basic():
    # ray_impl.rs:154-213: front-face hit returns distance, back-face is culled
    ray = orc.make_rays([[0.25, 0.25, -1.0]], [[0, 0, 1]])[0]
    a, b, c = [0, 0, 0], [0, 1, 0], [1, 0, 0]  # det > 0 for +z rays
    d, u, v = orc.ray_triangle(ray, a, b, c)
    d2, _, _ = orc.ray_triangle(ray, a, c, b)
    assert (np.isfinite(d) and abs(d - 1.0) < 1e-6 and np.isinf(d2)) or (
        np.isfinite(d2) and abs(d2 - 1.0) < 1e-6 and np.isinf(d))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_triangle_intersection_two_restatements_agree(dtype):
    """oracle_impl.inc orc_ray_triangle vs the independently written oracle/pyref.py ray_triangle: every field
    of Intersection bit-identical, including u / v of the early returns (ray_impl.rs:186-211)."""
    from oracle import pyref
    rng = np.random.default_rng(5)
    n = 3000
    a = rng.uniform(-50, 50, size=(n, 3)).astype(dtype)
    b = a + rng.normal(scale=3, size=(n, 3)).astype(dtype)
    c = a + rng.normal(scale=3, size=(n, 3)).astype(dtype)
    w = rng.uniform(0, 1, size=(n, 2)); w[w.sum(1) > 1] = 1 - w[w.sum(1) > 1]
    p = a + w[:, :1].astype(dtype) * (b - a) + w[:, 1:].astype(dtype) * (c - a)
    o = rng.uniform(-60, 60, size=(n, 3)).astype(dtype)
    d = (p - o).astype(dtype)
    d[:400] = rng.normal(size=(400, 3))
    c[400:500] = b[400:500]                                   # zero-area triangles
    rays = orc.make_rays(o, d, dtype)
    tris = np.stack([a, b, c], axis=1)
    isect, closest, prim = orc.triangle_stage(tris, rays, np.arange(n + 1, dtype=np.uint32), np.arange(n, dtype=np.uint32))
    hits = 0
    for i in range(n):
        pr = pyref.ray_triangle((rays[i]["o"], rays[i]["d"], rays[i]["inv"]), a[i], b[i], c[i])
        got = orc.ray_triangle(rays[i], a[i], b[i], c[i])
        assert np.array([*pr], dtype=dtype).tobytes() == np.array([*got], dtype=dtype).tobytes() == isect[i].tobytes()
        hits += np.isfinite(got[0])
        # one candidate per ray: the closest hit is that candidate iff it is finite
        assert (prim[i] == i) == bool(np.isfinite(got[0]))
    assert hits > n // 3


def test_triangle_stage_closest_is_first_minimum():
    """orc_triangle_stage: closest = the candidate with the smallest distance, the first one on ties."""
    tri = np.array([[0, 0, 0], [0, 1, 0], [1, 0, 0]], dtype=np.float32)
    tris = np.stack([tri + [0, 0, 5], tri + [0, 0, 2], tri + [0, 0, 2], tri + [0, 0, 9]]).astype(np.float32)
    rays = orc.make_rays([[0.25, 0.25, -1.0]], [[0, 0, 1]])
    off = np.array([0, 4], dtype=np.uint32); idx = np.array([0, 1, 2, 3], dtype=np.uint32)
    isect, closest, prim = orc.triangle_stage(tris, rays, off, idx)
    assert isect[:, 0].tolist() == [6.0, 3.0, 3.0, 10.0]
    assert prim[0] == 1 and closest[0, 0] == 3.0
    _, c2, p2 = orc.triangle_stage(tris, rays, np.array([0, 0], dtype=np.uint32), np.zeros(0, dtype=np.uint32))
    assert np.isinf(c2[0, 0]) and c2[0, 1] == 0 and c2[0, 2] == 0 and p2[0] == 0xFFFFFFFF


def test_reference_ray_hits_triangle_property_on_oracle():
    """the reference's proptest test_ray_hits_triangle (ray_impl.rs:361-420) restated for the oracle: a ray
    aimed at a point inside a triangle hits it unless it looks at the back face (culled)."""
    rng = np.random.default_rng(17)
    n = 20000
    f = np.float32
    a, b, c, origin = (rng.uniform(-1e3, 1e3, size=(n, 3)).astype(f) for _ in range(4))
    u = rng.integers(0, 65536, size=n) % 101
    v = np.minimum(100 - u, rng.integers(0, 65536, size=n) % 101)
    uf, vf = u.astype(f) / f(100), v.astype(f) / f(100)
    u_vec, v_vec = b - a, c - a
    normal = np.cross(u_vec, v_vec).astype(f)
    p = a + uf[:, None] * u_vec + vf[:, None] * v_vec
    rays = orc.make_rays(origin, (p - origin).astype(f), f)
    side = (normal.astype(np.float64) * (origin - a).astype(np.float64)).sum(axis=1)
    clear = np.abs(side) > 1e-4 * np.abs(normal.astype(np.float64) * (origin - a)).sum(axis=1)
    r, _, _ = orc.triangle_stage(np.stack([a, b, c], axis=1), rays, np.arange(n + 1, dtype=np.uint32),
                                 np.arange(n, dtype=np.uint32))
    back = side <= 0
    assert np.all(np.isinf(r[back & clear, 0]))
    eps = np.finfo(f).eps
    with np.errstate(all="ignore"):
        uv = r[:, 1] + r[:, 2]
    inside = (uv >= 0) & (uv <= 1) & np.isfinite(r[:, 0])
    border = (np.abs(uf) < eps) | (np.abs(uf - 1) < eps) | (np.abs(vf) < eps) | (np.abs(vf - 1) < eps) | (np.abs(uf + vf - 1) < eps)
    front = ~back & clear
    # points generated on an edge (u, v or u+v within rounding of the border) may fall outside by one ulp
    near_edge = (u == 0) | (v == 0) | (u + v == 100)
    assert np.all((inside | border)[front & ~near_edge])


# ---------------------------------------------------------------- point queries (nearest_to)
def test_reference_nearest_to_doc_tests():
    """aabb_impl.rs:603-614 (min_distance_squared of (20,0,0) to [0,10]^3 is 10^2) and the nearest_to doc example
    of bounding_hierarchy.rs / flat_bvh.rs:440-508: 1000 unit boxes at (i,i,i), query (5.0,5.7,5.3) → id 5."""
    assert np.sqrt(orc.aabb_min_dist2([0, 0, 0, 10, 10, 10], [20, 0, 0])) == 10.0
    pos = np.arange(1000, dtype=np.float32)[:, None].repeat(3, 1)
    aabbs = np.concatenate([pos + np.float32(-0.5), pos + np.float32(0.5)], axis=1)
    t = orc.build(aabbs)
    for h in (orc.flatten(t.nodes), t.nodes):           # FlatBvh and Bvh implementations of the trait
        shape, dist = orc.nearest(h, aabbs, [[5.0, 5.7, 5.3]])
        assert shape[0] == 5 and abs(dist[0] - 0.2) < 1e-6
    # empty hierarchy → None (flat_bvh.rs:518-520, bvh_impl.rs:229-231)
    e = orc.build(np.zeros((0, 6), np.float32))
    assert orc.nearest(orc.flatten(e.nodes), np.zeros((0, 6), np.float32), [[0, 0, 0]])[0][0] == 0xFFFFFFFF


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_point_distance_two_restatements_agree(dtype):
    from oracle import pyref
    rng = np.random.default_rng(8)
    tris = rng.uniform(-5, 5, size=(1500, 3, 3)).astype(dtype)
    tris[:100, 1] = tris[:100, 0]                     # degenerate: a == b
    tris[100:200, 2] = tris[100:200, 1]               # b == c
    tris[200:300, 2] = tris[200:300, 0]               # a == c
    tris[300:350, 1] = tris[300:350, 0]; tris[300:350, 2] = tris[300:350, 0]   # a point
    pts = rng.uniform(-8, 8, size=(1500, 3)).astype(dtype)
    pts[400:500] = tris[400:500, 0]                   # query on a vertex
    for i in range(len(tris)):
        a = orc.triangle_dist2(tris[i], pts[i], dtype)
        b = pyref.triangle_dist2(tris[i], pts[i], dtype)
        assert np.array([a]).tobytes() == np.array([b], dtype=dtype).tobytes() or (np.isnan(a) and np.isnan(b)), i
        box = np.concatenate([tris[i].min(0), tris[i].max(0)])
        assert orc.aabb_min_dist2(box, pts[i], dtype) == pyref.aabb_min_dist2(box, pts[i], dtype)


def test_reference_nearest_to_some_bh_property():
    """testbase.rs:270-312 (nearest_to_some_bh): nearest_to agrees with a brute-force scan, for both trait
    implementations; the distance is what the test pins (several triangles of a cube tie on shared vertices)."""
    tris, aabbs = orc.create_n_cubes(1000)
    t = orc.build(aabbs)
    flat = orc.flatten(t.nodes)
    rng = np.random.default_rng(2)
    pts = np.concatenate([rng.uniform(-1000, 1000, size=(40, 3)), rng.uniform(-1e5, 1e5, size=(40, 3))]).astype(np.float32)
    fs, fd = orc.nearest(flat, aabbs, pts, tris)
    ts, td = orc.nearest(t.nodes, aabbs, pts, tris)
    assert np.array_equal(fd, td)
    for i, p in enumerate(pts):
        d2 = np.array([orc.triangle_dist2(tr, p) for tr in tris[:: 1]])
        assert np.sqrt(d2.min()) == fd[i]
        assert d2[fs[i]] == d2.min() and d2[ts[i]] == d2.min()


# ---------------------------------------------------------------- scene generators
def test_scene_shape_and_invariants_1200():
    tris, aabbs = orc.create_n_cubes(100)
    assert len(tris) == GOLD["scene_shape"]["cubes_to_triangles"]["100"]
    tree = orc.build(aabbs)
    assert len(tree.nodes) == 2 * 1200 - 1
    assert orc.check_tree(tree.nodes, aabbs) == 0  # is_consistent + assert_tight + coverage
    flat = orc.flatten(tree.nodes)
    assert len(flat) == 3 * 1200 - 2
    par = orc.build(aabbs, parallel=True)
    assert par.nodes.tobytes() == tree.nodes.tobytes()  # executor-independent (bvh_node.rs:138-142)
    assert np.array_equal(par.shape_node, tree.shape_node)
    # cube triangles are unit-sized and inside the (slightly grown) bounds
    ext = aabbs[:, 3:] - aabbs[:, :3]
    assert np.all(ext <= 1.0 + 1e-2) and np.all(aabbs[:, :3] >= -100001) and np.all(aabbs[:, 3:] <= 100001)


def test_generators_match_independent_restatement():
    state = 0
    bounds = [float(v) for v in orc.DEFAULT_BOUNDS]
    tris, _ = orc.create_n_cubes(5)
    for i in range(5):
        state, p = pyref.next_point3(state, bounds)
        # top_front_right = pos + (0.5, 0.5, -0.5) is vertex 1 of triangle 0 (testbase.rs:491,500-504)
        tfr = tris[i * 12, 1]
        exp = [p[0] + np.float32(0.5), p[1] + np.float32(0.5), p[2] + np.float32(-0.5)]
        assert tfr.tolist() == [float(v) for v in exp]
    # ray stream, including O(1) seek
    rays = orc.create_rays(0, 8)
    rays_tail = orc.create_rays(5, 3)
    assert rays[5:].tobytes() == rays_tail.tobytes()
    state = 0
    for i in range(8):
        state, o = pyref.next_point3(state, bounds)
        state, d = pyref.next_point3(state, bounds)
        ro, rd, rinv = pyref.ray_new(o, d)
        assert rays[i]["o"].tolist() == [float(v) for v in ro]
        assert rays[i]["d"].tolist() == [float(v) for v in rd]
        assert rays[i]["inv"].tolist() == [float(v) for v in rinv]
    # bench quirk: scene and ray streams share seed 0, so ray k's origin is cube 2k's centre
    assert np.allclose(rays[1]["o"], tris[2 * 12:3 * 12].reshape(-1, 3).mean(axis=0), atol=1e-2)


# ---------------------------------------------------------------- C oracle vs independent pyref
def _nodes_equal(c_nodes, py_nodes):
    assert len(c_nodes) == len(py_nodes)
    for cn, pn in zip(c_nodes, py_nodes):
        assert cn["parent"] == pn["parent"]
        if pn["leaf"]:
            assert cn["shape"] == pn["shape"]
        else:
            assert cn["shape"] == orc.NONE and cn["l"] == pn["l"] and cn["r"] == pn["r"]
            assert np.concatenate([cn["l_min"], cn["l_max"]]).tolist() == [float(v) for v in pn["l_aabb"]]
            assert np.concatenate([cn["r_min"], cn["r_max"]]).tolist() == [float(v) for v in pn["r_aabb"]]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_c_oracle_equals_pyref_on_cubes(dtype):
    _, aabbs = orc.create_n_cubes(25)  # 300 triangles, includes 150 degenerate pair splits
    aabbs = aabbs.astype(dtype)
    tree = orc.build(aabbs)
    py_nodes, py_shape_node = pyref.build(aabbs)
    _nodes_equal(tree.nodes, py_nodes)
    assert tree.shape_node.tolist() == py_shape_node
    flat = orc.flatten(tree.nodes)
    py_flat = pyref.flatten(py_nodes, dtype)
    assert len(flat) == len(py_flat)
    for cf, pf in zip(flat, py_flat):
        assert (cf["entry"], cf["exit"], cf["shape"]) == pf[1:]
        assert np.concatenate([cf["min"], cf["max"]]).tolist() == [float(v) for v in pf[0]]
    rays = orc.create_rays(0, 40).astype(orc.RAY_F32)
    if dtype == np.float64:
        rays = orc.make_rays(rays["o"], rays["d"], np.float64)
    off, idx, _, _ = orc.traverse_flat(flat, aabbs, rays)
    for i in range(len(rays)):
        ray = ([dtype(v) for v in rays[i]["o"]], None, [dtype(v) for v in rays[i]["inv"]])
        exp = pyref.traverse_flat(py_flat, aabbs, ray)
        assert idx[off[i]:off[i + 1]].tolist() == exp
        assert pyref.traverse_tree(py_nodes, aabbs, ray) == exp


def test_c_oracle_equals_pyref_random_boxes():
    rng = np.random.default_rng(7)
    lo = rng.uniform(-50, 50, size=(257, 3)).astype(np.float32)
    ext = rng.uniform(0, 4, size=(257, 3)).astype(np.float32)
    ext[::7] = 0  # zero-thickness boxes
    lo[100:110] = lo[100]  # identical centroids → degenerate split path (bvh_node.rs:114-124)
    ext[100:110] = ext[100]
    aabbs = np.concatenate([lo, lo + ext], axis=1)
    tree = orc.build(aabbs)
    assert orc.check_tree(tree.nodes, aabbs) == 0
    py_nodes, py_sn = pyref.build(aabbs)
    _nodes_equal(tree.nodes, py_nodes)
    assert tree.shape_node.tolist() == py_sn


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_scalable_build_schedule_is_byte_equal_to_the_serial_build(dtype):
    """bench.py's cpu_baseline times orc.build(schedule="fast") (team-split big nodes, then a parallel for over the subtrees;
    VERDICT r4 #5).  It is a schedule, not another algorithm: BvhNode array and shape → node map must carry the bits of the serial
    recursion (bvh_node.rs:81-279) for every team size — balanced scenes, sizes around its cut-off, colliding centroids (the
    halving branch, :114-124), a long chain of big nodes (its work lists grow) and the task-parallel restatement beside it."""
    _, cubes = orc.create_n_cubes(2500)
    scenes = [cubes[:m].astype(dtype) for m in (0, 1, 2, 33, 4096, 4097, 8200, 30000)]
    n = 12000
    x = np.float32(1.004) ** np.arange(n, dtype=np.float32)
    lo = np.stack([x, np.zeros(n, np.float32), np.zeros(n, np.float32)], axis=1)
    scenes.append(np.concatenate([lo, lo + np.float32(0.5)], axis=1).astype(dtype))       # unbalanced: ~2 900 big nodes in a chain
    scenes.append(np.tile(np.array([[1, 2, 3, 4, 5, 6]], dtype), (9000, 1)))                # every centroid equal
    scenes.append(np.concatenate([scenes[-1][:5000], cubes[:6000].astype(dtype)]))          # a degenerate cluster inside a scene
    for a in scenes:
        ser = orc.build(a)
        for th in (1, 2, 5, 8):
            par = orc.build(a, threads=th, schedule="fast")
            assert par.nodes.tobytes() == ser.nodes.tobytes(), (len(a), th)
            assert np.array_equal(par.shape_node, ser.shape_node)
        tasks = orc.build(a, threads=4)
        assert tasks.nodes.tobytes() == ser.nodes.tobytes()


# ---------------------------------------------------------------- closed-form flatten
def test_flatten_closed_form():
    _, aabbs = orc.create_n_cubes(100)
    tree = orc.build(aabbs)
    flat = orc.flatten(tree.nodes)
    nodes = tree.nodes
    n_nodes = len(nodes)
    is_leaf = nodes["shape"] != orc.NONE
    leaves_before = np.concatenate([[0], np.cumsum(is_leaf)[:-1]])
    # k_i via pre-order subtree sizes
    k = np.zeros(n_nodes, dtype=np.int64)
    for i in range(n_nodes - 1, -1, -1):
        k[i] = 1 if is_leaf[i] else k[nodes[i]["l"]] + k[nodes[i]["r"]]
    for i in range(1, n_nodes):
        nav = i - 1 + leaves_before[i]
        assert flat[nav]["entry"] == nav + 1
        assert flat[nav]["exit"] == nav + 3 * k[i] - 1
        assert flat[nav]["shape"] == orc.NONE
        if is_leaf[i]:
            assert flat[nav + 1]["entry"] == orc.NONE and flat[nav + 1]["exit"] == nav + 2
            assert flat[nav + 1]["shape"] == nodes[i]["shape"]
            assert np.all(np.isinf(flat[nav + 1]["min"])) and np.all(flat[nav + 1]["max"] == -np.inf)


def test_workload_statistics_120k():
    """Re-derives the statistics SURVEY §8d quotes from a scratch script (depth 22, 60 000 degenerate splits)."""
    _, aabbs = orc.create_n_cubes(10_000)
    tree = orc.build(aabbs, parallel=False)
    st = orc.tree_stats(tree.nodes, aabbs)
    assert st["max_depth"] == 22 and st["degenerate_splits"] == 60_000
    assert abs(st["mean_leaf_depth"] - 17.87) < 0.01
    assert orc.check_tree(tree.nodes, aabbs) == 0
    flat = orc.flatten(tree.nodes)
    assert len(flat) == 359_998
    rays = orc.create_rays(0, 20_000)
    off, idx, _, stats = orc.traverse_flat(flat, aabbs, rays, threads=orc.max_threads())
    assert stats["hits"] == 10_000  # first 5 000 rays start inside a cube and return exactly 2 candidates
    assert np.all(np.diff(off)[:5000] == 2) and np.all(np.diff(off)[5000:] == 0)


# ---------------------------------------------------------------- ordered traversal
def test_ordered_iterator_restatements_agree_and_match_golden_sets():
    """oracle (state machine of child_distance_traverse.rs) vs pyref (the recursion it unrolls), both directions;
    on the 21 aligned boxes the sets are the reference's golden hit sets (testbase.rs:174-225) and the entry
    distances come out sorted, as the reference's own iterator tests assert."""
    from oracle import pyref
    aabbs = orc.aligned_boxes()
    t = orc.build(aabbs)
    for case in GOLD["aligned_boxes"]["rays"]:
        rays = orc.make_rays([case["origin"]], [case["direction"]])
        for asc in (True, False):
            off, idx = orc.traverse_child_ordered(t.nodes, aabbs, rays, asc)
            ids = [int(i) - 10 for i in idx]
            assert sorted(ids) == sorted(case["hit_ids"])
            entry = [orc.ray_slice(rays[0], aabbs[i])[0] for i in idx]
            assert entry == sorted(entry, reverse=not asc)
            assert pyref.traverse_child_ordered(t.nodes, aabbs, (rays[0]["o"], rays[0]["d"], rays[0]["inv"]), asc) == idx.tolist()
    tris, ab = orc.create_n_cubes(150)
    tt = orc.build(ab)
    rng = np.random.default_rng(4)
    c = tris.reshape(150, 36, 3).mean(axis=1)
    o = rng.uniform(-1e5, 1e5, size=(200, 3)).astype(np.float32)
    rays = orc.make_rays(o, (c[rng.integers(0, 150, 200)] - o).astype(np.float32))
    foff, fidx, _, _ = orc.traverse_flat(orc.flatten(tt.nodes), ab, rays)
    for asc in (True, False):
        off, idx = orc.traverse_child_ordered(tt.nodes, ab, rays, asc)
        assert np.array_equal(off, foff)
        for i in range(len(rays)):
            assert pyref.traverse_child_ordered(tt.nodes, ab, (rays[i]["o"], rays[i]["d"], rays[i]["inv"]), asc) == \
                idx[off[i]:off[i + 1]].tolist()
            assert sorted(idx[off[i]:off[i + 1]]) == sorted(fidx[foff[i]:foff[i + 1]])


def test_distance_iterator_restatements_agree_and_match_reference_tests():
    """DistanceTraverseIterator (distance_traverse.rs): oracle (C, std BinaryHeap restated with the moving hole) vs
    pyref (swap-based heap), both directions.  The reference's own tests for it: golden hit sets on the 21 aligned
    boxes with monotone entry / exit distances (:188-217, rays :224-262), the empty tree (:270-281), the single-node
    trees (bvh_impl.rs:667-690) and the three overlapping boxes whose nearest order must be sorted (:295-322)."""
    from oracle import pyref
    aabbs = orc.aligned_boxes()
    t = orc.build(aabbs)
    for case in GOLD["aligned_boxes"]["rays"]:
        rays = orc.make_rays([case["origin"]], [case["direction"]])
        r3 = (rays[0]["o"], rays[0]["d"], rays[0]["inv"])
        for asc in (True, False):
            off, idx = orc.traverse_distance(t.nodes, aabbs, rays, asc)
            assert sorted(int(i) - 10 for i in idx) == sorted(case["hit_ids"])
            d = [orc.ray_slice(rays[0], aabbs[i])[0] for i in idx]       # the test compares ENTRY distances both ways (:199-207)
            assert d == sorted(d, reverse=not asc)
            assert pyref.traverse_distance(t.nodes, aabbs, r3, asc) == idx.tolist()
    # empty tree: both iterators are empty
    e = orc.build(np.zeros((0, 6), np.float32))
    rays = orc.make_rays([[0, 0, 0]], [[1, 0, 0]])
    for asc in (True, False):
        off, idx = orc.traverse_distance(e.nodes, np.zeros((0, 6), np.float32), rays, asc)
        assert off.tolist() == [0, 0] and len(idx) == 0
    # one shape: the root leaf is pre-tested with the shape's AABB (bvh_impl.rs:667-690)
    one = np.array([[-1, -1, -1, 1, 1, 1]], np.float32) + np.array([0, 0, 0, 0, 0, 0], np.float32)
    t1 = orc.build(one)
    miss = orc.make_rays([[0, 2, 0]], [[1, 0, 0]])
    hit = orc.make_rays([[-5, 0, 0]], [[1, 0, 0]])
    assert len(orc.traverse_distance(t1.nodes, one, miss)[1]) == 0
    assert orc.traverse_distance(t1.nodes, one, hit)[1].tolist() == [0]
    # test_overlapping_child_order (:295-322)
    ov = np.array([[-0.33333334, -5000.3335, -5000.3335, 1.3333334, 0.33333334, 0.33333334],
                   [-5000.3335, -5000.3335, -5000.3335, 0.33333334, 0.33333334, -4998.6665],
                   [-5000.3335, -5000.3335, -5000.3335, 0.33333334, 0.33333334, 5000.3335]], np.float32)
    to = orc.build(ov)
    ray = orc.make_rays([[-5000.0, -5000.0, -5000.0]], [[1, 0, 0]])
    off, idx = orc.traverse_distance(to.nodes, ov, ray, True)
    assert sorted(idx.tolist()) == [0, 1, 2]
    d = [orc.ray_slice(ray[0], ov[i])[0] for i in idx]
    assert d == sorted(d)
    # random scene, rays aimed at cubes: same sets as FlatBvh::traverse, two restatements agree on the order
    tris, ab = orc.create_n_cubes(150)
    tt = orc.build(ab)
    rng = np.random.default_rng(5)
    c = tris.reshape(150, 36, 3).mean(axis=1)
    o = rng.uniform(-1e5, 1e5, size=(200, 3)).astype(np.float32)
    rays = orc.make_rays(o, (c[rng.integers(0, 150, 200)] - o).astype(np.float32))
    foff, fidx, _, _ = orc.traverse_flat(orc.flatten(tt.nodes), ab, rays)
    for asc in (True, False):
        off, idx, peak = orc.traverse_distance(tt.nodes, ab, rays, asc, want_peak=True)
        assert np.array_equal(off, foff) and peak >= 2
        for i in range(len(rays)):
            got = idx[off[i]:off[i + 1]].tolist()
            assert pyref.traverse_distance(tt.nodes, ab, (rays[i]["o"], rays[i]["d"], rays[i]["inv"]), asc) == got
            assert sorted(got) == sorted(fidx[foff[i]:foff[i + 1]])
    # f64 as well
    t64 = orc.build(ab.astype(np.float64))
    r64 = orc.make_rays(rays["o"][:50], rays["d"][:50], np.float64)
    for asc in (True, False):
        off, idx = orc.traverse_distance(t64.nodes, ab.astype(np.float64), r64, asc)
        for i in range(len(r64)):
            assert pyref.traverse_distance(t64.nodes, ab.astype(np.float64), (r64[i]["o"], r64[i]["d"], r64[i]["inv"]), asc) == \
                idx[off[i]:off[i + 1]].tolist()


def test_harness_loop_equals_the_staged_restatement():
    """orc.harness_loop (bench.py's CPU figure beside the harness entries) is intersect_bh written out whole — create_ray (or a primary ray),
    FlatBvh::traverse into a growable list, intersects_triangle on every candidate (testbase.rs:819-837).  It must agree with the pieces the
    parity legs use: the same candidates in total as traverse_flat finds on the generator's rays, and the same checksum — sum of the candidates'
    shape indices + the number of finite distances — as triangle_stage computes on that CSR.  Any team size, any start of the stream."""
    tris, aabbs = orc.create_n_cubes(400)
    flat = orc.flatten(orc.build(aabbs).nodes)
    bounds = orc.DEFAULT_BOUNDS
    for first, n in ((0, 4000), (4321, 2500)):
        rays = orc.create_rays(first, n, bounds)
        off, idx, _, st = orc.traverse_flat(flat, aabbs, rays)
        isect, _, _ = orc.triangle_stage(tris, rays, off, idx)
        want = int(idx.astype(np.uint64).sum() + np.isfinite(isect[:, 0]).sum())
        for th in (1, 3):
            total, ck = orc.harness_loop(flat, aabbs, tris, first, n, bounds, threads=th)
            assert total == st["hits"] == len(idx) and ck == want
    # primary rays of a camera (configs[2]'s generator): the 21 aligned unit boxes (testbase.rs:109-116), one triangle across each
    boxes = orc.aligned_boxes()
    tris = np.stack([boxes[:, :3], boxes[:, 3:], np.stack([boxes[:, 3], boxes[:, 1], boxes[:, 2]], axis=1)], axis=1).astype(np.float32)   # (wound to face the camera: ray_impl.rs:186 culls the other side)
    flat = orc.flatten(orc.build(boxes).nodes)
    cam = np.array([0.0, 0.0, -12.0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 1.0, 0.1], dtype=np.float32)
    W, H = 128, 32
    rays = orc.primary_rays(cam, W, H, 50, 4000)
    off, idx, _, st = orc.traverse_flat(flat, boxes, rays)
    isect, _, _ = orc.triangle_stage(tris, rays, off, idx)
    total, ck = orc.harness_loop(flat, boxes, tris, 50, 4000, bounds, cam=cam, width=W, height=H, threads=2)
    assert total == len(idx) and ck == int(idx.astype(np.uint64).sum() + np.isfinite(isect[:, 0]).sum())
    assert total > 1000 and np.isfinite(isect[:, 0]).sum() > 100
