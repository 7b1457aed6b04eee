"""Host-side ingest (no GPU): the C++ OBJ loader in libbvh_mi355x.so against the Python restatement of the
reference's `load_obj::<Triangle>` + fan triangulation (testbase.rs:445-487), on hand-written fixtures that
cover the four polygon formats, n-gons, relative indices, comments, continuations and the error cases, and on
the procedural atrium (stand-in for the reference's missing media/sponza.obj)."""
import numpy as np
import pytest

from bvh_amd import BvhGpuError, scene
from oracle import objref

FIXTURE = """# a cube corner, every face format
mtllib none.mtl
o thing
v 0 0 0
v 1 0 0 1.0
v 1 1 0
v 0 1 0   # trailing comment
v 0 0 1
v 1e0 0.0 1.
v +1 1 1
v -0 1 1
vt 0.5 0.5
vn 0 0 1
g faces
usemtl m
s off
f 1 2 3 4
f 5/1 6/1 7/1 8/1
f 1//1 5//1 8//1 4//1
f -8/1/1 -7/1/1 -3/1/1 \\
  -4/1/1
f 2 3 7 6 5
f 3 4
l 1 2
p 1
f 4 8 7
"""


def test_obj_fixture_matches_reference_restatement():
    t, a, b = scene.parse_obj(FIXTURE)
    rt, ra, rb = objref.parse_obj(FIXTURE)
    assert t.shape == (2 + 2 + 2 + 2 + 3 + 0 + 1, 3, 3)
    assert t.tobytes() == rt.tobytes() and a.tobytes() == ra.tobytes() and b.tobytes() == rb.tobytes()
    # fan order (testbase.rs:461-469): quad 1 2 3 4 → (1,2,3), (1,3,4)
    p = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], dtype=np.float32)
    assert np.array_equal(t[0], p[[0, 1, 2]]) and np.array_equal(t[1], p[[0, 2, 3]])
    # pentagon → 3 triangles sharing the anchor
    assert np.array_equal(t[8][0], t[9][0]) and np.array_equal(t[9][0], t[10][0])
    # bytes input works too
    assert scene.parse_obj(FIXTURE.encode())[0].tobytes() == t.tobytes()


@pytest.mark.parametrize("bad,why", [
    ("v 0 0\n", "vertex"), ("v 0 0 0\nf 1 2 3\n", "range"), ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/1 2 3\n", "mixes"),
    ("v 0 0 0\nf 1\n", "two"), ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 0\n", "range"), ("curv2 1 2\n", "unexpected"),
    ("v 0 0 zero\n", "vertex"), ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 -4\n", "range"), ("v 0 0 0\nv 1 0 0\nf 1/ 2/\n", None),
])
def test_obj_errors_like_the_reference_loader(bad, why):
    try:
        objref.parse_obj(bad)
        ref_ok = True
    except objref.ObjError:
        ref_ok = False
    if ref_ok:
        scene.parse_obj(bad)
    else:
        with pytest.raises(BvhGpuError) as e:
            scene.parse_obj(bad)
        if why:
            assert why in str(e.value)


def test_obj_empty_and_whitespace():
    for txt in ("", "\n\n# only comments\n", "   \t\r\n"):
        t, a, b = scene.parse_obj(txt)
        assert len(t) == 0 and len(a) == 0 and np.all(np.isinf(b))


def test_atrium_standin_ingest_matches_restatement():
    txt = scene.make_atrium_obj(1)
    t, a, b = scene.parse_obj(txt)
    rt, ra, rb = objref.parse_obj(txt)
    assert len(t) > 4000
    assert t.tobytes() == rt.tobytes() and a.tobytes() == ra.tobytes() and b.tobytes() == rb.tobytes()
    assert np.allclose(b, [-18, 0, -7, 18, 16, 7])
    # triangle AABBs are Triangle::new's empty.grow(a).grow(b).grow(c) (testbase.rs:325-333)
    assert np.array_equal(a[:, :3], t.min(axis=1)) and np.array_equal(a[:, 3:], t.max(axis=1))
    # deterministic
    assert scene.make_atrium_obj(1) == txt
    big = scene.parse_obj(scene.make_atrium_obj(4))[0]
    assert len(big) > 3 * len(t)
