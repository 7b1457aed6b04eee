"""Runs INSIDE a process that has libasan preloaded (tests/test_sanitizers_cpu.py starts it): the host-side code of the repo
under AddressSanitizer + UndefinedBehaviorSanitizer —
  * the product's OBJ loader (bvh_amd/csrc/obj.cpp, built alone as oracle/_san/libobj_san.so) on the fixtures and error cases of
    tests/test_scene_cpu.py and on the procedural atrium, against the Python restatement of the reference's loader;
  * the C oracle (oracle/_san/liboracle_san.so) through the whole of tests/test_oracle_golden.py.
Any sanitizer report aborts the process (-fno-sanitize-recover, ASAN halt_on_error)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SAN = os.path.join(ROOT, "oracle", "_san")


def obj_checks():
    from oracle import objref
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["BVH_AMD_NO_TORCH"] = "1"
    import test_scene_cpu as ts            # fixtures only (its tests go through the full engine library)
    lib = C.CDLL(os.path.join(SAN, "libobj_san.so"))
    lib.bvhgpu_obj_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_size_t), C.c_void_p]
    lib.bvhgpu_obj_free.argtypes = [C.POINTER(C.c_float)]
    lib.bvhgpu_obj_last_error.restype = C.c_char_p
    lib.bvhgpu_triangles_aabbs_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]

    def parse(text):
        data = text.encode() if isinstance(text, str) else bytes(text)
        out, n, bounds = C.POINTER(C.c_float)(), C.c_size_t(), np.zeros(6, np.float32)
        rc = lib.bvhgpu_obj_parse(data, len(data), C.byref(out), C.byref(n), bounds.ctypes.data_as(C.c_void_p))
        if rc != 0:
            return rc, lib.bvhgpu_obj_last_error().decode(), None
        tris = np.ctypeslib.as_array(out, shape=(n.value * 9,)).copy().reshape(-1, 3, 3) if n.value else np.zeros((0, 3, 3), np.float32)
        lib.bvhgpu_obj_free(out)
        aabbs = np.zeros((len(tris), 6), np.float32)
        if len(tris):
            assert lib.bvhgpu_triangles_aabbs_f32(tris.ctypes.data_as(C.c_void_p), len(tris), aabbs.ctypes.data_as(C.c_void_p)) == 0
        return 0, tris, aabbs

    rc, tris, aabbs = parse(ts.FIXTURE)
    rt, ra, _ = objref.parse_obj(ts.FIXTURE)
    assert rc == 0 and tris.tobytes() == rt.tobytes() and aabbs.tobytes() == ra.tobytes()
    n_bad = 0
    for bad in ("v 0 0\n", "v 0 0 0\nf 1 2 3\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/1 2 3\n", "v 0 0 0\nf 1\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 0\n",
                "curv2 1 2\n", "v 0 0 zero\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 -4\n"):   # the error cases of tests/test_scene_cpu.py
        assert parse(bad)[0] != 0, bad
        n_bad += 1
    for bad in ("f 1 2 3\n", "v 1 2\nf 1 1 1\n", "v 0 0 0\nf 1 2 3\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/ 2 3\n", "v a b c\n",
                "v 0 0 0\nf 0 0 0\n", "f\n", "v 1e999999 0 0\nv 0 0 0\nv 0 1 0\nf 1 2 3\n", "\\\n", "f 1 2 3 \\"):
        parse(bad)                         # whatever the verdict: no out-of-bounds read, no UB
        n_bad += 1
    for text in ("", "\n\n", "# only a comment", "v 0 0 0", "v 0 0 0\n" * 5000 + "f " + " ".join(str(i + 1) for i in range(5000)) + "\n"):
        assert parse(text)[0] == 0
    from bvh_amd import scene
    text = scene.make_atrium_obj(4)
    rc, tris, aabbs = parse(text)
    assert rc == 0 and tris.tobytes() == objref.parse_obj(text)[0].tobytes() and len(aabbs) == len(tris)
    # truncated copies of a real file: every prefix must parse or fail cleanly
    data = text.encode()
    for cut in range(0, min(len(data), 4000), 37):
        parse(data[:cut])
    print(f"obj loader under ASan/UBSan: ok ({len(tris)} atrium triangles, {n_bad} malformed inputs)")


def oracle_checks():
    from oracle import orc
    orc._lib = orc._load(os.path.join(SAN, "liboracle_san.so"))
    import pytest
    rc = pytest.main(["-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_golden.py")])
    assert rc == 0, rc
    print("oracle under ASan/UBSan: ok")


if __name__ == "__main__":
    obj_checks()
    oracle_checks()
