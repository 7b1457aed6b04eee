"""Crate-pinning: if `tools/golden_dump` (a small Rust program over the REAL `bvh` 0.12.0 crate, to be run once on any machine
with cargo) has left its arrays under tests/golden/crate/, the oracle must reproduce them byte for byte — Vec<BvhNode>, the
shape -> node map, the FlatNode array, the create_ray stream and FlatBvh::traverse's per-ray lists IN ORDER.  That upgrades
every oracle-based parity claim from "pinned by two restatements" to "pinned by the crate" (SURVEY §8c).  Without the dump
(this image has no cargo) the tests are skipped and the status stays "partially pinned" (oracle/bvh_oracle.h)."""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DUMP = os.path.join(HERE, "golden", "crate")
MANIFEST = os.path.join(DUMP, "manifest.json")

pytestmark = pytest.mark.skipif(not os.path.exists(MANIFEST), reason="no crate dump: run tools/golden_dump once where cargo exists")


def _sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def _expect(manifest, name, data: bytes):
    ent = manifest[name]
    assert len(data) == ent["bytes"], name
    assert _sha(data) == ent["sha256"], f"{name}: the oracle differs from the bvh crate"
    path = os.path.join(DUMP, name)
    if os.path.exists(path):   # the small scene ships the arrays themselves: say where the first difference is
        want = open(path, "rb").read()
        if want != data:
            a, b = np.frombuffer(want, np.uint8), np.frombuffer(data, np.uint8)
            raise AssertionError(f"{name}: first differing byte at {int(np.flatnonzero(a != b)[0])}")


@pytest.mark.parametrize("n_cubes,n_rays", [(100, 1000), (10_000, 100_000)])
def test_oracle_equals_the_crate(n_cubes, n_rays):
    from oracle import orc
    manifest = json.load(open(MANIFEST))
    tris, aabbs = orc.create_n_cubes(n_cubes)
    _expect(manifest, f"cubes{n_cubes}_aabbs.f32", aabbs.astype("<f4").tobytes())
    t = orc.build(aabbs)
    _expect(manifest, f"cubes{n_cubes}_nodes.bin", t.nodes.tobytes())
    _expect(manifest, f"cubes{n_cubes}_shape_nodes.u32", t.shape_node.astype("<u4").tobytes())
    flat = orc.flatten(t.nodes)
    _expect(manifest, f"cubes{n_cubes}_flat.bin", flat.tobytes())
    rays = orc.create_rays(0, n_rays)
    _expect(manifest, f"cubes{n_cubes}_rays.bin", rays.tobytes())
    off, idx, _, _ = orc.traverse_flat(flat, aabbs, rays, threads=orc.max_threads())
    _expect(manifest, f"cubes{n_cubes}_offsets.u32", off.astype("<u4").tobytes())
    _expect(manifest, f"cubes{n_cubes}_indices.u32", idx.astype("<u4").tobytes())


@pytest.mark.gpu
def test_engine_equals_the_crate():
    """the same arrays straight from the GPU engine (small scene: the dump ships the arrays)"""
    import bvh_amd
    from bvh_amd import Bvh, RayBatch
    from oracle import orc
    manifest = json.load(open(MANIFEST))
    _, aabbs = orc.create_n_cubes(100)
    bvh = Bvh.from_aabbs(aabbs)
    _expect(manifest, "cubes100_nodes.bin", bvh.nodes.tobytes())
    flat = bvh.flatten()
    _expect(manifest, "cubes100_flat.bin", flat.nodes.tobytes())
    rays = orc.create_rays(0, 1000)
    off, idx, _, _ = flat.traverse_batch(RayBatch(len(rays), np.float32, host=rays))
    _expect(manifest, "cubes100_offsets.u32", off.astype("<u4").tobytes())
    _expect(manifest, "cubes100_indices.u32", idx.astype("<u4").tobytes())
