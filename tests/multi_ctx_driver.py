"""Driver of tests/test_gpu_multictx.py (run as a subprocess on the GPU box): the multi-GPU protocol of the C ABI —
bvhgpu_comm_init_all, bvhgpu_bcast, bvhgpu_bcast_known, peers' trees, status header, rebroadcast — with K ranks.

    python tests/multi_ctx_driver.py <K> [real]

Default: the K ranks are K ctxs that SHARE device 0 and the collective library is tests/c_abi/libfakerccl.so (a
single-process stand-in whose broadcast is an event-ordered device copy; real RCCL wants one GPU per rank).  `real`: ctx i on
device i with the real librccl — for a multi-GPU node.  Prints one JSON object; every check is against the oracle."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    real = len(sys.argv) > 2 and sys.argv[2] == "real"
    if not real:
        os.environ["BVHGPU_RCCL_LIB"] = os.path.join(ROOT, "tests", "c_abi", "libfakerccl.so")
        os.environ["BVHGPU_RCCL_SHARED_DEVICE"] = "1"
    import torch
    import bvh_amd
    from bvh_amd import Bvh, Context, RayBatch, dist as bdist, testbase as tb
    from bvh_amd._lib import INVALID_ARG, NOT_FLATTENED, OK, RAY_F32, REBROADCAST, BvhGpuError
    from bvh_amd.api import _Hits
    from oracle import orc

    out = {"K": K, "real": real}
    ndev = bvh_amd.device_count()
    devs = [i if real else 0 for i in range(K)]
    assert max(devs) < ndev
    ctxs = [Context(d) for d in devs]
    comm = bdist.LocalCommunicator(ctxs)
    bounds = tb.default_bounds()
    _, aabbs = tb.create_n_cubes(2500)          # 30 000 triangles: level tier + workgroup tier + wave tier
    n = len(aabbs)
    T = 90_000                                   # rays of the job, strong-sharded over the ranks
    shards = [bdist.strong_shard(r, K, T) for r in range(K)]
    ot = orc.build(aabbs)
    oflat = orc.flatten(ot.nodes)
    ooff, oidx, _, _ = orc.traverse_flat(oflat, aabbs, orc.create_rays(0, T), threads=orc.max_threads())

    def dev_of(i):
        return torch.device("cuda", devs[i])

    rays, bufs = [], []
    for i, (first, cnt) in enumerate(shards):
        with torch.cuda.device(dev_of(i)):
            buf = torch.empty(max(cnt, 1) * RAY_F32.itemsize, dtype=torch.uint8, device=dev_of(i))
            bufs.append(buf)
            rays.append(RayBatch.generate(first, cnt, bounds, buf, np.float32, ctxs[i]))

    def gather(trees, hits):
        """every rank's CSR (already waited for) → one CSR over the job's rays"""
        offs, idxs, base = [np.zeros(1, np.uint32)], [], 0
        for i, (first, cnt) in enumerate(shards):
            off, idx = hits[i].fetch(cnt)
            offs.append(off[1:] + np.uint32(base)); idxs.append(idx); base += len(idx)
        return np.concatenate(offs), np.concatenate(idxs) if idxs else np.zeros(0, np.uint32)

    # ---- 1. synchronous root, both forms of the broadcast, every rank traverses its shard -------------------------------
    aabbs_dev = torch.from_numpy(aabbs).to(dev_of(0))
    root = Bvh.from_aabbs(aabbs_dev, ctxs[0])
    trees = [root] + [None] * (K - 1)
    rc_unflat = None
    try:
        comm.bcast(trees, 0)                      # not flattened: the root's call fails, nobody hangs, no peer tree is left behind
    except BvhGpuError as e:
        rc_unflat = e.status
    out["unflattened_root_status"] = rc_unflat
    out["unflattened_root_expected"] = NOT_FLATTENED
    root.flatten_in_place()
    trees = comm.bcast([root] + [None] * (K - 1), 0)                     # header form
    hits = [_Hits(c) for c in ctxs]
    for i in range(K):
        trees[i].traverse_async(rays[i], hits[i])
    for i in range(K):
        hits[i].wait()
    off, idx = gather(trees, hits)
    out["bcast_header_form_equal"] = bool(np.array_equal(off, ooff) and np.array_equal(idx, oidx))
    trees = comm.bcast(trees, 0, "f32", n)                                # known form, peers' trees reused
    for i in range(K):
        trees[i].traverse_async(rays[i], hits[i])
    for i in range(K):
        hits[i].wait()
    off, idx = gather(trees, hits)
    out["bcast_known_equal"] = bool(np.array_equal(off, ooff) and np.array_equal(idx, oidx))

    # ---- 2. the asynchronous step: rebuild_async → bcast_known → traverse_async on every rank → waits; no host sync between ----
    ok = True
    for _ in range(3):
        root.rebuild_async(aabbs_dev)
        trees = comm.bcast(trees, 0, "f32", n)
        for i in range(K):
            trees[i].traverse_async(rays[i], hits[i])
        for i in range(K):
            hits[i].wait()
        off, idx = gather(trees, hits)
        ok = ok and bool(np.array_equal(off, ooff) and np.array_equal(idx, oidx))
    out["async_step_equal"] = ok
    out["root_nodes_equal"] = bool(root.nodes.tobytes() == ot.nodes.tobytes())

    # ---- 3. the root announces something its tree is not: its call returns the reason, the peers' WAIT returns INVALID_ARG ----
    st = {}
    trees = comm.bcast(trees, 0, "f32", n + 1, raise_on_error=False)
    st["root_call"] = comm.last_status
    peer_status = []
    for i in range(1, K):
        try:
            trees[i].traverse_async(rays[i], hits[i]); hits[i].wait(); peer_status.append(OK)
        except BvhGpuError as e:
            peer_status.append(e.status)
    st["peers"] = peer_status
    out["wrong_announcement"] = st
    out["wrong_announcement_ok"] = bool(comm.last_status == INVALID_ARG and all(s == INVALID_ARG for s in peer_status))

    # ---- 4. invalid input discovered by the asynchronous build AFTER the broadcast was enqueued --------------------------
    bad = aabbs.copy(); bad[1234, 4] = np.nan
    root.rebuild_async(torch.from_numpy(bad).to(dev_of(0)))
    trees = comm.bcast(trees, 0, "f32", n, raise_on_error=False)
    st = {"root_call": comm.last_status, "peers": []}
    try:
        root.wait(); st["root_wait"] = OK
    except BvhGpuError as e:
        st["root_wait"] = e.status
    for i in range(1, K):
        try:
            trees[i].traverse_async(rays[i], hits[i]); hits[i].wait(); st["peers"].append(OK)
        except BvhGpuError as e:
            st["peers"].append(e.status)
    out["nan_input"] = st
    out["nan_input_ok"] = bool(st["root_call"] == OK and st["root_wait"] == INVALID_ARG and all(s == INVALID_ARG for s in st["peers"]))

    # ---- 5. an unbalanced tree on a first asynchronous build: REBROADCAST on every rank, then the repeat succeeds --------
    x = np.float32(1.004) ** np.arange(12000, dtype=np.float32)
    lo = np.stack([x, np.zeros_like(x), np.zeros_like(x)], axis=1)
    chain = np.concatenate([lo, lo + np.float32(0.5)], axis=1).astype(np.float32)
    o = np.zeros((T, 3), np.float32); o[:, 0] = -1; o[:, 1] = np.linspace(0.01, 0.49, T); o[:, 2] = 0.25
    d = np.tile(np.array([1, 0, 0], np.float32), (T, 1)); d[::3] = [1, 0.002, 0]
    crays = orc.make_rays(o, d)
    coff, cidx, _, _ = orc.traverse_flat(orc.flatten(orc.build(chain).nodes), chain, crays, threads=orc.max_threads())
    crb = []
    for i, (first, cnt) in enumerate(shards):
        tdev = torch.from_numpy(crays[first:first + cnt].view(np.uint8).reshape(-1).copy()).to(dev_of(i))
        bufs.append(tdev)
        crb.append(RayBatch.from_device(tdev, cnt, np.float32))
    root2 = Bvh.from_aabbs(torch.from_numpy(chain[:100].copy()).to(dev_of(0)), ctxs[0])   # (no level hint for 12 000 shapes)
    chain_dev = torch.from_numpy(chain).to(dev_of(0))
    root2.rebuild_async(chain_dev)
    trees2 = comm.bcast([root2] + [None] * (K - 1), 0, "f32", len(chain))
    statuses = []
    for i in range(K):
        trees2[i].traverse_async(crb[i], hits[i])
    for i in range(K):
        try:
            hits[i].wait(); statuses.append(OK)
        except BvhGpuError as e:
            statuses.append(e.status)
    out["unbalanced_first_statuses"] = statuses
    trees2 = comm.bcast(trees2, 0, "f32", len(chain))                    # every rank repeats the call; the root's tree is final now
    for i in range(K):
        trees2[i].traverse_async(crb[i], hits[i])
    for i in range(K):
        hits[i].wait()
    off, idx = gather(trees2, hits)
    out["unbalanced_rebroadcast_ok"] = bool(all(s == REBROADCAST for s in statuses) and np.array_equal(off, coff) and np.array_equal(idx, cidx))
    # the root's own result after the REBROADCAST status was complete already (its wait replayed the batch on the finished tree)
    # and the steady state (level hint learned) needs no rebroadcast any more
    root2.rebuild_async(chain_dev)
    trees2 = comm.bcast(trees2, 0, "f32", len(chain))
    for i in range(K):
        trees2[i].traverse_async(crb[i], hits[i])
    for i in range(K):
        hits[i].wait()
    off, idx = gather(trees2, hits)
    out["unbalanced_steady_state_ok"] = bool(np.array_equal(off, coff) and np.array_equal(idx, cidx))

    # ---- 6. exact_only travels: a scene whose splits have no SAH winner must be walked binary on the peers too -----------
    rng = np.random.default_rng(5)
    big = np.float32(1e19)
    lo4 = (rng.uniform(-1, 1, size=(500, 3)) * big).astype(np.float32)
    far = np.concatenate([lo4, lo4 + big * np.float32(0.01)], axis=1)
    o4 = (rng.uniform(-1, 1, size=(T, 3)) * big).astype(np.float32)
    d4 = rng.normal(size=(T, 3)).astype(np.float32)
    frays = orc.make_rays(o4, d4)
    foff, fidx, _, _ = orc.traverse_flat(orc.flatten(orc.build(far).nodes), far, frays, threads=orc.max_threads())
    frb = []
    for i, (first, cnt) in enumerate(shards):
        tdev = torch.from_numpy(frays[first:first + cnt].view(np.uint8).reshape(-1).copy()).to(dev_of(i))
        bufs.append(tdev)
        frb.append(RayBatch.from_device(tdev, cnt, np.float32))
    far_dev = torch.from_numpy(far).to(dev_of(0))
    root3 = Bvh.from_aabbs(far_dev, ctxs[0])
    root3.flatten_in_place()
    res = {}
    for form in ("header", "known", "known_async"):
        if form == "known_async":
            root3.rebuild_async(far_dev)
        trees3 = comm.bcast([root3] + [None] * (K - 1), 0, *(() if form == "header" else ("f32", len(far))))
        for i in range(K):
            trees3[i].traverse_async(frb[i], hits[i])
        for i in range(K):
            hits[i].wait()
        off, idx = gather(trees3, hits)
        res[form] = bool(np.array_equal(off, foff) and np.array_equal(idx, fidx))
        for t in trees3[1:]:
            t.close()
    out["exact_only_travels"] = res
    out["exact_only_hits"] = int(len(fidx))

    # ---- 7. the triangle vertices travel when asked for: every rank's fused closest hit == the harness loop of the oracle ------
    tris, _ = tb.create_n_cubes(2500)
    root.rebuild(aabbs_dev, flatten=True)
    root.set_triangles(torch.from_numpy(tris.reshape(-1, 9)).to(dev_of(0)))
    _, oclosest, oprim = orc.triangle_stage(tris, orc.create_rays(0, T), ooff, oidx)
    res = {}
    for form in ("header", "known"):
        trees7 = comm.bcast([root] + [None] * (K - 1), 0, *(() if form == "header" else ("f32", n)), triangles=True)
        ok = True
        for i, (first, cnt) in enumerate(shards):
            cl, prim, _ = trees7[i].closest_hits(rays[i])
            ok = ok and cl.tobytes() == oclosest[first:first + cnt].tobytes() and bool(np.array_equal(prim, oprim[first:first + cnt]))
        res[form] = bool(ok)
        for t in trees7[1:]:
            t.close()
    out["triangles_travel"] = res
    for h in hits:
        h.close()
    comm.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
