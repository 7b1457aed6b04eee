"""Driver of tests/test_gpu_multictx.py::test_rank_per_thread_protocol (a subprocess on the GPU box): the PROCESS-PER-GPU form of the
multi-GPU step — bvhgpu_comm_unique_id / bvhgpu_comm_init_rank, one communicator handle per rank, the root REMOTE for every peer —
with K host threads standing in for the K processes of `torchrun` (each with its own ctx = its own stream, all on device 0) and
tests/c_abi/libfakerccl.so standing in for RCCL (its ranks-are-threads worlds meet at a barrier in ncclGroupEnd).  The step every
thread runs is bench.py's N > 1 step: rank 0 rebuild_async → bcast_known → traverse_async → wait; a peer bcast_known →
traverse_async → wait; no host synchronisation before the wait.

    python tests/multi_thread_driver.py <K>          prints one JSON object; every check is against the oracle"""
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    os.environ["BVHGPU_RCCL_LIB"] = os.path.join(ROOT, "tests", "c_abi", "libfakerccl.so")
    os.environ["BVHGPU_RCCL_SHARED_DEVICE"] = "1"
    import torch
    from bvh_amd import Bvh, Context, RayBatch, dist as bdist, testbase as tb
    from bvh_amd._lib import INVALID_ARG, OK, RAY_F32, REBROADCAST, TRAVERSE_RAYS_READY, BvhGpuError
    from bvh_amd.api import _Hits
    from oracle import orc

    bounds = tb.default_bounds()
    _, aabbs = tb.create_n_cubes(2500)
    n = len(aabbs)
    T = 120_000
    shards = [bdist.strong_shard(r, K, T) for r in range(K)]
    oflat = orc.flatten(orc.build(aabbs).nodes)
    ooff, oidx, _, _ = orc.traverse_flat(oflat, aabbs, orc.create_rays(0, T), threads=orc.max_threads())
    x = np.float32(1.004) ** np.arange(12000, dtype=np.float32)
    lo = np.stack([x, np.zeros_like(x), np.zeros_like(x)], axis=1)
    chain = np.concatenate([lo, lo + np.float32(0.5)], axis=1).astype(np.float32)
    o = np.zeros((T, 3), np.float32); o[:, 0] = -1; o[:, 1] = np.linspace(0.01, 0.49, T); o[:, 2] = 0.25
    d = np.tile(np.array([1, 0, 0], np.float32), (T, 1)); d[::3] = [1, 0.002, 0]
    crays = orc.make_rays(o, d)
    coff, cidx, _, _ = orc.traverse_flat(orc.flatten(orc.build(chain).nodes), chain, crays, threads=orc.max_threads())

    uid = bdist.Communicator.unique_id()
    res = [dict() for _ in range(K)]
    errors = []
    sync = threading.Barrier(K)

    def csr_slice(off, idx, first, cnt):
        base = int(off[first])
        return off[first:first + cnt + 1] - np.uint32(base), idx[base:int(off[first + cnt])]

    def rank_main(rank):
        try:
            out = res[rank]
            ctx = Context(0)
            comm = bdist.Communicator(ctx, K, rank, uid)
            first, cnt = shards[rank]
            buf = torch.empty(max(cnt, 1) * RAY_F32.itemsize, dtype=torch.uint8, device="cuda")
            rays = RayBatch.generate(first, cnt, bounds, buf, np.float32, ctx)
            hits = _Hits(ctx)
            aabbs_dev = torch.from_numpy(aabbs).cuda() if rank == 0 else None
            tree = Bvh.from_aabbs(aabbs_dev, ctx) if rank == 0 else None
            if rank == 0:
                tree.flatten_in_place()
            want = csr_slice(ooff, oidx, first, cnt)

            def step(t, rays_, n_shapes, src=None):
                """bench.py's N > 1 step (bvh_amd.dist.broadcast_step, the very function bench.py calls); returns (tree, statuses seen)"""
                try:
                    t2, _, reb = bdist.broadcast_step(comm, rank, t, src, rays_, "f32", n_shapes, hits, TRAVERSE_RAYS_READY)
                    return t2, (REBROADCAST if reb else OK)
                except BvhGpuError as e:
                    return t, e.status

            # 1. the asynchronous step, three times; then the header form (peers know nothing)
            ok = True
            for _ in range(3):
                tree, st = step(tree, rays, n, aabbs_dev)
                off, idx = hits.fetch(cnt)
                ok = ok and st == OK and np.array_equal(off, want[0]) and np.array_equal(idx, want[1])
            out["async_step_equal"] = bool(ok)
            sync.wait()
            tree = comm.bcast(tree, 0)
            off, idx, _, _ = tree.traverse_batch(rays)
            out["header_form_equal"] = bool(np.array_equal(off, want[0]) and np.array_equal(idx, want[1]))
            sync.wait()

            # 2. the root announces a size its tree does not have: ITS call returns the reason, the peers' waits return INVALID_ARG
            if rank == 0:
                try:
                    comm.bcast(tree, 0, "f32", n + 1); out["wrong_announcement"] = OK
                except BvhGpuError as e:
                    out["wrong_announcement"] = e.status
            else:
                tree = comm.bcast(tree, 0, "f32", n + 1)
                try:
                    tree.traverse_async(rays, hits).wait(); out["wrong_announcement"] = OK
                except BvhGpuError as e:
                    out["wrong_announcement"] = e.status
            sync.wait()

            # 3. an unbalanced tree on a first asynchronous build: REBROADCAST on every rank, the repeat succeeds, the steady state too
            tdev = torch.from_numpy(crays[first:first + cnt].view(np.uint8).reshape(-1).copy()).cuda()
            crb = RayBatch.from_device(tdev, cnt, np.float32)
            cwant = csr_slice(coff, cidx, first, cnt)
            chain_dev = torch.from_numpy(chain).cuda() if rank == 0 else None
            t2 = Bvh.from_aabbs(torch.from_numpy(chain[:100].copy()).cuda(), ctx) if rank == 0 else None
            t2, st1 = step(t2, crb, len(chain), chain_dev)              # (the rebroadcast happens inside the step, on every rank alike)
            off, idx = hits.fetch(cnt)
            out["unbalanced"] = [st1, 0, bool(np.array_equal(off, cwant[0]) and np.array_equal(idx, cwant[1]))]
            t2, st3 = step(t2, crb, len(chain), chain_dev)              # level hint learned: no rebroadcast any more
            off, idx = hits.fetch(cnt)
            out["unbalanced_steady"] = [st3, bool(np.array_equal(off, cwant[0]) and np.array_equal(idx, cwant[1]))]
            sync.wait()
            hits.close()
            comm.close()
        except Exception as e:   # a failing rank must not leave the others at a barrier for ever
            errors.append(f"rank {rank}: {e!r}")
            try:
                sync.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(K)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    alive = [t.is_alive() for t in threads]
    print(json.dumps({"K": K, "ranks": res, "errors": errors, "hung": alive, "expect": {"INVALID_ARG": INVALID_ARG, "REBROADCAST": REBROADCAST}}))
    if any(alive):
        os._exit(3)


if __name__ == "__main__":
    main()
