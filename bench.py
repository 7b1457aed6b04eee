#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: Mrays/s for build + flatten + traverse.

One STEP = one pass of the hot path over one batch of synthetic input that is already resident in
HBM: Bvh::build_par (SAH) → Bvh::flatten → FlatBvh::traverse of R rays, results left in HBM as CSR.
Workload at every N: BASELINE.json configs[1] — create_n_cubes(10 000) = 120 000 triangles f32/3D and
R = 1 000 000 create_ray rays PER GPU (weak scaling: rank r traverses rays [r*R, (r+1)*R) of the
seed-0 stream).  N > 1, two plans for the scene (--scene-dist): "bcast" — rank 0 builds + flattens, the
traversal array + shape AABBs travel to the peers in ONE RCCL broadcast per step (torch.distributed, backend
nccl == RCCL over xGMI); "replicate" — every rank runs the deterministic build itself, no collective on the
data path.  The default "auto" times both plans for a few untimed steps and keeps the faster one (at
120 000 triangles a 0.26 ms build competes with moving 10.6 MB over xGMI); the probe times are reported.
Every rank traverses its own ray shard; hit lists stay on the GPU that produced them.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     — the dominant kernel (k_traverse) against the HBM roofline: ALGORITHMIC bytes per
                 launch (from exact visit counters) / HIP-event kernel time.
  cpu_baseline — the oracle (a C port of the reference algorithm, kind "port") timed on this box's
                 host cores on a bounded sample of the same workload; rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--cubes", type=int, default=10_000, help="create_n_cubes(n): 12 triangles each")
    ap.add_argument("--rays", type=int, default=1_000_000, help="rays per GPU per step")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--scene-dist", choices=["auto", "bcast", "replicate"], default="auto",
                    help="N>1: broadcast rank 0's flat scene over RCCL each step, or rebuild it on every rank (the build is "
                         "deterministic); auto times both during warmup and keeps the faster plan")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline-streams", type=int, default=2,
                    help="N=1 only, reported beside `value` (never as it): the same K steps issued from this many host threads on "
                         "this many HIP streams, so that the latency-bound build of one step overlaps the traversal of another; 0 = skip")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N>1 (nccl == RCCL; gloo only for the one-GPU test of this script)")
    ap.add_argument("--one-device", action="store_true",
                    help="test only: every rank uses cuda:0 (needs --backend gloo: RCCL refuses two ranks on one GPU)")
    ap.add_argument("--cpu-sample-rays", type=int, default=1_000_000)
    ap.add_argument("--pmc-traffic", type=float, default=None,
                    help="HBM bytes per launch of the traversal kernel from a separate rocprofv3 --pmc pass; default: "
                         "the newest profiles/*_traffic.json (written by tools/profile_round.sh on this command)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n_gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if args.one_device:
        if args.backend != "gloo":
            raise SystemExit("--one-device needs --backend gloo")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if n_gpus > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")

    import bvh_amd
    from bvh_amd import Bvh, Context, FlatBvh, RayBatch, dist as bdist, testbase as tb
    from bvh_amd._lib import RAY_F32, RAY_F64

    dtype = np.float32 if args.dtype == "f32" else np.float64
    tdtype = torch.float32 if args.dtype == "f32" else torch.float64
    ray_size = (RAY_F32 if args.dtype == "f32" else RAY_F64).itemsize
    elem = 4 if args.dtype == "f32" else 8

    # the engine enqueues on torch's current stream: torch events / synchronize see all of it
    stream = torch.cuda.current_stream(dev)
    ctx = Context(local_rank, stream=stream.cuda_stream)

    # ---- synthetic inputs, resident in HBM before the timed region ----
    bounds = tb.default_bounds()
    _, aabbs_np = tb.create_n_cubes(args.cubes, bounds)
    n_tri = len(aabbs_np)
    aabbs = torch.from_numpy(aabbs_np.astype(dtype)).to(dev)
    R = args.rays
    rays_buf = torch.empty(R * ray_size, dtype=torch.uint8, device=dev)
    first, _ = bdist.shard_range(rank, n_gpus, R)
    rays = RayBatch.generate(first, R, bounds, rays_buf, dtype, ctx)
    torch.cuda.synchronize(dev)

    # N>1, two plans for getting the scene to every GPU each step:
    #   bcast      rank 0 builds + flattens, ONE RCCL broadcast of the scene blob, peers import it
    #   replicate  every rank runs the (deterministic) build itself: no collective on the data path
    # auto probes both before the warmup and keeps the faster one; the probe times go into the JSON line.
    plans = ["single"] if n_gpus == 1 else (["bcast", "replicate"] if args.scene_dist == "auto" else [args.scene_dist])
    own_tree = rank == 0 or "replicate" in plans or n_gpus == 1
    bvh = Bvh.from_aabbs(aabbs, ctx) if own_tree else None
    if own_tree:
        bvh.flatten_in_place()
    blob = None
    peer = None
    if "bcast" in plans:
        nbytes = bdist.broadcast_nbytes(bvh.scene_nbytes() if rank == 0 else 0, dev, 0)
        blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    plan = plans[0]

    def step():
        nonlocal peer
        if plan == "bcast":
            if rank == 0:
                bvh.rebuild(aabbs, flatten=True)   # Bvh::build_par + Bvh::flatten (FlatBvh::build, flat_bvh.rs:328-331)
                bvh.scene_export(blob)
            bdist.broadcast_scene(blob, 0)         # RCCL over xGMI: traversal array + shape AABBs
            if rank != 0:
                peer = FlatBvh.scene_import(blob, blob.numel(), ctx, reuse=peer)
            tree = bvh if rank == 0 else peer
        else:
            bvh.rebuild(aabbs, flatten=True)
            tree = bvh
        return tree.traverse_batch(rays, fetch=False)[3]  # FlatBvh::traverse, CSR stays in HBM

    def barrier():
        if n_gpus > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(k):
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        barrier()
        dt = time.perf_counter() - t0
        if n_gpus > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    probe_ms = {}
    if len(plans) > 1:          # auto: a short probe of each plan (untimed for the metric), all ranks agree on the max-over-ranks time
        for pl in plans:
            plan = pl
            step(); step()
            probe_ms[pl] = timed(5) / 5 * 1e3
        plan = min(plans, key=lambda q: probe_ms[q])
    for _ in range(args.warmup):
        step()
    elapsed = timed(args.steps)
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)
    value = (n_gpus * R) / (ms_per_step * 1e-3) / 1e6  # Mrays/s, whole job

    # ---- per-phase HIP-event times + roofline of the dominant kernel (untimed extra steps) ----
    ctx.enable_timing(True)
    builder = plan != "bcast" or rank == 0
    tree = bvh if builder else peer
    ph = dict(build_ms=[], flatten_ms=[], traverse_kernel_ms=[], traverse_total_ms=[])
    for _ in range(max(5, min(args.steps, 20))):
        if builder:
            bvh.rebuild(aabbs)
            bvh.flatten_in_place()
        tree.traverse_batch(rays, fetch=False)
        t = ctx.last_timings()
        for k in ph:
            ph[k].append(t[k])
    ctx.enable_timing(False)
    phases = {k: float(np.mean(v)) for k, v in ph.items()}
    # exact visit counters (reference-equivalent loop iterations) for the algorithmic byte count
    stats = tree.traverse_batch(rays, stats=True, fetch=False)[3]
    V, VL, H = stats["visited"], stats["leaf_visits"], stats["hits"]
    hits_all = H
    if n_gpus > 1:   # whole-job hit count (untimed): lets a reader check the shards against one process over N*R rays
        ht = torch.tensor([H], dtype=torch.int64, device=dev)
        dist.all_reduce(ht, op=dist.ReduceOp.SUM)
        hits_all = int(ht.item())
    flat_sz = 36 if args.dtype == "f32" else 64
    # per ray  Ray in + V_nav*FlatNode + V_leaf*shape AABB + CSR out 4*(H+1)
    algo_bytes = R * ray_size + (V - VL) * flat_sz + VL * flat_sz + VL * 6 * elem + 4 * (H + R)
    kern_s = phases["traverse_kernel_ms"] * 1e-3
    achieved = algo_bytes / kern_s / 1e9
    traffic, traffic_src = args.pmc_traffic, "--pmc-traffic"
    if traffic is None and args.cubes == 10_000 and R == 1_000_000 and args.dtype == "f32":
        import glob
        found = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), key=os.path.getmtime)
        if found:
            tj = json.load(open(found[-1]))
            traffic, traffic_src = tj.get("hbm_bytes_per_launch"), os.path.relpath(found[-1], ROOT)
    roofline = {
        "kernel": "k_traverse_lds", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "traffic_source": None if traffic is None else f"{traffic_src}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                          "of this command, bytes per launch, FETCH_SIZE doubled per MI355X_MICROARCH.md",
        "algorithmic_bytes_per_launch": int(algo_bytes), "kernel_ms": round(phases["traverse_kernel_ms"], 4),
        "slab_tests_per_s": round(V / kern_s, 1), "visited": int(V), "leaf_visits": int(VL), "hits": int(H),
        "device_steps": int(stats["device_steps"]),
    }

    out = {
        "metric": "Mrays/s (build+traverse)", "value": round(value, 3), "unit": "Mrays/s", "n_gpus": n_gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {
            "workload": f"configs[1]: create_n_cubes({args.cubes}) = {n_tri} random-cube triangles {args.dtype}/3D, "
                        f"{R} create_ray rays per GPU; step = Bvh::build_par + flatten + FlatBvh::traverse (CSR hit lists in HBM)",
            "triangles": n_tri, "rays_per_gpu": R, "scene_dist": plan,
            "parallelism": f"rays sharded x{n_gpus}" + {"single": "", "bcast": ", flat scene RCCL-broadcast from rank 0 every step",
                                                        "replicate": ", every rank rebuilds the scene (deterministic build, no data-path collective)"}[plan],
        },
        "phases_ms": {k: round(v, 4) for k, v in phases.items()},
        "build_levels": bvh.build_levels if builder else None,
        "hits_all_ranks": int(hits_all),
        "scene_dist_probe_ms_per_step": {k: round(v, 4) for k, v in probe_ms.items()} or None,
        "roofline": roofline,
    }

    # ---- supplementary: independent steps in flight on several HIP streams (N = 1) ----
    # `value` above is the time of K steps issued one after the other on ONE stream.  Steps are independent of each other
    # (each rebuilds the scene from the shape AABBs), and the builder's ~20 small dependent kernels leave most CUs idle, so
    # a renderer would keep the next frame's build in flight while the current frame traces.  Same K steps, same work:
    # S host threads, each with its own context / stream / tree / hit buffers, K/S steps each.
    if n_gpus == 1 and args.pipeline_streams > 1:
        import threading
        S = args.pipeline_streams
        lanes = []
        for j in range(S):
            c = Context(local_rank)                       # its own non-blocking HIP stream
            tr = Bvh.from_aabbs(aabbs, c)
            tr.flatten_in_place()
            lanes.append((c, tr))

        def lane_steps(tr, k):
            for _ in range(k):
                tr.rebuild(aabbs, flatten=True)
                tr.traverse_batch(rays, fetch=False)

        per = max(args.steps // S, 1)
        for c, tr in lanes:
            lane_steps(tr, 3)
        torch.cuda.synchronize(dev)
        threads = [threading.Thread(target=lane_steps, args=(tr, per)) for _, tr in lanes]
        t0 = time.perf_counter()
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        torch.cuda.synchronize(dev)
        dtp = time.perf_counter() - t0
        hits_p = [tr.traverse_batch(rays, stats=True, fetch=False)[3]["hits"] for _, tr in lanes]
        out["pipelined"] = {
            "streams": S, "steps": per * S, "value": round(per * S * R / dtp / 1e6, 3), "unit": "Mrays/s",
            "ms_per_step": round(dtp * 1e3 / (per * S), 4), "hits_per_lane": hits_p,
            "note": f"the same steps issued from {S} host threads on {S} HIP streams (own context, tree and hit buffers each): "
                    "the build of one step overlaps the traversal of another; throughput of independent steps, not the latency of one",
        }
        for c, tr in lanes:
            tr.close()
            c.close()

    # ---- CPU baseline: the oracle (C port of the reference algorithm) on this box's host cores ----
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        from oracle import orc
        cores = orc.max_threads()
        a = aabbs_np.astype(dtype)
        orc.build(a)                                    # warm the allocator and the page cache
        tb_par, par_threads = 1e9, 0
        for th in sorted({4, 8, 16, 32, cores} & set(range(1, cores + 1))):   # libgomp's task queue stops scaling early
            t0 = time.perf_counter(); ot = orc.build(a, threads=th); dt = time.perf_counter() - t0
            if dt < tb_par:
                tb_par, par_threads = dt, th
        tb_ser = 1e9
        for _ in range(2):
            t0 = time.perf_counter(); ot = orc.build(a, parallel=False); tb_ser = min(tb_ser, time.perf_counter() - t0)
        t0 = time.perf_counter(); of = orc.flatten(ot.nodes); tf = time.perf_counter() - t0
        ns = min(args.cpu_sample_rays, R)
        rr = orc.create_rays(0, ns)
        if args.dtype == "f64":
            rr = orc.make_rays(rr["o"], rr["d"], np.float64)
        st_o = orc.TravStats()
        import ctypes as C
        offs = np.zeros(ns + 1, dtype=np.uint32)
        fn = getattr(orc.lib(), f"orc_traverse_flat_{args.dtype}")
        sa = np.ascontiguousarray(a)

        def trav(threads):
            t0 = time.perf_counter()
            fn(orc._p(of), C.c_size_t(len(of)), orc._p(sa), orc._p(rr), C.c_size_t(ns), orc._p(offs), None,
               C.c_uint64(0), None, C.byref(st_o), C.c_int(threads))
            return time.perf_counter() - t0
        trav(cores)                                     # first call creates the OpenMP team
        tt_all, trav_threads = 1e9, cores               # the box may grant fewer CPUs than it shows: keep the best team size
        for th in sorted({8, 16, 32, 64, cores} & set(range(1, cores + 1))):
            dt = min(trav(th), trav(th))
            if dt < tt_all:
                tt_all, trav_threads = dt, th
        n1 = max(ns // 16, 1000)
        t0 = time.perf_counter()
        fn(orc._p(of), C.c_size_t(len(of)), orc._p(sa), orc._p(rr), C.c_size_t(n1), orc._p(offs), None,
           C.c_uint64(0), None, C.byref(st_o), C.c_int(1))
        tt_1 = time.perf_counter() - t0
        tbuild = min(tb_par, tb_ser)
        cpu_total = tbuild + tf + tt_all * (R / ns)
        out["cpu_baseline"] = {
            "value": round(R / cpu_total / 1e6, 4), "unit": "Mrays/s", "cores": max(trav_threads, par_threads), "host_cpus_visible": cores, "kind": "port",
            "sample": f"oracle (C restatement, gcc -O2 -ffp-contract=off, OpenMP): full {n_tri}-triangle build "
                      f"(best of task-parallel {tb_par * 1e3:.1f} ms on {par_threads} threads / serial {tb_ser * 1e3:.1f} ms) + serial flatten "
                      f"{tf * 1e3:.1f} ms + traversal of {ns} of the {R} rays on {trav_threads} threads (best of 8/16/32/64/{cores}: {tt_all * 1e3:.1f} ms, count pass only), "
                      f"scaled to {R} rays; single-thread traversal {tt_1 / n1 * 1e9:.0f} ns/ray "
                      f"(README.md:175 quotes 866 ns/ray on a Ryzen 9 3900X for the Rust crate)",
            "build_ms": round(tbuild * 1e3, 2), "flatten_ms": round(tf * 1e3, 2),
            "traverse_ms_all_cores": round(tt_all * (R / ns) * 1e3, 2),
            "traverse_ns_per_ray_1thread": round(tt_1 / n1 * 1e9, 1),
        }
        out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 2)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if n_gpus > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
