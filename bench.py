#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: Mrays/s for build + flatten + traverse.

One STEP = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
Bvh::build_par (SAH) → Bvh::flatten (the FlatNode array in the reference's layout included) → FlatBvh::traverse of R rays, CSR left in HBM.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts the N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

This file: argument parsing and launch, the timed step (run_workload), the compact JSON line (compact_line).  Everything reported BESIDE
the timed step — per-phase times, rooflines, the parity legs against the oracle, step_excludes, the extra configs, the CPU baseline, the
N > 1 plans — lives in bench_sections.py.  What each field of the line means is written in DESIGN.md §7, not on the line.

stdout carries exactly ONE line: the compact JSON (numbers only, < 8 KB; tests/test_bench_line_cpu.py bounds it).  The detailed object
(every section in full) goes to the side file `--detail-out` (default gpurun_out/bench_detail.json, or ./bench_detail.json).

Headline workload (`value`): BASELINE.json configs[1] — create_n_cubes(10 000) = 120 000 triangles f32/3D and R = 1 000 000 create_ray
rays PER GPU (weak scaling: rank r traverses rays [r*R, (r+1)*R) of the seed-0 stream).  `extra_configs` carries the other BASELINE
configs, the reference's whole harness loop and a 12 M-triangle scene as one compact row each.  A run is never downgraded: `--gpus N`
with no WORLD_SIZE re-executes itself under torch.distributed.run with N ranks; with a WORLD_SIZE that is not N it stops with an error.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench_sections import Watchdog, timed_out_line  # noqa: E402,F401  (tests reach them through this module)

LINE_BUDGET = 8192      # bytes of the stdout line (BENCH_r05: a 35 KB line could not be parsed by the driver)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    A = ap.add_argument
    A("--gpus", type=int, default=1)
    A("--steps", type=int, default=200)
    A("--warmup", type=int, default=10)
    A("--workload", choices=["cubes120k", "cubes12m", "standin-primary", "standin-incoherent"], default="cubes120k")
    A("--cubes", type=int, default=10_000, help="cubes120k: create_n_cubes(n), 12 triangles each")
    A("--rays", type=int, default=None, help="rays per GPU per step (weak) / in total (strong); default per workload")
    A("--dtype", choices=["f32", "f64"], default="f32")
    A("--harness", choices=["closest", "triangles"], default=None,
      help="the headline step becomes the reference's WHOLE bench iteration (intersect_bh, testbase.rs:819-837), N = 1")
    A("--scaling", choices=["weak", "strong"], default=None, help="weak: --rays per GPU; strong: --rays in total, sharded over the GPUs")
    A("--scene-dist", choices=["auto", "bcast", "replicate", "bcast-torch"], default="auto",
      help="N>1: RCCL-broadcast rank 0's flattened tree through the C ABI each step, or rebuild it on every rank; auto measures both")
    A("--flat-array", choices=["eager", "all", "beside", "lazy"], default="eager",
      help="the reference-layout FlatNode array (what Bvh::flatten returns) is written in every step: eager = by the flatten kernel "
           "(the engine's own folded binary array on first use), all = both arrays, beside = both by a second pass beside the walk; "
           "lazy: on first use (NOT in the step)")
    A("--collective-timeout", type=float, default=60.0, help="N>1: seconds the exchange plan may take before the replicate line is printed")
    A("--regions", type=int, default=5, help="timed regions of exactly K steps each; `value` is the median region, all are on the line")
    A("--settle-steps", type=int, default=300, help="untimed steps before the W warmup steps (clock ramp; at most 20 for big batches)")
    A("--no-cpu-baseline", action="store_true")
    A("--no-extra", action="store_true", help="skip the extra_configs sub-runs")
    A("--extras-timeout", type=float, default=300.0, help="N>1: seconds after which the extra_configs section is given up")
    A("--no-parity", action="store_true")
    A("--no-excluded", action="store_true", help="skip the step_excludes section")
    A("--extra-steps", type=int, default=20)
    A("--parity-max-rays", type=int, default=200_000_000, help="rays of a batch diffed against the oracle (default: every ray)")
    A("--pipeline-streams", type=int, default=2, help="N=1, beside `value`: the same K steps in flight on this many streams; 0 = skip")
    A("--backend", choices=["nccl", "gloo"], default="nccl", help="torch.distributed backend for N>1 (nccl == RCCL; gloo: one-GPU tests)")
    A("--one-device", action="store_true", help="test only: every rank uses cuda:0 (needs --backend gloo)")
    A("--cpu-sample-rays", type=int, default=1_000_000)
    A("--standin-detail", type=int, default=16)
    A("--detail-out", default=None, help="file for the detailed JSON object (default gpurun_out/bench_detail.json or ./bench_detail.json)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------------------------
def resolve_launch(gpus, environ):
    """How this invocation becomes `gpus` ranks.  Returns ("run", world) when this process IS one rank of a launched job (or the single
    rank of an N = 1 run) and ("spawn", gpus) when it has to launch the ranks itself.  A run can never silently shrink: --gpus N with a
    WORLD_SIZE that is set and is not N is an error, and --gpus N > 1 without WORLD_SIZE launches N ranks."""
    if gpus < 1:
        raise SystemExit(f"--gpus {gpus}: need at least one GPU")
    ws = environ.get("WORLD_SIZE")
    if ws is None or ws == "":
        return ("run", 1) if gpus == 1 else ("spawn", gpus)
    try:
        world = int(ws)
    except ValueError:
        raise SystemExit(f"WORLD_SIZE={ws!r} is not a number")
    if world != gpus:
        raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={world}: the launcher and the flag must agree (a run is never downgraded)")
    for k in ("RANK", "LOCAL_RANK"):
        if world > 1 and environ.get(k) in (None, ""):
            raise SystemExit(f"WORLD_SIZE={world} but {k} is not set: launch with torch.distributed.run or `python bench.py --gpus {gpus}`")
    return ("run", world)


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(gpus, argv, port):
    """the command `python bench.py --gpus N` turns itself into: the driver's own launch line (one rank per GPU, rendezvous on 127.0.0.1)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(gpus, argv):
    """`python bench.py --gpus N` without a launcher: run N ranks under torch.distributed.run and pass rank 0's JSON line through."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["BVH_BENCH_SELF_LAUNCHED"] = "1"
    cmd = launch_command(gpus, argv, free_port())
    sys.stderr.write("bench.py: --gpus %d without a launcher: running %s\n" % (gpus, " ".join(cmd)))
    sys.stderr.flush()
    if os.environ.get("BVH_BENCH_LAUNCH_DRY_RUN"):      # tests: show what would run, run nothing
        print(json.dumps({"launch": cmd}))
        return 0
    return subprocess.call(cmd, env=env)


# ----------------------------------------------------------------------------------------------------------------------------------
def run_workload(wl, args, env, steps, warmup, detailed, force_plan=None, regions=None):
    """THE TIMED STEP.  K steps of one workload on this rank's GPU (all ranks call it together) → result dict.  force_plan: measure
    exactly this scene-distribution plan (Run.measure runs "replicate" first and the exchange plan afterwards, under a watchdog)."""
    import torch
    import torch.distributed as dist
    import bench_sections as sec
    from bvh_amd import Bvh, FlatBvh, dist as bdist
    from bvh_amd._lib import TRAVERSE_CLOSEST, TRAVERSE_COHERENT, TRAVERSE_RAYS_READY, TRAVERSE_TRIANGLES, TUNE_FLATTEN_LAZY
    rank, n_gpus, dev, ctx, comm = env["rank"], env["n_gpus"], env["dev"], env["ctx"], env["comm"]
    regions = args.regions if regions is None else regions
    aabbs, rays = wl.aabbs, wl.rays
    if wl.harness and n_gpus != 1:
        raise SystemExit("--harness is an N = 1 measurement (the triangle vertices are not part of the broadcast plan's step)")
    coherent = TRAVERSE_COHERENT if wl.coherent else 0
    mode_flags = coherent | {None: 0, "closest": TRAVERSE_CLOSEST, "triangles": TRAVERSE_TRIANGLES}[wl.harness]
    flags = coherent | TRAVERSE_RAYS_READY    # the ray batch is resident in HBM since before the timed region

    if n_gpus == 1:
        plans = ["single"]
    elif force_plan is not None:
        plans = [force_plan]
    elif args.scene_dist == "auto":
        plans = (["bcast"] if comm is not None else ["bcast-torch"]) + ["replicate"]
    else:
        plans = ["bcast-torch" if (args.scene_dist == "bcast" and comm is None) else args.scene_dist]
    own_tree = rank == 0 or "replicate" in plans
    bvh = Bvh.from_aabbs(aabbs, ctx) if own_tree else None
    if own_tree:
        bvh.flatten_in_place()
        if wl.harness:
            bvh.set_triangles(wl.tris)     # vertices of the shapes, resident in HBM like the AABBs (a rebuild of as many shapes keeps them)
    blob = None
    if "bcast-torch" in plans:
        blob = torch.empty(bdist.broadcast_nbytes(bvh.scene_nbytes() if rank == 0 else 0, dev, 0), dtype=torch.uint8, device=dev)
    state = {"plan": plans[0], "peer": None}

    def step():
        plan = state["plan"]
        if plan in ("single", "replicate"):
            # FlatBvh::build (flat_bvh.rs:328-331) and FlatBvh::traverse enqueued back to back, ONE host round trip per step: the wait
            # validates the build, completes the batch and is the end of the step (nothing of the next step is in flight)
            if wl.harness:      # intersect_bh: the rays are made inside the step (k_gen_rays / k_gen_primary), then walked + intersected
                wl.regen()
            bvh.rebuild_async(aabbs)
            return bvh.traverse_async(rays, flags=mode_flags if wl.harness else flags).wait()
        if plan == "bcast":
            # the same asynchronous triple with the exchange step in the middle and NO host synchronisation before the final wait on any
            # rank: rank 0 enqueues build + flatten, the RCCL broadcast out of the tree's own buffers (bvhgpu_bcast_known) and its own
            # walk; a peer enqueues the receive and its walk.  BVHGPU_REBROADCAST (unbalanced tree on a first build) repeats the exchange.
            tree, st, reb = bdist.broadcast_step(comm, rank, bvh if rank == 0 else state["peer"], aabbs if rank == 0 else None, rays,
                                                 wl.dtype_name, wl.n_tri, flags=flags)
            if rank != 0:
                state["peer"] = tree
            state["rebroadcasts"] = state.get("rebroadcasts", 0) + reb
            return st
        # "bcast-torch", the fallback transport: scene blob over torch.distributed (host round trips)
        if rank == 0:
            bvh.rebuild(aabbs, flatten=True)
            bvh.scene_export(blob)
        bdist.broadcast_scene(blob, 0)
        if rank != 0:
            state["peer"] = FlatBvh.scene_import(blob, blob.numel(), ctx, reuse=state["peer"])
        return (bvh if rank == 0 else state["peer"]).traverse_batch(rays, fetch=False, coherent=wl.coherent)[3]

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
        if n_gpus > 1:      # the job's time is the slowest rank's
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    probe_ms = {}
    if len(plans) > 1:          # explicit auto probe: a short run of each plan (untimed for the metric), max-over-ranks time
        for pl in plans:
            state["plan"] = pl
            step(); step()
            probe_ms[pl] = timed(5) / 5 * 1e3
        state["plan"] = min(plans, key=lambda q: probe_ms[q])
    # Settle the GPU's clocks before the W warmup steps (a 0.33 ms step timed over K = 20 right after start-up reads 2 % low).  Untimed,
    # the same step, a fixed count on every rank; plans with a collective per step and big batches settle in 20 steps.
    settle = args.settle_steps if (wl.R <= 2_000_000 and state["plan"] in ("single", "replicate")) else min(args.settle_steps, 20)
    for _ in range(settle + warmup):
        step()
    # EXACTLY K steps between barrier + synchronize, max over ranks — `--regions` times back to back, the MEDIAN region is the figure and
    # every region is on the line.  (The GPU boxes' host CPUs are shared: the one host thread of a step loop is descheduled for tens of
    # ms now and then — r6_a: one such stall inside a 7 ms region read 470 instead of 2 900 Mrays/s.  A median over regions of K steps
    # says what K steps take; a single region says what the neighbours did.)
    region_s = [timed(steps) for _ in range(max(1, regions))]
    elapsed = sorted(region_s)[len(region_s) // 2]
    ms_per_step = elapsed * 1e3 / max(steps, 1)
    value = wl.total_rays / (ms_per_step * 1e-3) / 1e6   # Mrays/s, whole job (weak: N*R, strong: T)
    # ---- end of the timed region: everything below is reported beside it ----

    plan = state["plan"]
    builder = plan not in ("bcast", "bcast-torch") or rank == 0
    tree = bvh if builder else state["peer"]
    extra, phases, stats = sec.phases_and_roofline(wl, env, bvh, tree, builder, mode_flags, steps, detailed)
    hits_all = stats["hits"]
    if n_gpus > 1:   # whole-job hit count (untimed): lets a reader check the shards against one process over all rays
        ht = torch.tensor([hits_all], dtype=torch.int64, device=dev)
        dist.all_reduce(ht, op=dist.ReduceOp.SUM)
        hits_all = int(ht.item())
    out = {
        "workload": wl.tag, "harness": wl.harness, "config": wl.config_id, "dtype": wl.dtype_name, "value": round(value, 3),
        "unit": "Mrays/s", "ms_per_step": round(ms_per_step, 4), "steps": steps, "warmup": warmup, "settle_steps": settle,
        "regions_ms_per_step": [round(t * 1e3 / max(steps, 1), 4) for t in region_s],
        "scaling": wl.scaling, "triangles": wl.n_tri, "rays_this_rank": wl.R, "rays_total": wl.total_rays, "scene_dist": plan,
        "flat_array": {0: "all", 1: "lazy", 2: "beside", 3: "eager"}[ctx.get_tuning(TUNE_FLATTEN_LAZY)],
        "hits_all_ranks": int(hits_all), "visited_per_ray": round(stats["visited"] / max(wl.R, 1), 2),
        "scene_dist_probe_ms_per_step": {k: round(v, 4) for k, v in probe_ms.items()} or None,
    }
    out.update(extra)
    env["last"] = dict(bvh=bvh, tree=tree, stats=stats, phases=phases, builder=builder)
    if state.get("rebroadcasts"):
        out["rebroadcasts"] = state["rebroadcasts"]
    return out


# ----------------------------------------------------------------------------------------------------------------------------------
def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d and d[k] is not None}


ROOF_KEYS = ("kernel", "kernel_ms", "bound", "achieved", "peak", "unit", "frac", "traffic", "hbm_frac", "valu_frac", "lds_frac",
             "wait_frac", "algorithmic_frac", "algorithmic_bytes_per_launch", "slab_tests_per_s", "source")
PLAN_KEYS = ("value", "ms_per_step", "hits_all_ranks", "hits_match_n1_reference", "timed_out", "after_s", "stage", "workload")


def _plans(p):
    return {k: _pick(v, PLAN_KEYS) for k, v in p.items()} if p else None


def _extra_row(e):
    """one extra config as one compact row"""
    if "error" in e:
        return _pick(e, ("workload", "dtype", "harness", "error"))
    roof, ph = e.get("roofline") or {}, e.get("phases_ms") or {}
    row = _pick(e, ("workload", "harness", "config", "dtype", "scaling", "value", "ms_per_step", "triangles", "rays_total",
                    "rays_this_rank", "first_ray", "scene_dist", "hits_all_ranks", "hits_match_n1_reference", "speedup_vs_cpu_harness"))
    row.update(parity=_pick(e.get("parity"), ("equal", "checked_rays")),
               build_flatten_ms=round(ph.get("build_ms", 0) + ph.get("flatten_ms", 0), 4), walk_ms=ph.get("traverse_kernel_ms"),
               traverse_ms=ph.get("traverse_total_ms"))
    if "ray_gen_ms" in ph:
        row["ray_gen_ms"] = round(ph["ray_gen_ms"], 4)
    row.update(_pick(roof, ("bound", "frac", "hbm_frac", "wait_frac", "algorithmic_frac")))
    if e.get("cpu_harness"):
        row["cpu_harness"] = e["cpu_harness"]["value"]
    if e.get("scene_dist_plans"):
        row["scene_dist_plans"] = _plans(e["scene_dist_plans"])
    if e.get("pure_f64_walk"):
        f = e["pure_f64_walk"]
        row["pure_f64_walk"] = dict(_pick(f, ("value", "ms_per_step")), parity_equal=(f.get("parity") or {}).get("equal"),
                                    walk_ms=(f.get("phases_ms") or {}).get("traverse_kernel_ms"),
                                    **_pick(f.get("roofline"), ("bound", "frac")))
    return row


def compact_line(d):
    """The ONE stdout line: the task contract's keys + numbers only (prose lives in DESIGN.md §7, every section in full in the detail
    file).  `d` is the detailed object bench_sections builds."""
    out = _pick(d, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "settle_steps", "ms_per_step", "regions_ms_per_step",
                    "higher_is_better", "scaling"))
    out["vs_baseline"] = d.get("vs_baseline")
    out.update(_pick(d, ("dtype", "data", "workload_name", "harness")))
    out["config"] = d.get("config")
    out.update(_pick(d, ("phases_ms", "build_levels", "hits_all_ranks", "scene_dist_probe_ms_per_step")))
    if d.get("scene_dist_plans"):
        out["scene_dist_plans"] = _plans(d["scene_dist_plans"])
    out["roofline"] = _pick(d.get("roofline"), ROOF_KEYS)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):       # the contract's six keys are always there (null = unmeasured)
        out["roofline"].setdefault(k, None)
    if d.get("roofline_build"):
        out["roofline_build"] = _pick(d["roofline_build"], ("frac", "ms", "algorithmic_bytes", "achieved", "unit"))
    if d.get("parity"):
        out["parity"] = _pick(d["parity"], ("equal", "checked_rays", "bvh_nodes_equal"))
    if d.get("cpu_harness"):
        out["cpu_harness"] = _pick(d["cpu_harness"], ("value", "unit", "cores", "kind", "sample_rays"))
    if d.get("pipelined"):
        out["pipelined"] = _pick(d["pipelined"], ("value", "ms_per_step", "streams", "hits_every_step_equal"))
    x = d.get("step_excludes")
    if x:
        if "error" in x:
            out["step_excludes"] = {"error": x["error"][:200]}
        else:
            keys = ("with_ray_gen", "lazy_flat_array", "eager_flat_array", "all_arrays_eager", "beside_flat_array")
            out["step_excludes"] = {k: x[k]["value"] for k in keys if k in x}
            h = x.get("host_io")
            if h:
                out["step_excludes"]["host_io"] = h["value"]
                out["step_excludes"]["host_io_detail"] = {k: _pick(v, ("value", "ms_per_step", "pcie_gbs", "csr_equal_to_pageable_path"))
                                                          for k, v in (h.get("paths") or {}).items()}
    if d.get("extra_configs") is not None:
        out["extra_configs"] = [_extra_row(e) for e in d["extra_configs"]]
    if d.get("cpu_baseline"):
        out["cpu_baseline"] = _pick(d["cpu_baseline"], ("value", "value_median", "unit", "cores", "kind", "sample", "build_ms",
                                                         "flatten_ms", "traverse_ms_all_cores", "traverse_ns_per_ray_1thread",
                                                         "host_load_1m", "threads", "error"))
        legs = d["cpu_baseline"].get("legs")
        if legs:
            out["cpu_baseline"]["legs"] = {k: v.get("value") for k, v in legs.items()}
        if d.get("speedup_vs_cpu_baseline") is not None:
            out["speedup_vs_cpu_baseline"] = d["speedup_vs_cpu_baseline"]
    out["launch"] = d.get("launch")
    out["rccl"] = d.get("rccl")
    out.update(_pick(d, ("rccl_comm_error", "collective_watchdog", "detail")))
    return out


def render_line(detail):
    """the compact line as text, never above LINE_BUDGET: a line the driver cannot read is worth nothing (BENCH_r05), so if the rows ever
    outgrow the budget they are dropped (and counted) and the contract's keys stay"""
    text = json.dumps(compact_line(detail), separators=(",", ":"))
    if len(text) > LINE_BUDGET:
        slim = compact_line(dict(detail, extra_configs=None))
        slim["extra_configs_dropped"] = len(detail.get("extra_configs") or [])
        text = json.dumps(slim, separators=(",", ":"))
    return text


def detail_path(args):
    if args.detail_out:
        return args.detail_out
    d = os.path.join(ROOT, "gpurun_out")
    return os.path.join(d, "bench_detail.json") if os.path.isdir(d) and os.access(d, os.W_OK) else os.path.join(ROOT, "bench_detail.json")


# ----------------------------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    how, world = resolve_launch(args.gpus, os.environ)
    if how == "spawn":
        raise SystemExit(self_launch(world, sys.argv[1:]))
    # Exactly ONE line goes to stdout: the JSON.  RCCL prints a version banner to stdout when a communicator is created (torch's
    # and the engine's), so everything else that writes to fd 1 during the run is sent to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    import bench_sections as sec

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = world
    assert n_gpus == args.gpus     # resolve_launch: the run has exactly the ranks the flag asked for
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if not args.one_device and torch.cuda.device_count() < (local_rank + 1):
        raise SystemExit(f"rank {rank} (local rank {local_rank}) has no GPU: {torch.cuda.device_count()} visible, --gpus {args.gpus}")
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

    from bvh_amd import Context
    from bvh_amd._lib import TUNE_FLATTEN_LAZY

    # the engine enqueues on torch's current stream when that is a stream of its own; torch's DEFAULT stream has handle 0, for which
    # the ctx creates a non-blocking stream of its own — either way the timed region is bracketed by torch.cuda.synchronize(dev)
    # (device-wide), and everything the engine does for one step is on that one stream
    ctx = Context(local_rank, stream=torch.cuda.current_stream(dev).cuda_stream)
    ctx.set_tuning(TUNE_FLATTEN_LAZY, {"all": 0, "lazy": 1, "beside": 2, "eager": 3}[args.flat_array])
    for k, v in os.environ.items():   # developer A/B runs: BVH_TUNE_<knob>=<value> (tools/ab_tune.sh); results never depend on a knob
        if k.startswith("BVH_TUNE_"):
            ctx.set_tuning(int(k[9:]), int(v))
    env = dict(rank=rank, n_gpus=n_gpus, dev=dev, ctx=ctx, comm=None)
    # how many ranks this job REALLY has: the launcher's WORLD_SIZE (= --gpus), a sum over the torch.distributed group, and (in
    # `rccl`) the size the C ABI's RCCL communicator reports for itself
    ranks_seen, devices_seen = 1, [torch.cuda.current_device()]
    if n_gpus > 1:
        one = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(one.item())
        devices_seen = [None] * n_gpus
        dist.all_gather_object(devices_seen, f"{os.uname().nodename}:{torch.cuda.current_device()}")
    launch_obj = {"world_size": n_gpus, "ranks_seen": ranks_seen, "self_launched": bool(os.environ.get("BVH_BENCH_SELF_LAUNCHED")),
                  "backend": args.backend if n_gpus > 1 else None, "devices": devices_seen, "distinct_devices": len(set(devices_seen))}
    if ranks_seen != n_gpus:
        raise SystemExit(f"--gpus {args.gpus}: the process group holds {ranks_seen} ranks")

    dpath = detail_path(args)

    def emit(detail):
        """rank 0: the detailed object to the side file, the compact line — the ONE stdout line — to fd 1"""
        detail = dict(detail, detail=os.path.relpath(dpath, ROOT))
        try:
            with open(dpath, "w") as f:
                f.write(json.dumps(detail) + "\n")
        except OSError as e:
            detail["detail"] = f"not written: {e!r}"
        text = render_line(detail)
        os.write(json_fd, (text + "\n").encode())

    run = sec.Run(args, env, launch_obj, json_fd, run_workload, emit)
    out = run.line
    wl = sec.Workload(args.workload, args, args.dtype, rank, n_gpus, dev, ctx, scaling=args.scaling, rays=args.rays, harness=args.harness)
    torch.cuda.synchronize(dev)
    res = run.measure(wl, args.steps, args.warmup, True, publish=lambda r: out.update(run.compose(r, wl)))
    main_env = dict(env["last"])
    out.clear()
    out.update(run.compose(res, wl))

    if not args.no_parity and rank == 0:     # the GPU result of the headline batch against the CPU oracle, in-process
        from oracle import orc
        if wl.harness:
            out["parity"], out["cpu_harness"] = sec.check_parity_harness(wl, env, orc, min(wl.R, args.parity_max_rays))
        else:
            out["parity"] = sec.check_parity(wl, env, orc, min(wl.R, args.parity_max_rays))
        if out.get("roofline_build") and env["last"].get("oracle_levels"):
            out["roofline_build"] = sec.build_roofline(wl, main_env["phases"], env["last"]["oracle_levels"])
    if n_gpus == 1 and args.pipeline_streams > 1:
        out["pipelined"] = sec.pipelined(wl, args, env, local_rank)
    if n_gpus == 1 and args.workload == "cubes120k" and args.harness is None and not args.no_excluded:
        try:
            out["step_excludes"] = sec.measure_excluded(wl, args, env, out["ms_per_step"])
        except Exception as e:   # a supplementary figure must never take the headline down
            out["step_excludes"] = {"error": repr(e)}
    if not args.no_extra and args.workload == "cubes120k" and args.dtype == "f32" and args.harness is None:
        out["extra_configs"] = run.extras(args, rank, dev)
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = sec.cpu_baseline(wl, args)
            out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 2)
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": None, "kind": "port", "sample": "", "error": repr(e)[:300]}

    if rank == 0:
        sys.stdout.flush()
        emit(out)
    if env["comm"] is not None:
        env["comm"].close()
    if n_gpus > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
