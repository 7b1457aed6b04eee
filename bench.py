#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: Mrays/s for build + flatten + traverse.

One STEP = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
Bvh::build_par (SAH) → Bvh::flatten → FlatBvh::traverse of R rays, results left in HBM as CSR.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts the N ranks itself, below)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A run is never downgraded: `--gpus N` with no WORLD_SIZE in the environment re-executes itself under torch.distributed.run with N
ranks on 127.0.0.1 (resolve_launch / self_launch); with a WORLD_SIZE that is not N it stops with an error.  The JSON line says how
many ranks there really were: "launch" (ranks counted through the process group, the devices they sat on) and "rccl" (the size the
C ABI's RCCL communicator reports for itself, RCCL's version and library file).

Headline workload (`value`): BASELINE.json configs[1] — create_n_cubes(10 000) = 120 000 triangles f32/3D and
R = 1 000 000 create_ray rays PER GPU (weak scaling: rank r traverses rays [r*R, (r+1)*R) of the seed-0 stream).
The same JSON line also carries, under "extra_configs", driver-observed figures for the other BASELINE configs:
  configs[2]  stand-in scene (media/sponza.obj is not in the reference checkout), 10 M coherent primary rays   (N = 1)
  configs[3]  stand-in scene, 100 M incoherent create_ray rays STRONG-sharded over the N GPUs (N = 1: the 12.5 M-ray
              shard one GPU of eight owns), the scene built on rank 0 and RCCL-broadcast
  configs[4]  the configs[1] scene and rays in f64                                                            (N = 1)
`--workload` / `--dtype` / `--scaling` make any of them the headline of a run instead.

N > 1: one process per GPU.  The one exchange step of the path — rank 0's flattened tree to the peers — is an RCCL
broadcast issued by the C ABI itself (bvhgpu_bcast_known, csrc/comm.hip: straight out of / into the trees' HBM buffers
over xGMI); torch.distributed only carries the 128-byte RCCL id, the barrier and the max-over-ranks time.  The
alternative plan "replicate" (the build is deterministic: every rank runs it, no collective on the data path) is probed
next to it and the faster one is kept (`--scene-dist`); both probe times are reported.

The JSON line (rank 0) follows the task contract, plus:
  parity       — in-process diff of the GPU result against the CPU oracle on ALL rays of the headline batch
  roofline     — the dominant kernel against its BINDING resource (PMC-derived, profiles/*_bound.json) + the builder
  cpu_baseline — the oracle (C restatement of the reference, kind "port"), rebuilt on this box with -O3 -march=native,
                 timed on this box's host cores; rank 0, N = 1 only
  pipelined    — the same steps kept in flight on two streams by ONE host thread through the asynchronous C ABI
  step_excludes — what the timed step leaves out, each measured as the same step with it inside: ray generation, the FlatNode
                 array written eagerly, host-resident inputs and outputs
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
N_CU, N_SIMD, CLK = 256, 1024, 2.4e9
VALU_PEAK = N_SIMD * CLK / 2  # wave64 VALU instructions per second: one per 2 cycles per SIMD-32
LDS_PEAK = N_CU * CLK         # LDS-array cycles per second


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=["cubes120k", "cubes12m", "standin-primary", "standin-incoherent"], default="cubes120k")
    ap.add_argument("--cubes", type=int, default=10_000, help="cubes120k: create_n_cubes(n), 12 triangles each")
    ap.add_argument("--rays", type=int, default=None, help="rays per GPU per step (weak) / in total (strong); default per workload")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--harness", choices=["closest", "triangles"], default=None,
                    help="make the headline step the reference's WHOLE bench iteration (intersect_bh, testbase.rs:819-837): ray generation on the device "
                         "+ build + flatten + traverse + intersects_triangle on every candidate (N = 1; the default line carries these as extra_configs)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="weak: --rays per GPU; strong: --rays in total, sharded over the GPUs (default for standin-incoherent)")
    ap.add_argument("--scene-dist", choices=["auto", "bcast", "replicate", "bcast-torch"], default="auto",
                    help="N>1: RCCL-broadcast rank 0's flattened tree through the C ABI each step, or rebuild it on every rank; "
                         "auto times both before the warmup and keeps the faster plan; bcast-torch = scene blob over torch.distributed")
    ap.add_argument("--collective-timeout", type=float, default=60.0,
                    help="N>1, --scene-dist auto: seconds the exchange plan (RCCL communicator + broadcast steps) may take before the line "
                         "measured with the replicate plan is printed and the run ends")
    ap.add_argument("--settle-steps", type=int, default=300,
                    help="untimed steps before the W warmup steps (clock ramp after start-up: ~0.1 s of the headline step; at most 20 for batches above 2 M rays)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs sub-runs")
    ap.add_argument("--extras-timeout", type=float, default=300.0,
                    help="N > 1: seconds after which the extra_configs section is given up and the line measured so far is printed (a rank that "
                         "fails alone would leave the others in a barrier for ever)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-excluded", action="store_true", help="skip the step_excludes section (ray generation / host I/O / eager FlatNode array beside `value`)")
    ap.add_argument("--extra-steps", type=int, default=20)
    ap.add_argument("--parity-max-rays", type=int, default=200_000_000,
                    help="rays of a batch diffed against the oracle (default: every ray of every config, 100 M included)")
    ap.add_argument("--pipeline-streams", type=int, default=2,
                    help="N=1, reported beside `value`: the same K steps kept in flight on this many HIP streams by ONE host "
                         "thread (bvhgpu_*_async); 0 = skip")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N>1 (nccl == RCCL; gloo only for the one-GPU test of this script)")
    ap.add_argument("--one-device", action="store_true",
                    help="test only: every rank uses cuda:0 (needs --backend gloo: RCCL refuses two ranks on one GPU)")
    ap.add_argument("--cpu-sample-rays", type=int, default=1_000_000)
    ap.add_argument("--standin-detail", type=int, default=16)
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
def resolve_launch(gpus, environ):
    """How this invocation becomes `gpus` ranks.  Returns ("run", world) when this process IS one rank of a launched job (or
    the single rank of an N = 1 run) and ("spawn", gpus) when it has to launch the ranks itself.  A run can never silently
    shrink: --gpus N with a WORLD_SIZE that is set and is not N is an error, and --gpus N > 1 without WORLD_SIZE launches
    N ranks — it never falls through to one rank reporting n_gpus = 1 (the old code did exactly that)."""
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
            raise SystemExit(f"WORLD_SIZE={world} but {k} is not set: launch with torch.distributed.run (or plain `python bench.py --gpus {gpus}`)")
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


class Watchdog:
    """A section that contains a data-path collective nobody has ever run here on more than one GPU (the RCCL broadcast of the C ABI)
    must not be able to take the whole scaling record down with it: if the section does not finish in `seconds`, `on_fire` runs on a
    helper thread (the main thread is blocked inside a foreign call, with the GIL released) — rank 0 prints the line measured so far,
    every rank exits."""

    def __init__(self, seconds, on_fire):
        import threading
        self.seconds, self.on_fire = seconds, on_fire
        self.done = threading.Event()
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        if not self.done.wait(self.seconds):
            self.on_fire()

    def __enter__(self):
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self.done.set()
        return False


def timed_out_line(line, pending, what, after, rccl):
    """What the watchdog makes of the line measured so far when a section with a collective did not come back (a silent
    fall-back must be impossible to misread).  `pending`: the exchange plan in flight — {"res": the result dict whose scene_dist_plans the
    line shows, "plan", "stage", "workload"} — or None (the hang was elsewhere).  The plan that timed out is NAMED in scene_dist_plans with
    "timed_out": true, never just absent; `rccl` says how far the communicator got.  Returns the JSON text (None if the line could not be
    serialised: the main thread may be publishing into it while this runs — retried)."""
    if pending is not None:
        plans = pending["res"].setdefault("scene_dist_plans", {})
        plans[pending["plan"]] = {"timed_out": True, "after_s": after, "stage": pending["stage"], "workload": pending["workload"]}
        if line.get("scene_dist_plans") is None and line.get("workload_name") == pending["workload"]:
            line["scene_dist_plans"] = plans
    line["collective_watchdog"] = (f"{what} did not finish within {after:.0f} s: this line is what had been measured until then (the replicate plan has "
                                   "no data-path collective); scene_dist_plans names the plan that timed out")
    line["rccl"] = rccl
    for _ in range(20):
        try:
            return json.dumps(line)
        except RuntimeError:      # "dictionary changed size during iteration"
            time.sleep(0.01)
    return None


# ---------------------------------------------------------------------------------------------------------------------
class Workload:
    """scene + ray stream of one BASELINE config, resident in HBM; also what the CPU checker needs to redo it"""

    def __init__(self, name, args, dtype_name, rank, n_gpus, dev, ctx, scaling=None, rays=None, harness=None):
        import torch
        from bvh_amd import RayBatch, dist as bdist, scene, testbase as tb
        from bvh_amd._lib import RAY_F32, RAY_F64
        from bvh_amd.api import camera
        self.name, self.dtype_name = name, dtype_name
        self.np_dtype = np.float32 if dtype_name == "f32" else np.float64
        self.coherent = False
        self.cam = None
        # harness: the step is the reference's WHOLE bench iteration (intersect_bh, testbase.rs:819-837): the rays are generated on the
        # device inside the step and Ray::intersects_triangle runs on every candidate ("triangles": every Intersection kept, CSR order;
        # "closest": the nearest one per ray kept)
        self.harness = harness
        self.tag = name + (f"+{harness}" if harness else "")
        self.ctx = ctx
        if name in ("cubes120k", "cubes12m"):
            self.bounds = tb.default_bounds()
            n_cubes = args.cubes if name == "cubes120k" else 1_000_000
            self.tris_np, self.aabbs_np = tb.create_n_cubes(n_cubes, self.bounds)
            if not harness:
                self.tris_np = None     # (12 M triangles: 432 MB of vertices nobody reads)
            self.config_id, per = (1 if dtype_name == "f32" else 4) if name == "cubes120k" else None, rays or (1_000_000 if name == "cubes120k" else 10_000_000)
            self.scaling = scaling or "weak"
            self.label = f"create_n_cubes({n_cubes}) = {len(self.aabbs_np)} random-cube triangles"
        else:
            self.tris_np, self.aabbs_np, self.bounds = scene.parse_obj(scene.make_atrium_obj(args.standin_detail))
            self.label = (f"procedural atrium STAND-IN for media/sponza.obj (absent from the reference checkout), "
                          f"{len(self.aabbs_np)} triangles through the OBJ loader")
            if name == "standin-primary":
                self.config_id, per, self.coherent = 2, rays or 10_000_000, True    # primary rays: BVHGPU_TRAVERSE_COHERENT (how the walk hands its hits over)
                self.scaling = scaling or "weak"
                c = (self.bounds[:3] + self.bounds[3:]) * 0.5   # pinhole at the scene-bounds centre
                self.cam = camera(c, c + np.array([1.0, -0.15, 0.25]), fov_y_deg=70.0, aspect=4000 / 2500)
                self.W, self.H = 4000, 2500
            else:
                self.config_id, per = 3, rays or 100_000_000
                self.scaling = scaling or "strong"
        self.n_tri = len(self.aabbs_np)
        if self.scaling == "strong":
            self.total_rays = per
            self.first, self.R = bdist.strong_shard(rank, n_gpus, per)
        else:
            self.first, self.R = bdist.shard_range(rank, n_gpus, per)
            self.total_rays = per * n_gpus
        ray_size = (RAY_F32 if dtype_name == "f32" else RAY_F64).itemsize
        self.ray_size = ray_size
        self.aabbs = torch.from_numpy(self.aabbs_np.astype(self.np_dtype)).to(dev)
        self.rays_buf = torch.empty(max(self.R, 1) * ray_size, dtype=torch.uint8, device=dev)
        self.tris = torch.from_numpy(np.ascontiguousarray(self.tris_np, dtype=self.np_dtype).reshape(-1, 9)).to(dev) if harness else None
        self.rays = self.regen()

    def regen(self):
        """the batch's rays written into its HBM buffer by the device generators, on the context's stream, no host wait: Ray::new per ray
        (ray_impl.rs:70-80) behind create_ray (testbase.rs:687-691) or the primary-ray camera"""
        from bvh_amd import RayBatch
        if self.cam is not None:
            return RayBatch.primary(self.cam, self.W, self.H, self.first, self.R, self.rays_buf, self.np_dtype, self.ctx)
        return RayBatch.generate(self.first, self.R, self.bounds, self.rays_buf, self.np_dtype, self.ctx)

    def oracle_rays(self, orc, first, n):
        """the same rays from the oracle's restatement of the generators (f64: the f32 points widened BEFORE Ray::new, like the device)"""
        if self.cam is not None:
            return orc.primary_rays(self.cam, self.W, self.H, first, n, self.np_dtype)
        return orc.create_rays(first, n, self.bounds, self.np_dtype)

    def describe(self):
        kind = "coherent primary rays (4000x2500 pinhole)" if self.coherent else "create_ray rays (seed-0 stream)"
        step = "step = Bvh::build_par + flatten + FlatBvh::traverse (CSR hit lists in HBM)"
        if self.harness:
            step = ("step = the reference's whole bench iteration (intersect_bh, testbase.rs:819-837) behind a rebuild: ray generation on the device "
                    "(Ray::new) + Bvh::build_par + flatten + FlatBvh::traverse + Ray::intersects_triangle on every candidate — "
                    + ("every Intersection kept (CSR order, in HBM)" if self.harness == "triangles" else "the nearest Intersection per ray kept (in HBM)"))
        return ((f"configs[{self.config_id}]: " if self.config_id is not None else "beyond BASELINE (HBM regime): ") + f"{self.label}, {self.dtype_name}/3D; {self.total_rays} {kind} "
                f"{'in total, sharded over the GPUs' if self.scaling == 'strong' else 'per GPU'}; " + step)


def newest_bound(kernel_prefix, workload="cubes120k", dtype="f32", rays=1_000_000):
    """profiles/*_bound.json of the newest profile round that holds PMC counters for this kernel ON THIS WORKLOAD (the counters
    of a walk depend on the scene and the ray stream; files written before round 3 carry no workload tag and are configs[1] f32)"""
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bound.json")), reverse=True)   # newest round / version tag first (r3_… > r2_v8 > r2_v1)
    for f in found:
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if (j.get("workload", "cubes120k"), j.get("dtype", "f32")) != (workload, dtype):
            continue
        for k in j.get("kernels", []):
            if k.get("kernel", "").startswith(kernel_prefix):
                prof_rays = j.get("rays_per_launch", 1_000_000)
                if prof_rays != rays:   # the same walk over another batch size: per-launch counters are per ray to first order
                    k = dict(k)
                    for c in ("hbm_bytes", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"):
                        if k.get(c) is not None:
                            k[c] = k[c] * rays / prof_rays
                    k["scaled_from_rays"] = prof_rays
                return k, os.path.relpath(f, ROOT)
    return None, None


def bound_fractions(c, seconds):
    """PMC counters per launch (profiles/*_bound.json) against the time of one launch → fraction of each resource's peak"""
    out = {}
    if c.get("hbm_bytes") is not None:
        out["hbm"] = c["hbm_bytes"] / seconds / (HBM_PEAK_GBS * 1e9)
    if c.get("SQ_INSTS_VALU") is not None:
        out["valu"] = c["SQ_INSTS_VALU"] / seconds / VALU_PEAK
    if c.get("SQ_INSTS_LDS") is not None:
        out["lds"] = (c["SQ_INSTS_LDS"] * 4 + c.get("SQ_LDS_BANK_CONFLICT", 0)) / seconds / LDS_PEAK
    return out


# ---------------------------------------------------------------------------------------------------------------------
def run_workload(wl, args, env, steps, warmup, detailed, force_plan=None):
    """K timed steps of one workload on this rank's GPU (all ranks call it together) → result dict.  force_plan: measure exactly
    this scene-distribution plan (main() runs "replicate" first and the exchange plan afterwards, under a watchdog)"""
    import torch
    import torch.distributed as dist
    from bvh_amd import Bvh, FlatBvh, dist as bdist
    from bvh_amd._lib import REBROADCAST, TRAVERSE_CLOSEST, TRAVERSE_COHERENT, TRAVERSE_RAYS_READY, TRAVERSE_TRIANGLES, BvhGpuError
    rank, n_gpus, dev, ctx, comm = env["rank"], env["n_gpus"], env["dev"], env["ctx"], env["comm"]
    R, aabbs, rays = wl.R, wl.aabbs, wl.rays
    if wl.harness and n_gpus != 1:
        raise SystemExit("--harness is an N = 1 measurement (the triangle vertices are not part of the broadcast plan's step)")
    mode_flags = (TRAVERSE_COHERENT if wl.coherent else 0) | {None: 0, "closest": TRAVERSE_CLOSEST, "triangles": TRAVERSE_TRIANGLES}[wl.harness]

    if n_gpus == 1:
        plans = ["single"]
    elif force_plan is not None:
        plans = [force_plan]
    elif args.scene_dist == "auto":
        plans = (["bcast"] if comm is not None else ["bcast-torch"]) + ["replicate"]
    else:
        plans = ["bcast-torch" if (args.scene_dist == "bcast" and comm is None) else args.scene_dist]
    own_tree = rank == 0 or "replicate" in plans or n_gpus == 1
    bvh = Bvh.from_aabbs(aabbs, ctx) if own_tree else None
    if own_tree:
        bvh.flatten_in_place()
        if wl.harness:
            bvh.set_triangles(wl.tris)     # vertices of the shapes, resident in HBM like the AABBs (a rebuild of as many shapes keeps them)
    blob, peer = None, None
    if "bcast-torch" in plans:
        nbytes = bdist.broadcast_nbytes(bvh.scene_nbytes() if rank == 0 else 0, dev, 0)
        blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    state = {"plan": plans[0], "peer": None}

    def step():
        plan = state["plan"]
        # (the ray batch is resident in HBM since before the timed region: it does not depend on the rebuild enqueued in this step)
        flags = (TRAVERSE_COHERENT if wl.coherent else 0) | TRAVERSE_RAYS_READY
        if plan == "bcast":
            # the same asynchronous triple as on one GPU, with the exchange step in the middle and NO host synchronisation before the
            # final wait on any rank: rank 0 enqueues Bvh::build_par + flatten, the broadcast out of the tree's own buffers
            # (bvhgpu_bcast_known: the status header is composed on the device from the build's outcome) and its own walk; a peer
            # enqueues the receive and its walk.  The wait is the end of the step; BVHGPU_REBROADCAST (an unbalanced tree on a first
            # build: every rank sees it) repeats the exchange with the finished tree.
            tree, st, reb = bdist.broadcast_step(comm, rank, bvh if rank == 0 else state["peer"], aabbs if rank == 0 else None, rays,
                                                 wl.dtype_name, wl.n_tri, flags=flags)
            if rank != 0:
                state["peer"] = tree
            state["rebroadcasts"] = state.get("rebroadcasts", 0) + reb
            return st
        if plan == "bcast-torch":      # fallback transport: scene blob over torch.distributed (host round trips)
            if rank == 0:
                bvh.rebuild(aabbs, flatten=True)
                bvh.scene_export(blob)
            bdist.broadcast_scene(blob, 0)
            if rank != 0:
                state["peer"] = FlatBvh.scene_import(blob, blob.numel(), ctx, reuse=state["peer"])
            tree = bvh if rank == 0 else state["peer"]
            return tree.traverse_batch(rays, fetch=False, coherent=wl.coherent)[3]   # FlatBvh::traverse, CSR stays in HBM
        # single / replicate: FlatBvh::build (flat_bvh.rs:328-331) and FlatBvh::traverse enqueued back to back, ONE host round trip
        # per step: the wait validates the build, completes the batch and is the end of the step (nothing of the next step is in flight)
        if wl.harness:      # intersect_bh: the rays are made inside the step (k_gen_rays / k_gen_primary on the same stream), then walked + intersected
            wl.regen()
            bvh.rebuild_async(aabbs)
            return bvh.traverse_async(rays, flags=mode_flags).wait()
        bvh.rebuild_async(aabbs)
        return bvh.traverse_async(rays, flags=flags).wait()

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
            state["plan"] = pl
            step(); step()
            probe_ms[pl] = timed(5) / 5 * 1e3
        state["plan"] = min(plans, key=lambda q: probe_ms[q])
    # Settle the GPU's clocks before the W warmup steps: a 0.34 ms step timed over K = 20 steps right after start-up reads 2 % low
    # (0.3417 against 0.334–0.337 ms over K >= 100).  Untimed, the same step, a fixed count on every rank (no collective decides it).
    # (plans with an exchange step — a collective per step, possibly the slow torch transport — and big batches settle in 20 steps)
    settle = args.settle_steps if (wl.R <= 2_000_000 and state["plan"] in ("single", "replicate")) else min(args.settle_steps, 20)
    for _ in range(settle):
        step()
    for _ in range(warmup):
        step()
    elapsed = timed(steps)
    ms_per_step = elapsed * 1e3 / max(steps, 1)
    value = wl.total_rays / (ms_per_step * 1e-3) / 1e6   # Mrays/s, whole job (weak: N*R, strong: T)
    plan = state["plan"]

    # ---- per-phase HIP-event times (untimed extra steps) ----
    ctx.enable_timing(True)
    builder = plan not in ("bcast", "bcast-torch") or rank == 0
    tree = bvh if builder else state["peer"]
    ph = dict(build_ms=[], flatten_ms=[], traverse_kernel_ms=[], traverse_total_ms=[])
    stats_walk, walk_kernel = 0, ""
    for _ in range(max(5, min(steps, 20))):
        if builder:
            bvh.rebuild(aabbs)
            bvh.flatten_in_place()
        hh = tree.traverse_async(rays, flags=mode_flags)     # (the synchronous entry points cover index batches only without a fetch)
        hh.wait()
        stats_walk, walk_kernel = hh.walk_flags(), hh.walk_kernel()
        t = ctx.last_timings()
        for k in ph:
            ph[k].append(t[k])
    ctx.enable_timing(False)
    phases = {k: float(np.mean(v)) for k, v in ph.items()}
    if wl.harness:    # the generator's share of the step, timed alone (its launch is one of the step's)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(20):
            wl.regen()
        torch.cuda.synchronize(dev)
        phases["ray_gen_ms"] = (time.perf_counter() - t0) / 20 * 1e3
    # exact visit counters (reference-equivalent loop iterations, from the binary walk) for the algorithmic byte count
    stats = tree.traverse_batch(rays, stats=True, fetch=False, coherent=wl.coherent)[3]
    V, VL, H = stats["visited"], stats["leaf_visits"], stats["hits"]
    hits_all = H
    if n_gpus > 1:   # whole-job hit count (untimed): lets a reader check the shards against one process over all rays
        ht = torch.tensor([H], dtype=torch.int64, device=dev)
        dist.all_reduce(ht, op=dist.ReduceOp.SUM)
        hits_all = int(ht.item())
    out = {
        "workload": wl.tag, "harness": wl.harness, "config": wl.config_id, "dtype": wl.dtype_name, "value": round(value, 3), "unit": "Mrays/s",
        "ms_per_step": round(ms_per_step, 4), "steps": steps, "warmup": warmup, "settle_steps": settle, "scaling": wl.scaling, "triangles": wl.n_tri,
        "rays_this_rank": R, "rays_total": wl.total_rays, "scene_dist": plan,
        "phases_ms": {k: round(v, 4) for k, v in phases.items()},
        "hits_all_ranks": int(hits_all), "visited_per_ray": round(V / max(R, 1), 2),
        "scene_dist_probe_ms_per_step": {k: round(v, 4) for k, v in probe_ms.items()} or None,
        "describe": wl.describe(),
    }
    env["last"] = dict(bvh=bvh, tree=tree, stats=stats, phases=phases, builder=builder)
    if state.get("rebroadcasts"):
        out["rebroadcasts"] = state["rebroadcasts"]

    # ---- roofline of the dominant kernel against its BINDING resource (every workload), and of the builder (headline only) ----
    elem = 4 if wl.dtype_name == "f32" else 8
    flat_sz = 36 if wl.dtype_name == "f32" else 64
    # per ray  Ray in + V*FlatNode + V_leaf*shape AABB + CSR out 4*(H+1)
    algo_bytes = R * wl.ray_size + V * flat_sz + VL * 6 * elem + 4 * (H + R)
    if wl.harness == "triangles":   # + the triangle stage: 9 vertices read, Intersection{distance,u,v} written per candidate
        algo_bytes += H * (9 + 3) * elem
    elif wl.harness == "closest":   # + 9 vertices read per candidate; one Intersection + shape per ray instead of the CSR
        algo_bytes += H * 9 * elem + R * (3 * elem + 4) - 4 * (H + R)
    kern_s = phases["traverse_kernel_ms"] * 1e-3
    # the walk kernel's name as the library reports it for the timed batch shape (bvhgpu_hits_walk_kernel: spelled the way rocprofv3
    # prints it, so the counters of profiles/*_bound.json are looked up under the name the launch really had)
    kern_name = walk_kernel
    from bvh_amd._lib import WALK_F64_GUIDE
    guide_ran = bool(stats_walk & WALK_F64_GUIDE)   # bvhgpu_hits_walk_info: an f64 index batch walked by the f32 kernel over the guide boxes
    pmc, src = newest_bound(kern_name, wl.tag, wl.dtype_name, R)
    roof = {
        "kernel": kern_name, "kernel_ms": round(phases["traverse_kernel_ms"], 4),
        "algorithmic_bytes_per_launch": int(algo_bytes),
        "algorithmic_gbs": round(algo_bytes / kern_s / 1e9, 1),
        # SURVEY §8d's own figure, stated so that nobody has to derive it: algorithmic bytes / kernel time / 8 TB/s.  Above 1 (or above
        # `hbm_frac` by a wide margin) means the kernel does not do the §8d traffic at all: it reads the tree out of LDS and L2
        "algorithmic_frac": round(algo_bytes / kern_s / 1e9 / HBM_PEAK_GBS, 4),
        "algorithmic_note": "reference-algorithm bytes (SURVEY §8d: Ray + V x FlatNode + V_leaf x shape AABB + CSR) / kernel time against the 8 TB/s "
                            "HBM peak.  It is NOT a bandwidth: the walk tests four grandchildren per step out of an LDS- and L2-resident image, "
                            "so the bytes the reference's loop would move never cross the HBM interface — `hbm_frac` (PMC) is what does, `frac` is "
                            "the binding resource",
        "slab_tests_per_s": round(V / kern_s, 1),
        "slab_tests_note": "reference-equivalent: ray/AABB tests of the reference's loop on these rays (the binary STATS walk's visit count = the "
                           "oracle's) per second of the wide walk's kernel time — not a count of the tests the wide kernel executes",
        "visited": int(V), "leaf_visits": int(VL), "hits": int(H),
    }
    if wl.dtype_name == "f64":
        roof["f64_walk"] = ("guide: inner-node tests in f32 on boxes that contain the f64 ones, every leaf candidate decided by the f64 slab test "
                            "(the f64 rays are converted where the walk loads them: no separate copy pass)" if guide_ran else
                            "pure f64: every slab test of the walk in double precision (BVHGPU_TUNE_WIDE_F64_GUIDE = 0); valu_frac prices every "
                            "wave64 VALU instruction at 2 cycles, f64 arithmetic issues at half that rate, so it understates this kernel's VALU share by up to 2x")
    if pmc is not None:
        fr = bound_fractions(pmc, kern_s)
        bound = max(fr, key=fr.get)
        peak, unit, ach = {"hbm": (HBM_PEAK_GBS, "GB/s", pmc.get("hbm_bytes", 0) / kern_s / 1e9),
                           "valu": (VALU_PEAK / 1e9, "G wave-instr/s", pmc.get("SQ_INSTS_VALU", 0) / kern_s / 1e9),
                           "lds": (LDS_PEAK / 1e9, "G LDS-cycles/s", (pmc.get("SQ_INSTS_LDS", 0) * 4 + pmc.get("SQ_LDS_BANK_CONFLICT", 0)) / kern_s / 1e9)}[bound]
        roof.update({
            "bound": bound, "achieved": round(ach, 2), "peak": round(peak, 1), "unit": unit, "frac": round(fr[bound], 4),
            "traffic": pmc.get("hbm_bytes"), "hbm_frac": round(fr.get("hbm", 0), 4), "valu_frac": round(fr.get("valu", 0), 4),
            "lds_frac": round(fr.get("lds", 0), 4), "wait_frac": pmc.get("wait_frac"), "profile_kernel_us": pmc.get("avg_us"),
            "algorithmic_void": ("working set cache-resident: the HBM-side traffic (PMC) is %.3f of the algorithmic bytes, so algorithmic_frac prices "
                                 "bytes that never reach HBM" % (pmc["hbm_bytes"] / algo_bytes)) if pmc.get("hbm_bytes") and pmc["hbm_bytes"] < 0.5 * algo_bytes else None,
            "profile_rays_per_launch": pmc.get("scaled_from_rays", R),
            "source": f"{src}: separate rocprofv3 --pmc passes of this workload (per-launch means; FETCH_SIZE doubled per "
                      "MI355X_MICROARCH.md) over the live HIP-event kernel time; peaks: 8 TB/s HBM, 1024 SIMDs x 2.4 GHz / 2 cycles per "
                      "wave64 VALU instruction, 256 CUs x 2.4 GHz LDS-array cycles",
        })
    else:
        roof.update({"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                     "source": "no profiles/*_bound.json for this kernel and workload yet (tools/profile_round.sh <tag> --workload … writes it)"})
    out["roofline"] = roof
    if not detailed:
        return out
    if builder:
        out["roofline_build"] = build_roofline(wl, phases, None)
        out["build_levels"] = bvh.build_levels
    return out


def build_roofline(wl, phases, levels):
    """builder chain against the HBM roofline (SURVEY §8d build bytes); `levels` = mean leaf depth from the oracle's tree when the
    parity leg ran (sum over the tree levels of the shapes still being partitioned / N), else log2 N"""
    n = wl.n_tri
    elem = 4 if wl.dtype_name == "f32" else 8
    flat_sz = 36 if wl.dtype_name == "f32" else 64
    lv = levels if levels else float(np.log2(max(n, 2)))
    bbytes = (32 if elem == 4 else 56) * lv * n + (2 * n - 1) * (64 if elem == 4 else 112)
    fbytes = (2 * n - 1) * (64 if elem == 4 else 112) + (3 * n - 2) * flat_sz
    ms = phases["build_ms"] + phases["flatten_ms"]
    return {
        "kernels": "k_prep, k_level x (levels + 1), k_mid, k_small, k_flatten", "bound": "hbm",
        "algorithmic_bytes": int(bbytes + fbytes), "ms": round(ms, 4),
        "achieved": round((bbytes + fbytes) / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round((bbytes + fbytes) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "levels_priced": round(lv, 2), "levels_source": "oracle tree_stats (mean leaf depth)" if levels else "log2 N (no parity leg)",
        "note": "a chain of dependent launches over a cache-resident working set: latency-bound, not bandwidth-bound "
                f"(SURVEY §8d: sum over levels of live shapes = {lv:.1f} x N)",
    }


def check_parity(wl, env, orc, n_check, chunk=1_000_000):
    """The GPU result of this rank's WHOLE batch (CSR of the default walk fetched once; visit counters of the binary walk) against
    the oracle on the first n_check rays (default: all of them).  The oracle works through the rays in chunks of `chunk` (memory),
    each chunk diffed against its slice of the one GPU result; its time is reported: it is two oracle walks per ray."""
    from bvh_amd import RayBatch
    last = env["last"]
    tree = last["tree"]
    n = min(n_check, wl.R)
    a = wl.aabbs_np.astype(wl.np_dtype)
    ot = orc.build(a, threads=orc.max_threads(), schedule="fast")     # (byte-equal to the serial build: tests/test_oracle_golden.py)
    oflat = orc.flatten(ot.nodes)
    try:
        last["oracle_levels"] = float(orc.tree_stats(ot.nodes, a)["mean_leaf_depth"])   # = sum over the levels of live shapes / N
    except Exception:
        pass
    sub = RayBatch(n, wl.np_dtype, host=None, device=wl.rays_buf, device_ptr=wl.rays_buf.data_ptr())
    off, idx, _, _ = tree.traverse_batch(sub, coherent=wl.coherent)
    st = tree.traverse_batch(sub, stats=True, fetch=False, coherent=wl.coherent)[3]
    csr_equal, V, VL, H, t_or, n_chunks = True, 0, 0, 0, 0.0, 0
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        rays_o = wl.oracle_rays(orc, wl.first + c0, m)
        t0 = time.perf_counter()
        ooff, oidx, _, ost = orc.traverse_flat(oflat, a, rays_o, threads=orc.max_threads())
        t_or += time.perf_counter() - t0
        base = int(off[c0])
        csr_equal = csr_equal and bool(np.array_equal(off[c0:c0 + m + 1] - np.uint32(base), ooff)
                                       and np.array_equal(idx[base:int(off[c0 + m])], oidx))
        V += ost["visited"]; VL += ost["leaf_visits"]; H += ost["hits"]; n_chunks += 1
    cnt_equal = bool(st["visited"] == V and st["leaf_visits"] == VL and st["hits"] == H and len(idx) == H)
    nodes_equal = None
    if last["builder"] and last["bvh"] is not None:
        nodes_equal = bool(last["bvh"].nodes.tobytes() == ot.nodes.tobytes())
    return {"checked_rays": int(n), "rays_this_rank": int(wl.R), "equal": bool(csr_equal and cnt_equal and nodes_equal is not False),
            "csr_offsets_and_indices_equal": csr_equal, "visit_counters_equal": cnt_equal, "bvh_nodes_equal": nodes_equal,
            "hits": int(H), "against": "oracle (C restatement of bvh_node.rs / flat_bvh.rs, see oracle/bvh_oracle.h)",
            "oracle_chunks": n_chunks, "oracle_traverse_s": round(t_or, 4)}


def check_parity_harness(wl, env, orc, n_check, chunk=1_000_000, cpu_sample=1_000_000):
    """The harness step's result on this rank's WHOLE batch against the oracle's restatement of the same loop (testbase.rs:826-836 behind
    FlatBvh::traverse): "closest" — (distance, u, v, shape) of every ray, bit for bit; "triangles" — CSR offsets / indices and the
    Intersection of every candidate, bit for bit.  Also times the oracle's whole loop (orc.harness_loop = intersect_bh: ray generation,
    one walk per ray into a growable list, intersects_triangle on every candidate) on `cpu_sample` rays for the CPU figure beside it."""
    from bvh_amd import RayBatch
    last = env["last"]
    tree = last["tree"]
    n = min(n_check, wl.R)
    a = wl.aabbs_np.astype(wl.np_dtype)
    tris = np.ascontiguousarray(wl.tris_np, dtype=wl.np_dtype).reshape(-1, 9)
    t0 = time.perf_counter()
    ot = orc.build(a, threads=orc.max_threads(), schedule="fast")
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    oflat = orc.flatten(ot.nodes)
    t_flat = time.perf_counter() - t0
    sub = RayBatch(n, wl.np_dtype, host=None, device=wl.rays_buf, device_ptr=wl.rays_buf.data_ptr())
    if wl.harness == "closest":
        g_isect, g_shape, _ = tree.closest_hits(sub, coherent=wl.coherent)
    else:
        g_off, g_idx, g_isect, _ = tree.intersect_triangles(sub, coherent=wl.coherent)
    equal, H, n_chunks = True, 0, 0
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        rays_o = wl.oracle_rays(orc, wl.first + c0, m)
        ooff, oidx, _, ost = orc.traverse_flat(oflat, a, rays_o, threads=orc.max_threads())
        o_isect, o_closest, o_prim = orc.triangle_stage(tris, rays_o, ooff, oidx)
        if wl.harness == "closest":
            equal = equal and g_isect[c0:c0 + m].tobytes() == o_closest.tobytes() and bool(np.array_equal(g_shape[c0:c0 + m], o_prim))
        else:
            base, end = int(g_off[c0]), int(g_off[c0 + m])
            equal = (equal and bool(np.array_equal(g_off[c0:c0 + m + 1] - np.uint32(base), ooff) and np.array_equal(g_idx[base:end], oidx))
                     and g_isect[base:end].tobytes() == o_isect.tobytes())
        H += ost["hits"]; n_chunks += 1
    nodes_equal = bool(last["bvh"].nodes.tobytes() == ot.nodes.tobytes()) if last["builder"] and last["bvh"] is not None else None
    out = {"checked_rays": int(n), "rays_this_rank": int(wl.R), "equal": bool(equal and nodes_equal is not False), "candidates": int(H),
           "what": ("closest (distance, u, v, shape) of every ray" if wl.harness == "closest" else "CSR + Intersection{distance,u,v} of every candidate")
                   + ", byte for byte", "bvh_nodes_equal": nodes_equal,
           "against": "oracle: traverse_flat + triangle_stage (restatement of flat_bvh.rs:396-431 + testbase.rs:826-836 + ray_impl.rs:154-213)"}
    cpu = None
    if wl.dtype_name == "f32":   # the reference's harness is f32
        ns = min(cpu_sample, wl.R)
        best, best_th = 1e9, 0
        cores = orc.max_threads()
        for th in sorted({16, 32, 64, 128, cores} & set(range(1, cores + 1))):
            t0 = time.perf_counter()
            orc.harness_loop(oflat, a, tris, wl.first, ns, wl.bounds, wl.cam, getattr(wl, "W", 0), getattr(wl, "H", 0), threads=th)
            dt = time.perf_counter() - t0
            if dt < best:
                best, best_th = dt, th
        total = t_build + t_flat + best * (wl.R / ns)
        cpu = {"value": round(wl.R / total / 1e6, 4), "unit": "Mrays/s", "cores": best_th, "kind": "port",
               "sample": f"oracle (C restatement, portable -O2 build), the same step: build {t_build * 1e3:.1f} ms (scalable schedule, all cores) + flatten "
                         f"{t_flat * 1e3:.1f} ms + intersect_bh on {ns} of the {wl.R} rays ({best * 1e3:.1f} ms on {best_th} threads: ray generation, one walk "
                         "per ray into a growable list, intersects_triangle on every candidate), scaled to the batch",
               "loop_ms_scaled": round(best * (wl.R / ns) * 1e3, 2), "build_ms": round(t_build * 1e3, 2)}
    return out, cpu


def measure_excluded(wl, args, env, ms_step):
    """The three things the timed step of `value` does not contain, each as the SAME step with that thing put inside, K steps timed the
    same way (device sync on both sides):
      with_ray_gen      Ray::new for every ray of the batch (ray_impl.rs:70-80 via create_ray, testbase.rs:687-691: the reference's bench
                        iteration starts with it) generated on the device inside the step — k_gen_rays on the step's stream
      with_flat_array   the FlatNode array in the reference's layout (flat_bvh.rs:60-143) written by every step's flatten
                        (BVHGPU_TUNE_FLATTEN_LAZY = 0) instead of on first use
      host_io           shape AABBs and rays start in HOST memory, the CSR ends in host memory: what GpuBvh::build + traverse_batch of the
                        Rust shim costs a caller whose data lives in Vecs (rust/bvh-mi355x/src/lib.rs) — upload, step, download"""
    import torch
    from bvh_amd import Bvh, RayBatch
    from bvh_amd._lib import TRAVERSE_RAYS_READY, TUNE_FLATTEN_LAZY
    dev, ctx = env["dev"], env["ctx"]
    K = max(args.steps, 100)
    bvh = Bvh.from_aabbs(wl.aabbs, ctx)
    bvh.flatten_in_place()

    def timed(fn, k):
        for _ in range(5):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / k * 1e3

    def entry(ms, what):
        return {"value": round(wl.R / (ms * 1e-3) / 1e6, 3), "unit": "Mrays/s", "ms_per_step": round(ms, 4),
                "delta_ms_vs_value": round(ms - ms_step, 4), "what": what}

    res = {"steps": K}

    def step_gen():
        bvh.rebuild_async(wl.aabbs)
        RayBatch.generate(wl.first, wl.R, wl.bounds, wl.rays_buf, wl.np_dtype, ctx)   # the same buffer, rewritten every step
        return bvh.traverse_async(wl.rays, flags=0).wait()
    res["with_ray_gen"] = entry(timed(step_gen, K), "create_ray + Ray::new of all rays on the device inside every step (k_gen_rays), then the step of `value`")

    ctx.set_tuning(TUNE_FLATTEN_LAZY, 0)
    try:
        def step_eager():
            bvh.rebuild_async(wl.aabbs)
            return bvh.traverse_async(wl.rays, flags=TRAVERSE_RAYS_READY).wait()
        res["with_flat_array"] = entry(timed(step_eager, K), "every flatten also writes the reference-layout FlatNode array + the folded binary array "
                                                             "(BVHGPU_TUNE_FLATTEN_LAZY = 0); `value` writes them on first use (bvhgpu_flat_nodes, a binary walk …)")
    finally:
        ctx.set_tuning(TUNE_FLATTEN_LAZY, 1)

    # host I/O: pageable numpy arrays, like a Rust caller's Vecs
    a_host = np.ascontiguousarray(wl.aabbs_np.astype(wl.np_dtype))
    rays_host = torch.empty(wl.R * wl.ray_size, dtype=torch.uint8)
    rays_host.copy_(wl.rays_buf[:wl.R * wl.ray_size])
    from bvh_amd._lib import RAY_F32, RAY_F64
    rb_host = RayBatch(wl.R, wl.np_dtype, host=rays_host.numpy().view(RAY_F32 if wl.dtype_name == "f32" else RAY_F64))
    nbytes = {"aabbs_up": int(a_host.nbytes), "rays_up": int(wl.R * wl.ray_size)}

    def step_host():
        bvh.rebuild(a_host, flatten=True)
        off, idx, _, _ = bvh.traverse_batch(rb_host, fetch=True)
        return off, idx
    off, idx = step_host()
    nbytes["csr_down"] = int(off.nbytes + idx.nbytes)
    kh = max(10, min(K, 30))
    e = entry(timed(step_host, kh), "AABBs + rays uploaded from pageable host memory and the CSR fetched to host memory inside every step "
                                   "(bvhgpu_rebuild_flat(HOST) + bvhgpu_traverse(HOST) + bvhgpu_hits_fetch(HOST): the Rust shim's GpuBvh::build + traverse_batch)")
    e["steps"] = kh
    e["bytes_per_step"] = nbytes
    e["pcie_gbs"] = round(sum(nbytes.values()) / (e["ms_per_step"] * 1e-3) / 1e9, 2)
    res["host_io"] = e
    bvh.close()
    return res


# ---------------------------------------------------------------------------------------------------------------------
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

    from bvh_amd import Bvh, Context, dist as bdist
    from bvh_amd._lib import TRAVERSE_RAYS_READY as RAYS_READY
    from bvh_amd.api import _Hits

    # the engine enqueues on torch's current stream when that is a stream of its own; torch's DEFAULT stream has handle 0, for which
    # the ctx creates a non-blocking stream of its own — either way the timed region is bracketed by torch.cuda.synchronize(dev)
    # (device-wide), and everything the engine does for one step is on that one stream
    stream = torch.cuda.current_stream(dev)
    ctx = Context(local_rank, stream=stream.cuda_stream)
    for k, v in os.environ.items():   # developer A/B runs: BVH_TUNE_<knob number>=<value> (tools/ab_tune.sh); results never depend on a knob
        if k.startswith("BVH_TUNE_"):
            ctx.set_tuning(int(k[9:]), int(v))
    env = dict(rank=rank, n_gpus=n_gpus, dev=dev, ctx=ctx, comm=None)
    # how many ranks this job REALLY has, counted three ways: the launcher's WORLD_SIZE (= --gpus, resolve_launch), a sum over the
    # torch.distributed group, and the size the C ABI's RCCL communicator reports for itself (bvhgpu_comm_info)
    ranks_seen, devices_seen = 1, [torch.cuda.current_device()]
    if n_gpus > 1:
        one = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(one.item())
        dl = [None] * n_gpus
        dist.all_gather_object(dl, f"{os.uname().nodename}:{torch.cuda.current_device()}")
        devices_seen = dl
    launch_obj = {"world_size": n_gpus, "ranks_seen": ranks_seen, "self_launched": bool(os.environ.get("BVH_BENCH_SELF_LAUNCHED")),
                  "backend": args.backend if n_gpus > 1 else None, "devices": devices_seen,
                  "distinct_devices": len(set(devices_seen))}
    if ranks_seen != n_gpus:
        raise SystemExit(f"--gpus {args.gpus}: the process group holds {ranks_seen} ranks")

    line = {}            # the JSON line as far as it has been measured: what the watchdog prints if an exchange section hangs
    xstate = {"comm_err": None, "tried_comm": False, "pending": None}

    def emit_and_exit(what, seconds=None):
        """runs on the watchdog's helper thread: print the line as far as it has been measured and end the process — whatever happens on
        the way (os._exit sits in a `finally`: an exception here must not bring back the hang the watchdog exists to prevent)"""
        code = 3
        try:
            after = seconds if seconds is not None else args.collective_timeout
            try:
                rccl = env["comm"].info() if env["comm"] is not None else {"nranks": None, "formed": False, "error": xstate["comm_err"]}
            except Exception as e:
                rccl = {"nranks": None, "formed": env["comm"] is not None, "error": repr(e)}
            text = timed_out_line(line, xstate.get("pending"), what, after, rccl)
            if text is not None and line.get("value") is not None:
                if rank == 0:
                    os.write(json_fd, (text + "\n").encode())
                code = 0
        finally:
            os._exit(code)

    def make_comm():
        """the RCCL communicator of the C ABI (torch.distributed only carries the 128-byte id) — made AFTER the replicate plan has been
        measured, inside the watchdog: it is the first thing in the run that has never been exercised with more than one rank"""
        if xstate["tried_comm"] or args.backend != "nccl":
            return env["comm"]
        xstate["tried_comm"] = True
        comm = None
        try:
            comm = bdist.Communicator.from_torch_distributed(ctx, dev)
        except Exception as e:   # keep the run alive on the torch transport, and say so
            xstate["comm_err"] = repr(e)
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            comm = None
        env["comm"] = comm
        return comm

    def measure(w, steps, warmup, detailed, publish=None):
        """one workload → (result dict of the plan that is reported, probe {plan: ms per step}).  N = 1 or an explicit --scene-dist: one
        run.  N > 1 with --scene-dist auto: the replicate plan first (every rank builds: no data-path collective, so this result is safe
        — `publish` puts it into the line at once), then the exchange plan under the watchdog; the faster one is reported."""
        if n_gpus == 1 or args.scene_dist != "auto":
            if n_gpus > 1 and args.scene_dist == "bcast":
                with Watchdog(args.collective_timeout, lambda: emit_and_exit("forming the RCCL communicator")):
                    make_comm()
            return run_workload(w, args, env, steps, warmup, detailed), None
        res = run_workload(w, args, env, steps, warmup, detailed, force_plan="replicate")
        keep = dict(env["last"])
        pick = lambda r: {k: r[k] for k in ("value", "ms_per_step", "phases_ms", "hits_all_ranks") if k in r}
        res["scene_dist_plans"] = {"replicate": pick(res)}
        if publish:
            publish(res)
        guess = "bcast" if args.backend == "nccl" else "bcast-torch"
        xstate["pending"] = {"res": res, "plan": guess, "stage": "forming the RCCL communicator", "workload": w.tag}
        with Watchdog(args.collective_timeout, lambda: emit_and_exit(f"the exchange plan of {w.name}")):
            if os.environ.get("BVH_BENCH_TEST_HANG_EXCHANGE"):   # tests: a collective that never returns (tests/test_gpu_dist.py)
                time.sleep(10 ** 6)
            comm = make_comm()
            xplan = "bcast" if comm is not None else "bcast-torch"
            xstate["pending"].update(plan=xplan, stage="the exchange plan's steps (communicator formed)")
            res_x = run_workload(w, args, env, steps, warmup, detailed, force_plan=xplan)
        xstate["pending"] = None
        probe = {"replicate": res["ms_per_step"], xplan: res_x["ms_per_step"]}
        both = {"replicate": pick(res), xplan: pick(res_x)}
        if res_x["ms_per_step"] < res["ms_per_step"]:
            res = res_x
        else:
            env["last"] = keep
        res["scene_dist_probe_ms_per_step"] = probe
        res["scene_dist_plans"] = both
        return res, probe

    def compose(res):
        out = {
            "metric": "Mrays/s (build+traverse)", "value": res["value"], "unit": "Mrays/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "settle_steps": res.get("settle_steps"),
            "settle_note": "untimed steps of the same kind run BEFORE the W warmup steps (clock ramp after start-up: the 0.33 ms step read 2-3 % "
                           "low without them); --settle-steps 0 switches them off",
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True, "scaling": wl.scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "workload_name": wl.tag, "harness": wl.harness,
            "config": {
                "workload": wl.describe(), "triangles": wl.n_tri, "rays_per_gpu": wl.R, "rays_total": wl.total_rays,
                "scene_dist": res["scene_dist"],
                "parallelism": f"rays sharded x{n_gpus}" + {
                    "single": "", "bcast": ", flattened tree RCCL-broadcast from rank 0 every step by the C ABI (bvhgpu_bcast_known)",
                    "bcast-torch": ", scene blob broadcast from rank 0 every step over torch.distributed",
                    "replicate": ", every rank rebuilds the scene (deterministic build, no data-path collective)"}[res["scene_dist"]],
            },
            "phases_ms": res["phases_ms"], "build_levels": res.get("build_levels"), "hits_all_ranks": res["hits_all_ranks"],
            "scene_dist_probe_ms_per_step": res.get("scene_dist_probe_ms_per_step"), "scene_dist_plans": res.get("scene_dist_plans"),
            "roofline": res["roofline"], "roofline_build": res.get("roofline_build"),
            "launch": launch_obj,
            # None: no RCCL communicator in this run (N = 1, --backend gloo, --scene-dist replicate, or rccl_comm_error)
            "rccl": env["comm"].info() if env["comm"] is not None else None,
        }
        if xstate["comm_err"]:
            out["rccl_comm_error"] = xstate["comm_err"]
        return out

    wl = Workload(args.workload, args, args.dtype, rank, n_gpus, dev, ctx, scaling=args.scaling, rays=args.rays, harness=args.harness)
    torch.cuda.synchronize(dev)
    res, _ = measure(wl, args.steps, args.warmup, True, publish=lambda r: line.update(compose(r)))
    main_env = dict(env["last"])
    out = line
    out.clear()
    out.update(compose(res))

    # ---- parity: the GPU result of the headline batch against the CPU oracle, in-process ----
    parity_run = None
    if not args.no_parity and rank == 0:
        from oracle import orc
        if wl.harness:
            parity_run, cpu_h = check_parity_harness(wl, env, orc, min(wl.R, args.parity_max_rays))
            out["cpu_harness"] = cpu_h
        else:
            parity_run = check_parity(wl, env, orc, min(wl.R, args.parity_max_rays))
        out["parity"] = parity_run
        if out.get("roofline_build") and env["last"].get("oracle_levels"):
            out["roofline_build"] = build_roofline(wl, main_env["phases"], env["last"]["oracle_levels"])

    # ---- supplementary: independent steps kept in flight on several HIP streams by ONE host thread (N = 1) ----
    # `value` above is the time of K steps issued one after the other, each waited for.  Steps are independent (each rebuilds the
    # scene from the shape AABBs) and the builder's ~20 small dependent kernels leave most CUs idle, so a frame loop keeps the
    # next frame's build in flight while the current frame traces: bvhgpu_rebuild_flat_async + bvhgpu_traverse_async on S
    # contexts (S streams), bvhgpu_hits_wait only when a lane's result object is needed again.
    if n_gpus == 1 and args.pipeline_streams > 1:
        S = args.pipeline_streams
        lanes = []
        for j in range(S):
            c = Context(local_rank)                       # its own non-blocking HIP stream
            tr = Bvh.from_aabbs(wl.aabbs, c)
            tr.flatten_in_place()
            lanes.append([c, tr, _Hits(c), False])
        for k in range(3 * S):
            ln = lanes[k % S]
            ln[1].rebuild_async(wl.aabbs); ln[1].traverse_async(wl.rays, ln[2], flags=RAYS_READY); ln[2].wait()
        torch.cuda.synchronize(dev)
        K = args.steps
        hits_p = []
        t0 = time.perf_counter()
        for k in range(K):
            ln = lanes[k % S]
            if ln[3]:
                hits_p.append(ln[2].wait()["hits"])
            ln[1].rebuild_async(wl.aabbs)
            ln[1].traverse_async(wl.rays, ln[2], flags=RAYS_READY)
            ln[3] = True
        for ln in lanes:
            if ln[3]:
                hits_p.append(ln[2].wait()["hits"])
        torch.cuda.synchronize(dev)
        dtp = time.perf_counter() - t0
        out["pipelined"] = {
            "streams": S, "host_threads": 1, "steps": K, "value": round(K * wl.R / dtp / 1e6, 3), "unit": "Mrays/s",
            "ms_per_step": round(dtp * 1e3 / K, 4), "hits_every_step_equal": bool(len(set(hits_p)) == 1 and len(hits_p) == K),
            "hits": hits_p[0] if hits_p else None,
            "note": f"the same {K} steps issued by ONE host thread on {S} HIP streams through the asynchronous C ABI "
                    "(bvhgpu_rebuild_flat_async / bvhgpu_traverse_async / bvhgpu_hits_wait): the build of one step overlaps the traversal "
                    "of another; throughput of independent steps, not the latency of one — never reported as `value`",
        }
        for c, tr, h, _ in lanes:
            h.close(); tr.close(); c.close()

    # (A `back_to_back` figure — one stream, the host one step behind — was reported in round 4 and withdrawn: bvhgpu_hits_wait
    #  synchronises the whole stream and a rebuild completes the batches pending on its tree, so that loop was the `value` loop again
    #  (ADVICE r4).  What overlapping steps buy is the `pipelined` figure above; what the host costs inside a step is in the kernel
    #  trace: ≈ 12 µs between the last kernel of a step and the first of the next, EXPERIMENTS.md "Where the step's 342 µs are".)

    # ---- what the timed step leaves out, measured beside it (N = 1) — never reported as `value` ----
    if n_gpus == 1 and args.workload == "cubes120k" and args.harness is None and not args.no_excluded:
        try:
            out["step_excludes"] = measure_excluded(wl, args, env, out["ms_per_step"])
        except Exception as e:   # a supplementary figure must never take the headline down
            out["step_excludes"] = {"error": repr(e)}

    # ---- the other BASELINE configs, the reference's whole harness loop and a scene beyond the caches, driver-observed in the same line ----
    if not args.no_extra and args.workload == "cubes120k" and args.dtype == "f32" and args.harness is None:
        extras = []
        out["extra_configs"] = extras
        E = lambda name, dt="f32", scaling=None, rays=None, harness=None, **kw: dict(name=name, dt=dt, scaling=scaling, rays=rays, harness=harness, **kw)
        if n_gpus == 1:
            plan = [
                # intersect_bh (testbase.rs:819-837) whole, behind a rebuild: ray generation + build + flatten + walk + triangle stage
                E("cubes120k", harness="closest"), E("cubes120k", harness="triangles"), E("standin-primary", harness="closest"),
                E("standin-primary"), E("standin-incoherent", scaling="weak", rays=12_500_000), E("cubes120k", dt="f64"),
                E("standin-incoherent", scaling="strong", rays=100_000_000),   # configs[3] whole on ONE GPU: the N = 1 point of the strong curve
                # the regime the north star's HBM language is about: a tree far beyond L2 + MALL (create_n_cubes(1 000 000) = 12 M triangles)
                E("cubes12m", rays=10_000_000, parity_rays=1_000_000),
            ]
            only = os.environ.get("BVH_BENCH_EXTRAS")     # developer runs: comma-separated entry numbers of the list above
            if only:
                plan = [plan[int(k)] for k in only.split(",")]
        else:
            # (tests shrink the stream: BVH_BENCH_STRONG_RAYS; the driver's run keeps BASELINE's 100 M)
            plan = [E("standin-incoherent", scaling="strong", rays=int(os.environ.get("BVH_BENCH_STRONG_RAYS", 100_000_000)))]
        # N > 1: the section's barriers and all-reduces are only safe while every rank gets through it — a rank that fails alone (its
        # `except` below skips the collectives) would leave the others waiting for ever, and the headline with them
        import contextlib
        guard = (Watchdog(args.extras_timeout, lambda: emit_and_exit("the extra_configs section", args.extras_timeout)) if n_gpus > 1
                 else contextlib.nullcontext())
        with guard:
            for e in plan:
                name, dt, scaling, nrays = e["name"], e["dt"], e["scaling"], e["rays"]
                try:
                    w2 = Workload(name, args, dt, rank, n_gpus, dev, ctx, scaling=scaling, rays=nrays, harness=e["harness"])
                    if name == "standin-incoherent" and n_gpus == 1 and scaling == "weak":   # the shard rank 5 of 8 owns (tests/test_gpu_scene.py checks the same one)
                        w2.first = 62_500_000
                        w2.rays = w2.regen()
                    provisional = []

                    def publish_extra(r):      # N > 1: the replicate result is on the line before the exchange plan is tried
                        provisional.append(r)
                        extras.append(r)
                    r2, _ = measure(w2, args.extra_steps, 3, False, publish=publish_extra)
                    for r in provisional:       # (replaced by the finished entry below)
                        if r in extras:
                            extras.remove(r)
                    if name == "standin-incoherent" and n_gpus == 1:
                        r2["note"] = ("one GPU's share of configs[3]: rays [62.5 M, 75 M) of the 100 M-ray stream (rank 5 of 8)" if scaling == "weak" else
                                      "configs[3] whole: all 100 M rays of the stream on one GPU in one batch — the N = 1 point of the strong-scaling "
                                      "curve whose N > 1 points the same entry carries when bench.py runs with --gpus N")
                    if name == "standin-incoherent" and scaling == "strong" and nrays == 100_000_000 and args.standin_detail == 16:
                        # the whole stream's hit count as one GPU produced it with oracle parity on all 100 M rays (BENCH_r03 extra_configs):
                        # the shards of an N > 1 run must add up to exactly this — under EVERY plan that was measured
                        r2["hits_n1_reference"] = 457_389_170
                        r2["hits_match_n1_reference"] = bool(r2["hits_all_ranks"] == 457_389_170)
                        for pl in (r2.get("scene_dist_plans") or {}).values():
                            if "hits_all_ranks" in pl:
                                pl["hits_match_n1_reference"] = bool(pl["hits_all_ranks"] == 457_389_170)
                    if name == "cubes12m":
                        r2["note"] = ("NOT a BASELINE config: the scene where SURVEY §8d's HBM roofline applies — 12 M triangles (node + shape arrays ≈ 1.9 GB, far "
                                      "beyond L2 + MALL), 10 M create_ray rays; roofline.hbm_frac is this walk's own PMC pass when profiles/ holds one; parity on "
                                      f"the first {e['parity_rays']} rays of the batch (the oracle's 12 M-triangle tree is built once for it)")
                    if dt == "f64":
                        r2["note"] = ("tree, rays, builder and every test that decides a hit in f64; the walk's inner-node tests run on f32 boxes that contain "
                                      "the f64 ones (BVHGPU_TUNE_WIDE_F64_GUIDE, DESIGN.md §4 \"f64 guide walk\"): same lists, checked against the f64 oracle below")
                    if rank == 0 and not args.no_parity:
                        from oracle import orc
                        n_par = min(w2.R, args.parity_max_rays, e.get("parity_rays") or w2.R)
                        if w2.harness:
                            r2["parity"], r2["cpu_harness"] = check_parity_harness(w2, env, orc, n_par)
                            if r2["cpu_harness"]:
                                r2["speedup_vs_cpu_harness"] = round(r2["value"] / r2["cpu_harness"]["value"], 2)
                        else:
                            r2["parity"] = check_parity(w2, env, orc, n_par)
                    if dt == "f64":
                        # the same step with EVERY slab test of the walk in double precision (BASELINE configs[4] names "double-precision slab
                        # test"): k_traverse_wide<double, …>, its own timing, roofline (its own counter passes when profiles/ holds them) and parity
                        from bvh_amd._lib import TUNE_WIDE_F64_GUIDE
                        ctx.set_tuning(TUNE_WIDE_F64_GUIDE, 0)
                        try:
                            r3 = run_workload(w2, args, env, args.extra_steps, 3, detailed=False)
                            if rank == 0 and not args.no_parity:
                                r3["parity"] = check_parity(w2, env, orc, min(w2.R, args.parity_max_rays))
                            r2["pure_f64_walk"] = {k: r3[k] for k in ("value", "unit", "ms_per_step", "steps", "phases_ms", "hits_all_ranks", "roofline", "parity") if k in r3}
                        finally:
                            ctx.set_tuning(TUNE_WIDE_F64_GUIDE, 1)
                    extras.append(r2)
                    del w2
                    env["last"] = {}
                    torch.cuda.empty_cache()
                except Exception as ex:   # an extra config must never take the headline line down
                    import traceback
                    extras.append({"workload": name, "dtype": dt, "harness": e["harness"], "error": repr(ex), "where": traceback.format_exc(limit=3)[-400:]})
        out["extra_configs"] = extras

    # ---- CPU baseline: the oracle (C port of the reference algorithm) on this box's host cores ----
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        from oracle import orc
        cores = orc.max_threads()
        a = wl.aabbs_np.astype(wl.np_dtype)
        ns = min(args.cpu_sample_rays, wl.R)
        rr = wl.oracle_rays(orc, wl.first, ns)
        n1 = max(ns // 16, 1000)
        tb_par, par_threads, tb_task, task_threads, tb_ser, tf, tt_all, trav_threads, tt_1, tt_csr = 1e9, 0, 1e9, 0, 1e9, 1e9, 1e9, cores, 1e9, 1e9
        builds_timed = []
        # the portable -O2 build that travelled with the repository, then the -O3 -march=native build made on THIS box
        # (SURVEY §8d); every phase keeps its best time over the two (neither flag set wins everywhere)
        for which in ("O2-portable", "O3-native"):
            if which == "O3-native" and not orc.use_native():
                break
            builds_timed.append(which)
            orc.build(a)                                    # warm the allocator and the page cache
            # Bvh::build_par on the host cores.  "fast": the schedule that scales like rayon's work stealing (big nodes split by the whole
            # team, then one parallel for over the subtrees; byte-equal arrays) — the figure the baseline uses.  "tasks": the literal
            # restatement of rayon_executor's join recursion as OpenMP tasks, which libgomp's single task queue caps at ~4 threads — kept
            # beside it so that nobody has to guess what round 4's 20 ms were.
            for th in sorted({8, 16, 32, 64, 96, 128, cores} & set(range(1, cores + 1))):
                for _ in range(3):
                    t0 = time.perf_counter(); ot = orc.build(a, threads=th, schedule="fast"); dt = time.perf_counter() - t0
                    if dt < tb_par:
                        tb_par, par_threads = dt, th
            for th in sorted({4, 8, 16} & set(range(1, cores + 1))):
                t0 = time.perf_counter(); orc.build(a, threads=th); dt = time.perf_counter() - t0
                if dt < tb_task:
                    tb_task, task_threads = dt, th
            for _ in range(2):
                t0 = time.perf_counter(); ot = orc.build(a, parallel=False); tb_ser = min(tb_ser, time.perf_counter() - t0)
            t0 = time.perf_counter(); of = orc.flatten(ot.nodes); tf = min(tf, time.perf_counter() - t0)
            for th in sorted({8, 16, 32, 64, 128, cores} & set(range(1, cores + 1))):   # the box may grant fewer CPUs than it shows
                for _ in range(2):
                    t0 = time.perf_counter()
                    orc.traverse_flat_once(of, a, rr, threads=th)   # the harness loop (testbase.rs:826-836): ONE walk per ray, hits pushed into a per-ray Vec
                    dt = time.perf_counter() - t0
                    if dt < tt_all:
                        tt_all, trav_threads = dt, th
            t0 = time.perf_counter()
            orc.traverse_flat(of, a, rr, threads=trav_threads)      # the CSR form the parity leg uses: count pass + fill pass (two walks per ray)
            tt_csr = min(tt_csr, time.perf_counter() - t0)
            t0 = time.perf_counter()
            orc.traverse_flat_once(of, a, rr[:n1], threads=1)
            tt_1 = min(tt_1, time.perf_counter() - t0)
        native = "O3-native" in builds_timed
        tbuild = min(tb_par, tb_ser)
        cpu_total = tbuild + tf + tt_all * (wl.R / ns)
        out["cpu_baseline"] = {
            "value": round(wl.R / cpu_total / 1e6, 4), "unit": "Mrays/s", "cores": max(trav_threads, par_threads), "host_cpus_visible": cores,
            "kind": "port",
            "sample": f"oracle = C restatement of the reference, NOT the Rust crate (no cargo here); gcc -ffp-contract=off + OpenMP, best per phase of "
                      f"{' and '.join(builds_timed)}{'' if native else ' (the native rebuild failed)'}; full {wl.n_tri}-triangle build "
                      f"{tb_par * 1e3:.1f} ms on {par_threads} threads (scalable schedule: big nodes split by the whole team, then a parallel for over the subtrees — "
                      f"byte-equal to the serial build; the literal OpenMP-task restatement of rayon_executor, bvh_impl.rs:527-543, takes {tb_task * 1e3:.1f} ms on its best "
                      f"{task_threads} threads, the serial build {tb_ser * 1e3:.1f} ms; README.md:155 quotes 8.9 ms for the crate's rayon build on a 12-core 3900X) "
                      f"+ serial flatten {tf * 1e3:.1f} ms + traversal of {ns} of the {wl.R} rays as the reference's harness does it "
                      f"(testbase.rs:826-836: one walk per ray, hits pushed into a per-ray growable list), rays-parallel on {trav_threads} threads (best team size: "
                      f"{tt_all * 1e3:.1f} ms; the two-walk CSR form of the parity leg: {tt_csr * 1e3:.1f} ms), scaled to {wl.R} rays; single-thread "
                      f"traversal {tt_1 / n1 * 1e9:.0f} ns/ray (README.md:175 quotes 866 ns/ray for the Rust crate on a Ryzen 9 3900X)",
            "build_ms": round(tbuild * 1e3, 2), "build_threads": par_threads if tb_par <= tb_ser else 1, "build_schedule": "team-split top + parallel for over subtrees",
            "build_ms_task_recursion": round(tb_task * 1e3, 2), "build_ms_serial": round(tb_ser * 1e3, 2), "flatten_ms": round(tf * 1e3, 2),
            "traverse_ms_all_cores": round(tt_all * (wl.R / ns) * 1e3, 2), "traverse_csr_two_pass_ms": round(tt_csr * (wl.R / ns) * 1e3, 2),
            "traverse_ns_per_ray_1thread": round(tt_1 / n1 * 1e9, 1), "native_build": bool(native),
        }
        out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 2)

    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if env["comm"] is not None:
        env["comm"].close()
    if n_gpus > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
