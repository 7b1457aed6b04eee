"""bench_sections.py — everything bench.py reports BESIDE the timed step: workloads (scene + ray stream of each BASELINE config, in HBM),
per-phase HIP-event times and the rooflines, the in-process parity legs against the oracle, `step_excludes`, `pipelined`, the
extra configs, the CPU baseline and the N > 1 plans with their watchdog.  bench.py holds the argument parsing, the timed step and the
compact JSON line; what the fields mean is written down in DESIGN.md §7 ("Reading the bench line"), not on the line.

Nothing here is inside a timed region of `value` (bench.run_workload's `timed`).  `oracle` is imported by the parity legs and the
cpu_baseline leg only — as the checker and the reported baseline, never as the thing measured."""
from __future__ import annotations

import contextlib
import glob
import json
import os
import subprocess
import sys
import tempfile
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
N_CU, N_SIMD, CLK = 256, 1024, 2.4e9
VALU_PEAK = N_SIMD * CLK / 2  # wave64 VALU instructions per second: one per 2 cycles per SIMD-32
LDS_PEAK = N_CU * CLK         # LDS-array cycles per second
HITS_CONFIG3_N1 = 457_389_170  # configs[3] whole (100 M rays, stand-in detail 16) as one GPU produced it, parity on every ray (BENCH_r03)


# ----------------------------------------------------------------------------------------------------------------------------------
# N > 1 plumbing that must survive a hung collective
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
    """What the watchdog makes of the (detailed) line measured so far when a section with a collective did not come back (a
    silent fall-back must be impossible to misread).  `pending`: the exchange plan in flight — {"res": the result dict whose
    scene_dist_plans the line shows, "plan", "stage", "workload"} — or None (the hang was elsewhere).  The plan that timed out is NAMED in
    scene_dist_plans with "timed_out": true, never just absent; `rccl` says how far the communicator got.  Returns the JSON text (None
    if the line could not be serialised: the main thread may be publishing into it while this runs — retried)."""
    if pending is not None:
        plans = pending["res"].setdefault("scene_dist_plans", {})
        plans[pending["plan"]] = {"timed_out": True, "after_s": after, "stage": pending["stage"], "workload": pending["workload"]}
        if line.get("scene_dist_plans") is None and line.get("workload_name") == pending["workload"]:
            line["scene_dist_plans"] = plans
    line["collective_watchdog"] = (f"{what} did not finish within {after:.0f} s: this line is what had been measured until then (the "
                                   "replicate plan has no data-path collective); scene_dist_plans names the plan that timed out")
    line["rccl"] = rccl
    for _ in range(20):
        try:
            return json.dumps(line)
        except RuntimeError:      # "dictionary changed size during iteration"
            time.sleep(0.01)
    return None


# ----------------------------------------------------------------------------------------------------------------------------------
class Workload:
    """scene + ray stream of one BASELINE config, resident in HBM; also what the CPU checker needs to redo it"""

    def __init__(self, name, args, dtype_name, rank, n_gpus, dev, ctx, scaling=None, rays=None, harness=None):
        import torch
        from bvh_amd import dist as bdist, scene, testbase as tb
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
            self.config_id = ((1 if dtype_name == "f32" else 4) if name == "cubes120k" else None)
            per = rays or (1_000_000 if name == "cubes120k" else 10_000_000)
            self.scaling = scaling or "weak"
            self.scene = f"create_n_cubes({n_cubes})"
        else:
            self.tris_np, self.aabbs_np, self.bounds = scene.parse_obj(scene.make_atrium_obj(args.standin_detail))
            self.scene = "atrium stand-in for the absent media/sponza.obj"
            if name == "standin-primary":
                # primary rays: BVHGPU_TRAVERSE_COHERENT (how the walk hands its hits over)
                self.config_id, per, self.coherent = 2, rays or 10_000_000, True
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
        self.tris = None
        if harness:
            self.tris = torch.from_numpy(np.ascontiguousarray(self.tris_np, dtype=self.np_dtype).reshape(-1, 9)).to(dev)
        self.rays = self.regen()

    def regen(self):
        """the batch's rays written into its HBM buffer by the device generators, on the context's stream, no host wait: Ray::new per
        ray (ray_impl.rs:70-80) behind create_ray (testbase.rs:687-691) or the primary-ray camera"""
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
        """one short sentence for config.workload (the long form is DESIGN.md §7)"""
        cfg = f"configs[{self.config_id}]" if self.config_id is not None else "beyond BASELINE"
        kind = "primary rays 4000x2500" if self.coherent else "create_ray rays"
        step = "build_par+flatten+traverse" if not self.harness else f"intersect_bh whole ({self.harness}) behind a rebuild"
        per = "total" if self.scaling == "strong" else "per GPU"
        return f"{cfg}: {self.scene}, {self.n_tri} triangles {self.dtype_name}/3D, {self.total_rays} {kind} {per}; step = {step}"


# ----------------------------------------------------------------------------------------------------------------------------------
# rooflines
def newest_bound(kernel_prefix, workload="cubes120k", dtype="f32", rays=1_000_000):
    """profiles/*_bound.json of the newest profile round that holds PMC counters for this kernel ON THIS WORKLOAD (the counters
    of a walk depend on the scene and the ray stream; files written before round 3 carry no workload tag and are configs[1] f32)"""
    def key(f):      # newest round / version tag first (r6_v1 > r5_v7 > r2_v8)
        import re
        m = re.match(r"r(\d+)_v(\d+)", os.path.basename(f))
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bound.json")), key=key, reverse=True)
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


def walk_roofline(wl, phases, stats, kern_name, guide_ran):
    """The dominant kernel (the walk) against its BINDING resource: PMC counters per launch from profiles/*_bound.json (looked up under
    the kernel name the library reports for the timed batch shape, bvhgpu_hits_walk_kernel) over the live HIP-event kernel time.  Beside it
    SURVEY §8d's own figure: algorithmic bytes (Ray in + V x FlatNode + V_leaf x shape AABB + CSR out) / kernel time / 8 TB/s."""
    R = wl.R
    V, VL, H = stats["visited"], stats["leaf_visits"], stats["hits"]
    elem = 4 if wl.dtype_name == "f32" else 8
    flat_sz = 36 if wl.dtype_name == "f32" else 64
    algo_bytes = R * wl.ray_size + V * flat_sz + VL * 6 * elem + 4 * (H + R)
    if wl.harness == "triangles":   # + the triangle stage: 9 vertices read, Intersection{distance,u,v} written per candidate
        algo_bytes += H * (9 + 3) * elem
    elif wl.harness == "closest":   # + 9 vertices read per candidate; one Intersection + shape per ray instead of the CSR
        algo_bytes += H * 9 * elem + R * (3 * elem + 4) - 4 * (H + R)
    kern_s = phases["traverse_kernel_ms"] * 1e-3
    pmc, src = newest_bound(kern_name, wl.tag, wl.dtype_name, R)
    roof = {
        "kernel": kern_name, "kernel_ms": round(phases["traverse_kernel_ms"], 4),
        "algorithmic_bytes_per_launch": int(algo_bytes), "algorithmic_gbs": round(algo_bytes / kern_s / 1e9, 1),
        "algorithmic_frac": round(algo_bytes / kern_s / 1e9 / HBM_PEAK_GBS, 4),
        "slab_tests_per_s": round(V / kern_s, 1), "visited": int(V), "leaf_visits": int(VL), "hits": int(H),
    }
    if wl.dtype_name == "f64":
        roof["f64_walk"] = "guide" if guide_ran else "pure"
    if pmc is None:
        roof.update({"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                     "source": None})
        return roof
    fr = bound_fractions(pmc, kern_s)
    bound = max(fr, key=fr.get)
    lds_cycles = pmc.get("SQ_INSTS_LDS", 0) * 4 + pmc.get("SQ_LDS_BANK_CONFLICT", 0)
    peak, unit, ach = {"hbm": (HBM_PEAK_GBS, "GB/s", pmc.get("hbm_bytes", 0) / kern_s / 1e9),
                       "valu": (VALU_PEAK / 1e9, "G wave-instr/s", pmc.get("SQ_INSTS_VALU", 0) / kern_s / 1e9),
                       "lds": (LDS_PEAK / 1e9, "G LDS-cycles/s", lds_cycles / kern_s / 1e9)}[bound]
    roof.update({
        "bound": bound, "achieved": round(ach, 2), "peak": round(peak, 1), "unit": unit, "frac": round(fr[bound], 4),
        "traffic": pmc.get("hbm_bytes"), "hbm_frac": round(fr.get("hbm", 0), 4), "valu_frac": round(fr.get("valu", 0), 4),
        "lds_frac": round(fr.get("lds", 0), 4), "wait_frac": pmc.get("wait_frac"), "profile_kernel_us": pmc.get("avg_us"),
        "traffic_over_algorithmic": round(pmc["hbm_bytes"] / algo_bytes, 4) if pmc.get("hbm_bytes") else None,
        "profile_rays_per_launch": pmc.get("scaled_from_rays", R), "source": src,
    })
    return roof


def build_roofline(wl, phases, levels):
    """builder chain against the HBM roofline (SURVEY §8d build bytes); `levels` = mean leaf depth from the oracle's tree when the
    parity leg ran (sum over the tree levels of the shapes still being partitioned / N), else log2 N"""
    n = wl.n_tri
    f32 = wl.dtype_name == "f32"
    lv = levels if levels else float(np.log2(max(n, 2)))
    bbytes = (32 if f32 else 56) * lv * n + (2 * n - 1) * (64 if f32 else 112)
    fbytes = (2 * n - 1) * (64 if f32 else 112) + (3 * n - 2) * (36 if f32 else 64)
    ms = phases["build_ms"] + phases["flatten_ms"]
    gbs = (bbytes + fbytes) / (ms * 1e-3) / 1e9
    return {"kernels": "k_prep, k_level x (levels + 1), k_mid, k_small, k_flatten", "bound": "hbm",
            "algorithmic_bytes": int(bbytes + fbytes), "ms": round(ms, 4), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "levels_priced": round(lv, 2), "levels_source": "oracle" if levels else "log2 N"}


def phases_and_roofline(wl, env, bvh, tree, builder, mode_flags, steps, detailed):
    """untimed extra steps behind the timed region: per-phase HIP-event times, the walk's name as the library reports it, the exact visit
    counters of the reference's loop (binary STATS walk) and the rooflines that follow from them"""
    import torch
    from bvh_amd._lib import WALK_F64_GUIDE
    ctx, dev = env["ctx"], env["dev"]
    ctx.enable_timing(True)
    ph = dict(build_ms=[], flatten_ms=[], traverse_kernel_ms=[], traverse_total_ms=[])
    stats_walk, walk_kernel = 0, ""
    for _ in range(max(5, min(steps, 20))):
        if builder:
            bvh.rebuild(wl.aabbs)
            bvh.flatten_in_place()
        hh = tree.traverse_async(wl.rays, flags=mode_flags)   # (the synchronous entry points cover index batches only without a fetch)
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
    stats = tree.traverse_batch(wl.rays, stats=True, fetch=False, coherent=wl.coherent)[3]
    roof = walk_roofline(wl, phases, stats, walk_kernel, bool(stats_walk & WALK_F64_GUIDE))
    out = {"phases_ms": {k: round(v, 4) for k, v in phases.items()}, "roofline": roof}
    if detailed and builder:
        out["roofline_build"] = build_roofline(wl, phases, None)
        out["build_levels"] = bvh.build_levels
    return out, phases, stats


# ----------------------------------------------------------------------------------------------------------------------------------
# parity legs (rank 0, outside every timed region): the GPU result against the CPU oracle, in-process
def check_parity(wl, env, orc, n_check, chunk=1_000_000):
    """The GPU result of this rank's WHOLE batch (CSR of the default walk fetched once; visit counters of the binary walk) against
    the oracle on the first n_check rays (default: all of them).  The oracle works through the rays in chunks of `chunk` (memory),
    each chunk diffed against its slice of the one GPU result."""
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
    csr_equal, V, VL, H, t_or = True, 0, 0, 0, 0.0
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        rays_o = wl.oracle_rays(orc, wl.first + c0, m)
        t0 = time.perf_counter()
        ooff, oidx, _, ost = orc.traverse_flat(oflat, a, rays_o, threads=orc.max_threads())
        t_or += time.perf_counter() - t0
        base = int(off[c0])
        csr_equal = csr_equal and bool(np.array_equal(off[c0:c0 + m + 1] - np.uint32(base), ooff)
                                       and np.array_equal(idx[base:int(off[c0 + m])], oidx))
        V += ost["visited"]; VL += ost["leaf_visits"]; H += ost["hits"]
    cnt_equal = bool(st["visited"] == V and st["leaf_visits"] == VL and st["hits"] == H and len(idx) == H)
    nodes_equal = None
    if last["builder"] and last["bvh"] is not None:
        nodes_equal = bool(last["bvh"].nodes.tobytes() == ot.nodes.tobytes())
    return {"checked_rays": int(n), "rays_this_rank": int(wl.R), "equal": bool(csr_equal and cnt_equal and nodes_equal is not False),
            "csr_offsets_and_indices_equal": csr_equal, "visit_counters_equal": cnt_equal, "bvh_nodes_equal": nodes_equal,
            "hits": int(H), "against": "oracle", "oracle_traverse_s": round(t_or, 4)}


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
    equal, H = True, 0
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
        H += ost["hits"]
    nodes_equal = bool(last["bvh"].nodes.tobytes() == ot.nodes.tobytes()) if last["builder"] and last["bvh"] is not None else None
    out = {"checked_rays": int(n), "rays_this_rank": int(wl.R), "equal": bool(equal and nodes_equal is not False), "candidates": int(H),
           "what": "closest (distance,u,v,shape) per ray" if wl.harness == "closest" else "CSR + Intersection per candidate",
           "bvh_nodes_equal": nodes_equal, "against": "oracle"}
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
        # (which oracle library ran — the portable -O2 build unless a cpu_baseline leg of this process switched to the native one)
        cpu = {"value": round(wl.R / total / 1e6, 4), "unit": "Mrays/s", "cores": best_th, "kind": "port", "sample_rays": int(ns),
               "oracle_library": os.path.basename(orc.library_path()), "loop_ms_scaled": round(best * (wl.R / ns) * 1e3, 2),
               "build_ms": round(t_build * 1e3, 2), "flatten_ms": round(t_flat * 1e3, 2)}
    return out, cpu


# ----------------------------------------------------------------------------------------------------------------------------------
def measure_excluded(wl, args, env, ms_step):
    """What the timed step of `value` does not contain, each as the SAME step with that thing put inside, K steps timed the same way
    (device sync on both sides):
      with_ray_gen      Ray::new for every ray of the batch (ray_impl.rs:70-80 via create_ray, testbase.rs:687-691: the reference's
                        bench iteration starts with it) generated on the device inside the step — k_gen_rays on the step's stream
      lazy / all / beside ...   the step of `value` writes the reference-layout FlatNode array in every flatten (flat_bvh.rs:60-143;
                        --flat-array eager; the engine's own folded binary array follows on first use); the other ways: on first use
                        (lazy: NOT in the step), both arrays in the flatten kernel (all_arrays_eager), both by a second pass beside the walk
      host_io           shape AABBs and rays start in HOST memory, the CSR ends in host memory: what GpuBvh::build + traverse_batch
                        of the Rust shim costs a caller whose data lives in Vecs (rust/bvh-mi355x/src/lib.rs) — upload, step, download"""
    import torch
    from bvh_amd import Bvh, RayBatch
    from bvh_amd._lib import TRAVERSE_RAYS_READY, TUNE_FLATTEN_LAZY
    dev, ctx = env["dev"], env["ctx"]
    K = max(args.steps, 100)
    bvh = Bvh.from_aabbs(wl.aabbs, ctx)
    bvh.flatten_in_place()

    def timed(fn, k, blocks=3):      # median block (bench.run_workload says why)
        for _ in range(5):
            fn()
        ts = []
        for _ in range(blocks):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(k):
                fn()
            torch.cuda.synchronize(dev)
            ts.append((time.perf_counter() - t0) / k * 1e3)
        return sorted(ts)[len(ts) // 2]

    def entry(ms):
        return {"value": round(wl.R / (ms * 1e-3) / 1e6, 3), "unit": "Mrays/s", "ms_per_step": round(ms, 4),
                "delta_ms_vs_value": round(ms - ms_step, 4)}

    res = {"steps": K}

    def step_gen():
        bvh.rebuild_async(wl.aabbs)
        RayBatch.generate(wl.first, wl.R, wl.bounds, wl.rays_buf, wl.np_dtype, ctx)   # the same buffer, rewritten every step
        return bvh.traverse_async(wl.rays, flags=0).wait()
    res["with_ray_gen"] = entry(timed(step_gen, K))

    prev = ctx.get_tuning(TUNE_FLATTEN_LAZY)

    def step_index():
        bvh.rebuild_async(wl.aabbs)
        return bvh.traverse_async(wl.rays, flags=TRAVERSE_RAYS_READY).wait()
    for mode, key in ((1, "lazy_flat_array"), (3, "eager_flat_array"), (0, "all_arrays_eager"), (2, "beside_flat_array")):
        if mode == prev:
            continue
        ctx.set_tuning(TUNE_FLATTEN_LAZY, mode)
        try:
            res[key] = entry(timed(step_index, K))
        finally:
            ctx.set_tuning(TUNE_FLATTEN_LAZY, prev)

    res["host_io"] = host_io_steps(wl, env, bvh, timed, entry, K)
    bvh.close()
    return res


def host_io_steps(wl, env, bvh, timed, entry, K):
    """host-resident callers (the drop-in boundary as a Rust caller meets it): inputs in host memory, CSR back in host memory, inside
    every step.  "pageable" = the synchronous entry points on plain host arrays (bvhgpu_rebuild_flat + bvhgpu_traverse +
    bvhgpu_hits_fetch: Ray structs, 36 B per ray); "pinned" = ABI 7's host batch on pinned buffers (bvhgpu_host_alloc): origin + direction
    only (24 B per ray, one array, Ray::new on the device), the ray upload in chunks beside the build, the walk of a chunk beside the upload
    of the next, offsets / indices written into the caller's arrays by the device — as two calls and as one.  Median of 5 blocks of
    steps (a host thread of a shared box is descheduled for tens of ms now and then: one stall in one block must not become the figure)."""
    import torch
    from bvh_amd import HostStep, RayBatch
    from bvh_amd._lib import RAY_F32, RAY_F64
    a_host = np.ascontiguousarray(wl.aabbs_np.astype(wl.np_dtype))
    rays_host = torch.empty(wl.R * wl.ray_size, dtype=torch.uint8)
    rays_host.copy_(wl.rays_buf[:wl.R * wl.ray_size])
    rays_np = rays_host.numpy().view(RAY_F32 if wl.dtype_name == "f32" else RAY_F64)
    rb_host = RayBatch(wl.R, wl.np_dtype, host=rays_np)
    kh = max(10, min(K, 30))
    out = {}

    def median_of_blocks(fn):
        v = sorted(timed(fn, kh, blocks=1) for _ in range(5))
        return v[2], [round(x, 4) for x in v]

    def step_pageable():
        bvh.rebuild(a_host, flatten=True)
        off, idx, _, _ = bvh.traverse_batch(rb_host, fetch=True)
        return off, idx
    off, idx = step_pageable()
    nbytes = {"aabbs_up": int(a_host.nbytes), "rays_up": int(wl.R * wl.ray_size), "csr_down": int(off.nbytes + idx.nbytes)}
    ms, blocks = median_of_blocks(step_pageable)
    e = entry(ms)
    e.update(steps=kh, blocks_ms=blocks, bytes_per_step=nbytes, pcie_gbs=round(sum(nbytes.values()) / (ms * 1e-3) / 1e9, 2))
    out["pageable"] = e
    # origins and un-normalised directions whose Ray::new is the batch's rays: the stream's raw points (create_ray, testbase.rs:687-691)
    from bvh_amd import testbase as tb
    k = np.arange(wl.first, wl.first + wl.R, dtype=np.uint64)
    hs = HostStep(bvh, wl.n_tri, wl.R, wl.np_dtype, od6=True)     # origin and direction side by side: one transfer per chunk
    hs.aabbs[:] = a_host
    if wl.cam is None:
        hs.origins[:] = tb.next_point3_at(2 * k + 1, wl.bounds).astype(wl.np_dtype)
        hs.directions[:] = tb.next_point3_at(2 * k + 2, wl.bounds).astype(wl.np_dtype)
    else:       # (a camera batch: the normalised directions; Ray::new of a unit vector need not give its bits back — not compared below)
        hs.origins[:] = rays_np["o"]
        hs.directions[:] = rays_np["d"]
    for key, fused in (("pinned", False), ("pinned_one_call", True)):
        # two calls = GpuBvh::rebuild_async + traverse_batch_od6 of the Rust shim (bvhgpu_rebuild_flat_async + bvhgpu_traverse_host);
        # one call = bvhgpu_build_traverse_host (shapes and rays through the same upload stream, the build behind an event)
        off2, idx2 = hs.run(fused=fused)
        same = bool(np.array_equal(off, off2) and np.array_equal(idx, idx2)) if wl.cam is None else None
        nb2 = {"aabbs_up": int(a_host.nbytes), "rays_up": int(hs.od.nbytes), "csr_down": int(off2.nbytes + idx2.nbytes)}
        ms2, blocks2 = median_of_blocks(lambda: hs.run(fused=fused))
        e2 = entry(ms2)
        e2.update(steps=kh, blocks_ms=blocks2, bytes_per_step=nb2, pcie_gbs=round(sum(nb2.values()) / (ms2 * 1e-3) / 1e9, 2),
                  csr_equal_to_pageable_path=same)
        out[key] = e2
    hs.close()
    best = max(out.values(), key=lambda q: q["value"])
    return dict(best, paths=out)


def pipelined(wl, args, env, local_rank):
    """N = 1, reported beside `value`, never as it: the same K steps kept in flight on S HIP streams by ONE host thread through the
    asynchronous C ABI (bvhgpu_rebuild_flat_async / bvhgpu_traverse_async / bvhgpu_hits_wait only when a lane's result object is needed
    again) — the build of one step overlaps the walk of another; throughput of independent steps, not the latency of one."""
    import torch
    from bvh_amd import Bvh, Context
    from bvh_amd._lib import TRAVERSE_RAYS_READY as RAYS_READY
    from bvh_amd.api import _Hits
    S, dev = args.pipeline_streams, env["dev"]
    lanes = []
    for _ in range(S):
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
    for c, tr, h, _ in lanes:
        h.close(); tr.close(); c.close()
    return {"streams": S, "host_threads": 1, "steps": K, "value": round(K * wl.R / dtp / 1e6, 3), "unit": "Mrays/s",
            "ms_per_step": round(dtp * 1e3 / K, 4), "hits_every_step_equal": bool(len(set(hits_p)) == 1 and len(hits_p) == K),
            "hits": hits_p[0] if hits_p else None}


def cpu_baseline(wl, args):
    """the oracle (C port of the reference algorithm, kind "port") on this box's host cores, in a process of its own:
    oracle/baseline_leg.py, run twice — with the OpenMP threads pinned (OMP_PROC_BIND=close, OMP_PLACES=cores) and with the scheduler free
    to move them.  The GPU boxes' host CPUs are shared: on a loaded box pinned threads queue behind the other tenants' (r6_a: 94 ms of
    traversal pinned at load 56, against 9 - 30 ms free in round 5), on a quiet one pinning wins — the faster leg is `value`, both are in
    the detail file, and the load average is on the line.  Min and median of 5 per phase."""
    from oracle import orc
    ns = min(args.cpu_sample_rays, wl.R)
    legs = {}
    with tempfile.TemporaryDirectory(prefix="bvh_cpu_leg_") as d:
        np.save(os.path.join(d, "a.npy"), wl.aabbs_np.astype(wl.np_dtype))
        np.save(os.path.join(d, "r.npy"), wl.oracle_rays(orc, wl.first, ns))
        cmd = [sys.executable, "-m", "oracle.baseline_leg", os.path.join(d, "a.npy"), os.path.join(d, "r.npy"), str(wl.R), "5"]
        base = {k: v for k, v in os.environ.items() if k not in ("OMP_PROC_BIND", "OMP_PLACES")}
        for name, env in (("free", base), ("pinned", dict(base, OMP_PROC_BIND="close", OMP_PLACES="cores"))):
            p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
            if p.returncode == 0:
                legs[name] = json.loads(p.stdout.strip().splitlines()[-1])
            else:
                legs[name] = {"error": p.stderr[-400:]}
    good = {k: v for k, v in legs.items() if "error" not in v}
    if not good:
        raise RuntimeError("oracle.baseline_leg failed: " + json.dumps(legs)[:600])
    best = max(good, key=lambda k: good[k]["value"])
    keep = ("value", "value_median", "host_load_1m", "traverse_ms_all_cores", "build_ms")
    return dict(good[best], threads=best, legs={k: ({q: v[q] for q in keep} if "error" not in v else v) for k, v in legs.items()})


# ----------------------------------------------------------------------------------------------------------------------------------
class Run:
    """One bench.py process: the JSON line as far as it has been measured (what the watchdog prints if an exchange section hangs), the
    RCCL communicator of the C ABI, and `measure` — one workload under the plan(s) this run reports."""

    def __init__(self, args, env, launch_obj, json_fd, run_workload, emit):
        self.args, self.env, self.launch_obj, self.json_fd = args, env, launch_obj, json_fd
        self.run_workload, self.emit = run_workload, emit
        self.line = {}
        self.xstate = {"comm_err": None, "tried_comm": False, "pending": None}

    def emit_and_exit(self, what, seconds=None):
        """runs on the watchdog's helper thread: print the line as far as it has been measured and end the process — whatever happens
        on the way (os._exit sits in a `finally`: an exception here must not bring back the hang the watchdog exists to prevent)"""
        code = 3
        try:
            env = self.env
            after = seconds if seconds is not None else self.args.collective_timeout
            try:
                none = {"nranks": None, "formed": False, "error": self.xstate["comm_err"]}
                rccl = env["comm"].info() if env["comm"] is not None else none
            except Exception as e:
                rccl = {"nranks": None, "formed": env["comm"] is not None, "error": repr(e)}
            text = timed_out_line(self.line, self.xstate.get("pending"), what, after, rccl)
            if text is not None and self.line.get("value") is not None:
                if env["rank"] == 0:
                    self.emit(json.loads(text))
                code = 0
        finally:
            os._exit(code)

    def make_comm(self):
        """the RCCL communicator of the C ABI (torch.distributed only carries the 128-byte id) — made AFTER the replicate plan has been
        measured, inside the watchdog: it is the first thing in the run that has never been exercised with more than one rank"""
        import torch
        import torch.distributed as dist
        from bvh_amd import dist as bdist
        env, xs = self.env, self.xstate
        if xs["tried_comm"] or self.args.backend != "nccl":
            return env["comm"]
        xs["tried_comm"] = True
        comm = None
        try:
            comm = bdist.Communicator.from_torch_distributed(env["ctx"], env["dev"])
        except Exception as e:   # keep the run alive on the torch transport, and say so
            xs["comm_err"] = repr(e)
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=env["dev"])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            comm = None
        env["comm"] = comm
        return comm

    def measure(self, w, steps, warmup, detailed, publish=None):
        """one workload → result dict of the plan that is reported.  N = 1 or an explicit --scene-dist: one run.  N > 1 with
        --scene-dist auto: the replicate plan first (every rank builds: no data-path collective, so this result is safe — `publish`
        puts it into the line at once), then the exchange plan under the watchdog; the faster one is reported, both are on the line."""
        args, env, xs = self.args, self.env, self.xstate
        if env["n_gpus"] == 1 or args.scene_dist != "auto":
            if env["n_gpus"] > 1 and args.scene_dist == "bcast":
                with Watchdog(args.collective_timeout, lambda: self.emit_and_exit("forming the RCCL communicator")):
                    self.make_comm()
            return self.run_workload(w, args, env, steps, warmup, detailed)
        res = self.run_workload(w, args, env, steps, warmup, detailed, force_plan="replicate")
        keep = dict(env["last"])
        pick = lambda r: {k: r[k] for k in ("value", "ms_per_step", "phases_ms", "hits_all_ranks") if k in r}
        res["scene_dist_plans"] = {"replicate": pick(res)}
        if publish:
            publish(res)
        guess = "bcast" if args.backend == "nccl" else "bcast-torch"
        xs["pending"] = {"res": res, "plan": guess, "stage": "forming the RCCL communicator", "workload": w.tag}
        with Watchdog(args.collective_timeout, lambda: self.emit_and_exit(f"the exchange plan of {w.name}")):
            if os.environ.get("BVH_BENCH_TEST_HANG_EXCHANGE"):   # tests: a collective that never returns (tests/test_gpu_dist.py)
                time.sleep(10 ** 6)
            comm = self.make_comm()
            xplan = "bcast" if comm is not None else "bcast-torch"
            xs["pending"].update(plan=xplan, stage="the exchange plan's steps (communicator formed)")
            res_x = self.run_workload(w, args, env, steps, warmup, detailed, force_plan=xplan)
        xs["pending"] = None
        probe = {"replicate": res["ms_per_step"], xplan: res_x["ms_per_step"]}
        both = {"replicate": pick(res), xplan: pick(res_x)}
        if res_x["ms_per_step"] < res["ms_per_step"]:
            res = res_x
        else:
            env["last"] = keep
        res["scene_dist_probe_ms_per_step"] = probe
        res["scene_dist_plans"] = both
        return res

    def compose(self, res, wl):
        """the headline part of the detailed line from one workload's result"""
        args, env = self.args, self.env
        n = env["n_gpus"]
        plan_words = {"single": "", "bcast": ", tree RCCL-broadcast from rank 0 every step (bvhgpu_bcast_known)",
                      "bcast-torch": ", scene blob broadcast from rank 0 every step over torch.distributed",
                      "replicate": ", every rank rebuilds the scene (no data-path collective)"}
        out = {
            "metric": "Mrays/s (build+traverse)", "value": res["value"], "unit": "Mrays/s", "n_gpus": n,
            "steps": args.steps, "warmup": args.warmup, "settle_steps": res.get("settle_steps"), "ms_per_step": res["ms_per_step"],
            "regions_ms_per_step": res.get("regions_ms_per_step"),
            "higher_is_better": True, "scaling": wl.scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "workload_name": wl.tag, "harness": wl.harness,
            "config": {"workload": wl.describe(), "triangles": wl.n_tri, "rays_per_gpu": wl.R, "rays_total": wl.total_rays,
                       "scene_dist": res["scene_dist"], "flat_array": res.get("flat_array"),
                       "parallelism": f"rays sharded x{n}" + plan_words[res["scene_dist"]]},
            "phases_ms": res["phases_ms"], "build_levels": res.get("build_levels"), "hits_all_ranks": res["hits_all_ranks"],
            "scene_dist_probe_ms_per_step": res.get("scene_dist_probe_ms_per_step"), "scene_dist_plans": res.get("scene_dist_plans"),
            "roofline": res["roofline"], "roofline_build": res.get("roofline_build"), "launch": self.launch_obj,
            # None: no RCCL communicator in this run (N = 1, --backend gloo, --scene-dist replicate, or rccl_comm_error)
            "rccl": env["comm"].info() if env["comm"] is not None else None,
        }
        if self.xstate["comm_err"]:
            out["rccl_comm_error"] = self.xstate["comm_err"]
        return out

    # ---- the other BASELINE configs, the reference's whole harness loop and a scene beyond the caches, in the same line ----
    def extras(self, args, rank, dev):
        import torch
        env, n_gpus, ctx = self.env, self.env["n_gpus"], self.env["ctx"]
        extras = []
        self.line["extra_configs"] = extras
        E = lambda name, dt="f32", scaling=None, rays=None, harness=None, **kw: dict(name=name, dt=dt, scaling=scaling, rays=rays,
                                                                                     harness=harness, **kw)
        if n_gpus == 1:
            plan = [
                # intersect_bh (testbase.rs:819-837) whole, behind a rebuild: ray generation + build + flatten + walk + triangle stage
                E("cubes120k", harness="closest"), E("cubes120k", harness="triangles"), E("standin-primary", harness="closest"),
                E("standin-primary"), E("standin-incoherent", scaling="weak", rays=12_500_000), E("cubes120k", dt="f64"),
                E("standin-incoherent", scaling="strong", rays=100_000_000),   # configs[3] whole on ONE GPU: N = 1 of the strong curve
                # the regime the north star's HBM language is about: a tree far beyond L2 + MALL (12 M triangles)
                E("cubes12m", rays=10_000_000, parity_rays=1_000_000),
                E("cubes120k", dt="f64", harness="closest"),                   # the harness loop in f64 (closest hit over items, f64 key)
            ]
            only = os.environ.get("BVH_BENCH_EXTRAS")     # developer runs: comma-separated entry numbers of the list above
            if only:
                plan = [plan[int(k)] for k in only.split(",")]
        else:
            # (tests shrink the stream: BVH_BENCH_STRONG_RAYS; the driver's run keeps BASELINE's 100 M)
            plan = [E("standin-incoherent", scaling="strong", rays=int(os.environ.get("BVH_BENCH_STRONG_RAYS", 100_000_000)))]
        # N > 1: the section's barriers and all-reduces are only safe while every rank gets through it — a rank that fails alone (its
        # `except` below skips the collectives) would leave the others waiting for ever, and the headline with them
        guard = (Watchdog(args.extras_timeout, lambda: self.emit_and_exit("the extra_configs section", args.extras_timeout))
                 if n_gpus > 1 else contextlib.nullcontext())
        with guard:
            for e in plan:
                try:
                    extras.append(self.one_extra(e, args, rank, dev, extras))
                    env["last"] = {}
                    torch.cuda.empty_cache()
                except Exception as ex:   # an extra config must never take the headline line down
                    extras.append({"workload": e["name"], "dtype": e["dt"], "harness": e["harness"], "error": repr(ex),
                                   "where": traceback.format_exc(limit=3)[-400:]})
        return extras

    def one_extra(self, e, args, rank, dev, extras):
        from bvh_amd._lib import TUNE_WIDE_F64_GUIDE
        env, n_gpus, ctx = self.env, self.env["n_gpus"], self.env["ctx"]
        name, dt, scaling, nrays = e["name"], e["dt"], e["scaling"], e["rays"]
        w2 = Workload(name, args, dt, rank, n_gpus, dev, ctx, scaling=scaling, rays=nrays, harness=e["harness"])
        shard = name == "standin-incoherent" and n_gpus == 1 and scaling == "weak"
        if shard:   # the shard rank 5 of 8 owns: rays [62.5 M, 75 M) of the stream (tests/test_gpu_scene.py checks the same one)
            w2.first = 62_500_000
            w2.rays = w2.regen()
        provisional = []

        def publish_extra(r):      # N > 1: the replicate result is on the line before the exchange plan is tried
            provisional.append(r)
            extras.append(r)
        r2 = self.measure(w2, args.extra_steps, 3, False, publish=publish_extra)
        for r in provisional:       # (replaced by the finished entry the caller appends)
            if r in extras:
                extras.remove(r)
        if shard:
            r2["first_ray"] = w2.first
        if name == "standin-incoherent" and scaling == "strong" and nrays == 100_000_000 and args.standin_detail == 16:
            # the shards of an N > 1 run must add up to exactly the N = 1 count — under EVERY plan that was measured
            r2["hits_n1_reference"] = HITS_CONFIG3_N1
            r2["hits_match_n1_reference"] = bool(r2["hits_all_ranks"] == HITS_CONFIG3_N1)
            for pl in (r2.get("scene_dist_plans") or {}).values():
                if "hits_all_ranks" in pl:
                    pl["hits_match_n1_reference"] = bool(pl["hits_all_ranks"] == HITS_CONFIG3_N1)
        check = rank == 0 and not args.no_parity
        if check:
            from oracle import orc
            n_par = min(w2.R, args.parity_max_rays, e.get("parity_rays") or w2.R)
            if w2.harness:
                r2["parity"], r2["cpu_harness"] = check_parity_harness(w2, env, orc, n_par)
                if r2["cpu_harness"]:
                    r2["speedup_vs_cpu_harness"] = round(r2["value"] / r2["cpu_harness"]["value"], 2)
            else:
                r2["parity"] = check_parity(w2, env, orc, n_par)
        if dt == "f64" and not w2.harness:
            # the same step with EVERY slab test of the walk in double precision (BASELINE configs[4] names "double-precision slab
            # test"): k_traverse_wide<double, …>, its own timing, roofline (its own counter passes when profiles/ holds them) and parity
            prev = ctx.get_tuning(TUNE_WIDE_F64_GUIDE)
            ctx.set_tuning(TUNE_WIDE_F64_GUIDE, 0)
            try:
                r3 = self.run_workload(w2, args, env, args.extra_steps, 3, detailed=False)
                if check:
                    r3["parity"] = check_parity(w2, env, orc, min(w2.R, args.parity_max_rays))
                keys = ("value", "unit", "ms_per_step", "steps", "phases_ms", "hits_all_ranks", "roofline", "parity")
                r2["pure_f64_walk"] = {k: r3[k] for k in keys if k in r3}
            finally:
                ctx.set_tuning(TUNE_WIDE_F64_GUIDE, prev)
        return r2
