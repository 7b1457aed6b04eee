"""bench.py's `cpu_baseline` leg as a process of its own — TEST INFRASTRUCTURE, like the rest of oracle/ (never imported by bvh_amd/).

  python -m oracle.baseline_leg <aabbs.npy> <rays.npy> <total rays of the step> [reps]

Times the oracle (the C restatement of the reference: kind "port", NOT the Rust crate) on this box's host cores on the step bench.py's
`value` measures: Bvh::build_par (bvh_impl.rs:527-543; the scalable schedule, byte-equal to the serial recursion) + flatten
(flat_bvh.rs:60-143) + one FlatBvh::traverse per ray into a growable list (the harness loop, testbase.rs:826-836) on the sample of rays in
rays.npy, scaled to the step's ray count.  A process of its own so that the OpenMP runtime starts with pinned threads
(OMP_PROC_BIND=close, OMP_PLACES=cores — bench.py sets them in this process's environment; torch's own OpenMP runtime never loads
here) and so that its figures do not depend on what the parent has mapped.  Every phase is repeated `reps` times at its best team size;
the line carries the figure from the per-phase MINIMA (`value`) and from the per-phase MEDIANS (`value_median`), and the 1-minute load
average of the box before and after: the GPU boxes' host CPUs are shared, and that is what moves this number between runs.
Prints one JSON object."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np


def _timed(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return ts


def main(argv):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import orc
    a = np.load(argv[0])
    rays = np.load(argv[1])
    total_rays = int(argv[2])
    reps = int(argv[3]) if len(argv) > 3 else 5
    load0 = os.getloadavg()[0]
    native = orc.use_native()      # -O3 -march=native built on THIS box; the portable -O2 build if there is no compiler
    cores = orc.max_threads()
    ns = len(rays)
    orc.build(a)                   # warm the allocator and the page cache
    teams = lambda *c: sorted({*c, cores} & set(range(1, cores + 1)))

    # Bvh::build_par: pick the team size on one pass, then `reps` timed builds at it
    best = min(((min(_timed(lambda: orc.build(a, threads=th, schedule="fast"), 2)), th) for th in teams(8, 16, 32, 64, 96, 128)))
    build_threads = best[1]
    tb = _timed(lambda: orc.build(a, threads=build_threads, schedule="fast"), reps)
    tb_task, task_threads = min(((min(_timed(lambda: orc.build(a, threads=th), 1)), th) for th in teams(4, 8, 16) if th <= 16))
    tb_ser = min(_timed(lambda: orc.build(a, parallel=False), 2))
    ot = orc.build(a, parallel=False)
    tf = _timed(lambda: orc.flatten(ot.nodes), reps)
    of = orc.flatten(ot.nodes)
    best = min(((min(_timed(lambda: orc.traverse_flat_once(of, a, rays, threads=th), 2)), th) for th in teams(8, 16, 32, 64, 128)))
    trav_threads = best[1]
    tt = _timed(lambda: orc.traverse_flat_once(of, a, rays, threads=trav_threads), reps)
    n1 = max(ns // 16, 1000)
    tt1 = min(_timed(lambda: orc.traverse_flat_once(of, a, rays[:n1], threads=1), 2))
    scale = total_rays / ns

    def whole(pick):
        return min(pick(tb), tb_ser) + pick(tf) + pick(tt) * scale
    med = lambda v: float(np.median(v))
    out = {
        "value": round(total_rays / whole(min) / 1e6, 4), "value_median": round(total_rays / whole(med) / 1e6, 4), "unit": "Mrays/s",
        "cores": max(trav_threads, build_threads), "host_cpus_visible": cores, "kind": "port", "reps": reps,
        "sample": f"oracle (C port, {'-O3 -march=native' if native else '-O2 portable'}), {'pinned' if os.environ.get('OMP_PROC_BIND') else 'free'} threads: full {len(a)}-shape build + flatten + "
                  f"{ns} of {total_rays} rays walked once each, scaled; min / median of {reps} per phase",
        "build_ms": round(min(min(tb), tb_ser) * 1e3, 3), "build_ms_median": round(med(tb) * 1e3, 3), "build_threads": build_threads,
        "build_ms_task_recursion": round(tb_task * 1e3, 2), "task_threads": task_threads, "build_ms_serial": round(tb_ser * 1e3, 2),
        "flatten_ms": round(min(tf) * 1e3, 3), "traverse_ms_all_cores": round(min(tt) * scale * 1e3, 3),
        "traverse_ms_median": round(med(tt) * scale * 1e3, 3), "traverse_threads": trav_threads,
        "traverse_ns_per_ray_1thread": round(tt1 / n1 * 1e9, 1), "sample_rays": ns, "native_build": bool(native),
        "oracle_library": os.path.basename(orc.library_path()), "omp_proc_bind": os.environ.get("OMP_PROC_BIND"),
        "host_load_1m": [round(load0, 2), round(os.getloadavg()[0], 2)],
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1:])
