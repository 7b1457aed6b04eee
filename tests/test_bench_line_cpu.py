"""The driver reads ONE stdout line of bench.py; round 5's grew to 35 KB and could not be parsed (BENCH_r05.json: "parsed": null).  These tests
bound the line on every CPU run: a full synthetic detail object — every section, all nine extra configs, the N = 8 form with both plans on
the headline and on configs[3] strong — must render below bench.LINE_BUDGET (8 KB) and keep the contract's keys; an object that would not
fit loses its rows, never the contract."""
import glob
import importlib.util
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod_line", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def _roof(kernel="bvhgpu::k_traverse_wide<float, 0, 2, 1024, 8, 0>"):
    return {"kernel": kernel, "kernel_ms": 0.1214, "algorithmic_bytes_per_launch": 2674956488, "algorithmic_gbs": 22034.2, "algorithmic_frac": 2.754,
            "slab_tests_per_s": 602787524757.0, "visited": 73173216, "leaf_visits": 10000, "hits": 10000, "bound": "valu", "achieved": 496.41,
            "peak": 1228.8, "unit": "G wave-instr/s", "frac": 0.404, "traffic": 117370481.43972835, "hbm_frac": 0.1208, "valu_frac": 0.404,
            "lds_frac": 0.3518, "wait_frac": 0.4829, "profile_kernel_us": 119.783, "traffic_over_algorithmic": 0.0439,
            "profile_rays_per_launch": 1000000, "source": "profiles/r6_v1_c1_bound.json"}


def _phases(harness=False):
    p = {"build_ms": 0.1867, "flatten_ms": 0.0122, "traverse_kernel_ms": 0.1214, "traverse_total_ms": 0.1331}
    if harness:
        p["ray_gen_ms"] = 0.0136
    return p


def _extra(workload, harness=None, dtype="f32", config=1, scaling="weak", rays=1_000_000, plans=None):
    e = {"workload": workload + (f"+{harness}" if harness else ""), "harness": harness, "config": config, "dtype": dtype, "value": 2947.32, "unit": "Mrays/s",
         "ms_per_step": 0.3393, "steps": 20, "warmup": 3, "settle_steps": 300, "scaling": scaling, "triangles": 165320, "rays_this_rank": rays,
         "rays_total": rays, "scene_dist": "single", "flat_array": "eager", "hits_all_ranks": 457389170, "visited_per_ray": 83.5,
         "scene_dist_probe_ms_per_step": None, "phases_ms": _phases(bool(harness)), "roofline": _roof(),
         "parity": {"checked_rays": rays, "rays_this_rank": rays, "equal": True, "csr_offsets_and_indices_equal": True, "visit_counters_equal": True,
                    "bvh_nodes_equal": True, "hits": 457389170, "against": "oracle", "oracle_traverse_s": 12.5}}
    if harness:
        e["cpu_harness"] = {"value": 20.6229, "unit": "Mrays/s", "cores": 64, "kind": "port", "sample_rays": 1000000, "oracle_library": "liboracle.so",
                            "loop_ms_scaled": 45.1, "build_ms": 3.2, "flatten_ms": 1.3}
        e["speedup_vs_cpu_harness"] = 142.91
    if plans:
        e["scene_dist_plans"] = plans
        e["hits_n1_reference"], e["hits_match_n1_reference"] = 457389170, True
    return e


def _detail(n_gpus=1):
    plan = {"value": 23456.789, "ms_per_step": 0.3411, "phases_ms": _phases(), "hits_all_ranks": 80000, "hits_match_n1_reference": True}
    plans = {"replicate": dict(plan), "bcast": dict(plan)} if n_gpus > 1 else None
    d = {
        "metric": "Mrays/s (build+traverse)", "value": 3089.913, "unit": "Mrays/s", "n_gpus": n_gpus, "steps": 20, "warmup": 5, "settle_steps": 300,
        "ms_per_step": 0.3236, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "workload_name": "cubes120k", "harness": None,
        "config": {"workload": "configs[1]: create_n_cubes(10000), 120000 triangles f32/3D, 1000000 create_ray rays per GPU; step = build_par+flatten+traverse",
                   "triangles": 120000, "rays_per_gpu": 1000000, "rays_total": 1000000 * n_gpus, "scene_dist": "bcast" if n_gpus > 1 else "single",
                   "flat_array": "eager", "parallelism": "rays sharded x%d, tree RCCL-broadcast from rank 0 every step (bvhgpu_bcast_known)" % n_gpus},
        "phases_ms": _phases(), "build_levels": 8, "hits_all_ranks": 10000 * n_gpus, "scene_dist_probe_ms_per_step": {"replicate": 0.34, "bcast": 0.36},
        "scene_dist_plans": plans, "roofline": _roof(),
        "roofline_build": {"kernels": "k_prep, k_level x (levels + 1), k_mid, k_small, k_flatten", "bound": "hbm", "algorithmic_bytes": 112295352, "ms": 0.1989,
                           "achieved": 564.6, "peak": 8000.0, "unit": "GB/s", "frac": 0.0706, "levels_priced": 17.9, "levels_source": "oracle"},
        "launch": {"world_size": n_gpus, "ranks_seen": n_gpus, "self_launched": False, "backend": "nccl" if n_gpus > 1 else None,
                   "devices": ["mi355x-node-0123456789:%d" % k for k in range(n_gpus)], "distinct_devices": n_gpus},
        "rccl": {"nranks": n_gpus, "first_rank": 0, "n_local": 1, "version": "2.26.6", "version_code": 22606,
                 "library": "/usr/lib/python3.10/site-packages/torch/lib/librccl.so"} if n_gpus > 1 else None,
        "parity": {"checked_rays": 1000000, "rays_this_rank": 1000000, "equal": True, "csr_offsets_and_indices_equal": True, "visit_counters_equal": True,
                   "bvh_nodes_equal": True, "hits": 10000, "against": "oracle", "oracle_traverse_s": 0.21},
        "detail": "gpurun_out/bench_detail.json",
    }
    if n_gpus > 1:
        d["extra_configs"] = [_extra("standin-incoherent", config=3, scaling="strong", rays=100_000_000, plans=plans)]
        return d
    d["pipelined"] = {"streams": 2, "host_threads": 1, "steps": 20, "value": 4011.687, "unit": "Mrays/s", "ms_per_step": 0.2493,
                      "hits_every_step_equal": True, "hits": 10000}
    path = {"value": 1456.2, "unit": "Mrays/s", "ms_per_step": 0.6867, "delta_ms_vs_value": 0.36, "steps": 30,
            "bytes_per_step": {"aabbs_up": 2880000, "rays_up": 24000000, "csr_down": 4040004}, "pcie_gbs": 45.03, "csr_equal_to_pageable_path": True}
    d["step_excludes"] = {"steps": 100, "with_ray_gen": {"value": 2966.178, "unit": "Mrays/s", "ms_per_step": 0.3371, "delta_ms_vs_value": 0.0135},
                          "lazy_flat_array": {"value": 3150.2, "unit": "Mrays/s", "ms_per_step": 0.3174, "delta_ms_vs_value": -0.0062},
                          "host_io": dict(path, paths={"pageable": dict(path, value=802.5), "pinned": path, "pinned_one_call": dict(path, value=1440.1)})}
    d["extra_configs"] = [
        _extra("cubes120k", "closest"), _extra("cubes120k", "triangles"), _extra("standin-primary", "closest", config=2, rays=10_000_000),
        _extra("standin-primary", config=2, rays=10_000_000), _extra("standin-incoherent", config=3, rays=12_500_000),
        dict(_extra("cubes120k", dtype="f64", config=4), pure_f64_walk={"value": 2058.081, "unit": "Mrays/s", "ms_per_step": 0.4859, "steps": 20,
                                                                       "phases_ms": _phases(), "hits_all_ranks": 10000, "roofline": _roof(),
                                                                       "parity": {"equal": True, "checked_rays": 1000000}}),
        dict(_extra("standin-incoherent", config=3, scaling="strong", rays=100_000_000), hits_n1_reference=457389170, hits_match_n1_reference=True),
        _extra("cubes12m", config=None, rays=10_000_000), _extra("cubes120k", "closest", dtype="f64", config=4),
    ]
    d["extra_configs"][4]["first_ray"] = 62_500_000
    d["cpu_baseline"] = {
        "value": 73.2209, "value_median": 65.1, "unit": "Mrays/s", "cores": 128, "host_cpus_visible": 128, "kind": "port", "reps": 5,
        "sample": "oracle (C port, -O3 -march=native), pinned threads: full 120000-shape build + flatten + 1000000 of 1000000 rays walked once each, "
                  "scaled; min / median of 5 per phase",
        "build_ms": 2.91, "build_ms_median": 3.05, "build_threads": 64, "build_ms_task_recursion": 19.1, "task_threads": 8, "build_ms_serial": 33.9,
        "flatten_ms": 1.28, "traverse_ms_all_cores": 9.47, "traverse_ms_median": 11.2, "traverse_threads": 128, "traverse_ns_per_ray_1thread": 953.5,
        "sample_rays": 1000000, "native_build": True, "oracle_library": "liboracle_native.so", "omp_proc_bind": "close", "host_load_1m": [12.4, 30.1]}
    d["speedup_vs_cpu_baseline"] = 42.2
    return d


def test_full_line_fits_the_budget_and_keeps_the_contract():
    b = _bench()
    for n in (1, 8):
        text = b.render_line(_detail(n))
        assert len(text) < b.LINE_BUDGET == 8192, (n, len(text))
        assert "\n" not in text
        out = json.loads(text)
        for k in CONTRACT:
            assert k in out, k
        assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(out["roofline"]) and out["roofline"]["kernel"]
        assert out["config"]["workload"].startswith("configs[1]") and "model" not in out["config"]
        assert out["detail"] == "gpurun_out/bench_detail.json"
        if n == 1:
            cb = out["cpu_baseline"]
            assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb) and cb["kind"] == "port" and len(cb["host_load_1m"]) == 2
            assert len(out["extra_configs"]) == 9 and all(e["parity"]["equal"] is True for e in out["extra_configs"])
            assert out["step_excludes"]["host_io"] == 1456.2 and set(out["step_excludes"]["host_io_detail"]) == {"pageable", "pinned", "pinned_one_call"}
            assert out["extra_configs"][5]["pure_f64_walk"]["parity_equal"] is True and out["parity"] == {"equal": True, "checked_rays": 1000000,
                                                                                                         "bvh_nodes_equal": True}
        else:
            assert "cpu_baseline" not in out and out["rccl"]["nranks"] == 8 and len(out["launch"]["devices"]) == 8
            assert set(out["scene_dist_plans"]) == {"replicate", "bcast"}
            e = out["extra_configs"][0]
            assert e["hits_match_n1_reference"] is True and e["scene_dist_plans"]["bcast"]["hits_match_n1_reference"] is True
    # no prose on the line: the longest string is the one-sentence workload description / the cpu sample
    longest = max((len(v) for v in re.findall(r'"((?:[^"\\]|\\.)*)"', b.render_line(_detail(1)))), default=0)
    assert longest <= 220, longest


def test_a_line_that_would_not_fit_drops_rows_not_the_contract():
    b = _bench()
    d = _detail(1)
    d["extra_configs"] = d["extra_configs"] * 6        # 54 rows
    text = b.render_line(d)
    assert len(text) < b.LINE_BUDGET
    out = json.loads(text)
    assert out["extra_configs_dropped"] == 54 and "extra_configs" not in out
    assert out["value"] == 3089.913 and out["roofline"]["frac"] == 0.404 and out["cpu_baseline"]["value"] == 73.2209


def test_watchdog_line_stays_compact():
    """the line the watchdog prints when an exchange plan hangs goes through the same renderer"""
    b = _bench()
    d = _detail(8)
    pending = {"res": d["extra_configs"][0], "plan": "bcast", "stage": "the exchange plan's steps (communicator formed)", "workload": "standin-incoherent"}
    text = b.timed_out_line(d, pending, "the exchange plan of standin-incoherent", 60.0, {"nranks": 8, "version": "2.26.6"})
    out = json.loads(b.render_line(json.loads(text)))
    assert len(b.render_line(json.loads(text))) < b.LINE_BUDGET
    assert out["extra_configs"][0]["scene_dist_plans"]["bcast"]["timed_out"] is True and "did not finish within 60 s" in out["collective_watchdog"]


def test_committed_lines_of_this_round_are_compact():
    """every stdout line committed from round 6 on (profiles/r6+_*bench_default.json) is below the budget and parses with roofline and cpu_baseline"""
    b = _bench()
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")):
        m = re.match(r"r(\d+)_", os.path.basename(f))
        if int(m.group(1)) < 6:
            continue
        line = open(f).read().strip().splitlines()[-1]
        assert len(line) < b.LINE_BUDGET, (f, len(line))
        j = json.loads(line)
        assert j["roofline"]["frac"] is not None and j["cpu_baseline"]["value"] > 0 and j["parity"]["equal"] is True, f


def test_committed_line_is_what_the_renderer_makes_of_the_committed_detail():
    """the evidence chain's first link: every stdout line committed from round 6 on is exactly bench.render_line() of the detail file committed
    beside it — nothing on the line that the detail does not carry, nothing reformatted by hand"""
    b = _bench()
    seen = 0
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json"))):
        d = f.replace("_bench_default.json", "_bench_detail.json")
        if int(re.match(r"r(\d+)_", os.path.basename(f)).group(1)) < 6 or not os.path.exists(d):
            continue
        line = json.loads(open(f).read().strip().splitlines()[-1])
        again = json.loads(b.render_line(json.loads(open(d).read())))
        assert again == line, f
        seen += 1
    assert seen >= 1


def test_cpu_baseline_leg_runs_as_a_process_of_its_own(tmp_path):
    """oracle/baseline_leg.py — bench.py's cpu_baseline, pinned and free — on a tiny scene: one JSON object with the contract's keys"""
    import subprocess
    import sys
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import orc
    _, a = orc.create_n_cubes(300)
    np.save(tmp_path / "a.npy", a)
    np.save(tmp_path / "r.npy", orc.create_rays(0, 20000))
    for env_extra in ({}, {"OMP_PROC_BIND": "close", "OMP_PLACES": "cores"}):
        p = subprocess.run([sys.executable, "-m", "oracle.baseline_leg", str(tmp_path / "a.npy"), str(tmp_path / "r.npy"), "40000", "2"],
                           cwd=ROOT, env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-800:]
        j = json.loads(p.stdout.strip().splitlines()[-1])
        assert j["kind"] == "port" and j["value"] >= j["value_median"] > 0 and j["cores"] >= 1 and j["sample_rays"] == 20000
        assert j["build_ms"] <= j["build_ms_serial"] * 1.5 and len(j["host_load_1m"]) == 2 and len(j["sample"]) < 220
        assert ("pinned" if env_extra else "free") in j["sample"]


def test_source_files_keep_their_shape_budget():
    """bench.py = argument parsing + the timed step + the compact line; nothing wider than 140 columns in either file"""
    for name, max_lines in (("bench.py", 520), ("bench_sections.py", 800)):
        lines = open(os.path.join(ROOT, name)).read().splitlines()
        assert len(lines) <= max_lines, (name, len(lines))
        wide = [(i + 1, len(ln)) for i, ln in enumerate(lines) if len(ln) > 140]
        assert not wide, (name, wide[:5])
