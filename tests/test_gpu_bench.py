"""bench.py is load-bearing evidence (the driver's record is its one JSON line), so its N = 1 sections run here at a small size on every GPU
test round: the headline with its roofline lookup by the kernel name the LIBRARY reports, `step_excludes` (ray generation / eager FlatNode
array / host I/O beside `value`), the reference's whole harness loop as extra configs with parity and the CPU loop beside it, the fair CPU
baseline, and `--harness` as the headline of a run (what the profile rounds use)."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_args, env_extra=None):
    import bvh_amd
    if bvh_amd.device_count() <= 0:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--cubes", "2500", "--rays", "200000", "--steps", "5", "--warmup", "2",
           "--settle-steps", "10", "--extra-steps", "3", "--cpu-sample-rays", "100000"] + extra_args
    detail_file = os.path.join(tempfile.mkdtemp(prefix="bvh_bench_"), "detail.json")
    p = subprocess.run(cmd + ["--detail-out", detail_file], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and p.stdout.strip().splitlines()[-1] == lines[0], p.stdout[-2000:]     # ONE line, and it is the last one
    assert len(lines[0]) < 8192, len(lines[0])                                                     # the driver must be able to read it (BENCH_r05)
    return json.loads(lines[0]), json.loads(open(detail_file).read())


def test_default_line_sections_at_a_small_size():
    out, det = _run([], {"BVH_BENCH_EXTRAS": "0,1,8"})      # extras: the harness loop on the cube scene — closest, triangles, closest in f64
    assert out["value"] > 0 and out["n_gpus"] == 1 and out["steps"] == 5 and out["warmup"] == 2 and out["settle_steps"] == 10
    assert out["value"] == det["value"] and out["config"]["flat_array"] == "eager"    # the FlatNode array is written inside the timed step
    assert out["parity"]["equal"] is True and out["parity"]["checked_rays"] == 200000
    roof = out["roofline"]
    assert roof["kernel"].startswith("bvhgpu::k_traverse_wide<float, 0, ") and roof["kernel"].endswith(">")      # as the library spells it
    assert roof["algorithmic_frac"] > 0 and set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(roof)
    ex, exd = out["step_excludes"], det["step_excludes"]
    for k in ("with_ray_gen", "lazy_flat_array", "all_arrays_eager", "beside_flat_array", "host_io"):
        assert ex[k] > 0 and exd[k]["value"] == ex[k] and exd[k]["ms_per_step"] > 0, ex
    pg, pn = exd["host_io"]["paths"]["pageable"], exd["host_io"]["paths"]["pinned"]
    assert pg["bytes_per_step"]["aabbs_up"] == 2500 * 12 * 24 and pg["bytes_per_step"]["rays_up"] == 200000 * 36
    assert pn["bytes_per_step"]["rays_up"] == 200000 * 24 and pn["csr_equal_to_pageable_path"] is True      # origins + directions only
    p1 = exd["host_io"]["paths"]["pinned_one_call"]
    assert p1["csr_equal_to_pageable_path"] is True and p1["value"] > 0
    assert ex["host_io"] == max(pg["value"], pn["value"], p1["value"]) < out["value"]          # PCIe-inclusive: never the faster one
    assert out["pipelined"]["hits_every_step_equal"] is True
    modes = []
    for e, ed in zip(out["extra_configs"], det["extra_configs"]):
        assert "error" not in e, ed
        modes.append((e["harness"], e["dtype"]))
        assert e["workload"] == "cubes120k+" + e["harness"] and e["parity"]["equal"] is True and e["parity"]["checked_rays"] == e["rays_this_rank"]
        # (the f64 closest-hit walk runs over the tree's f32 guide boxes: k_traverse_wide<float, 3, 2, 1024, 8, 1>)
        assert e["ray_gen_ms"] > 0 and ed["roofline"]["kernel"].startswith("bvhgpu::k_traverse_wide<float, %d, " % (3 if e["harness"] == "closest" else 2))
        assert ed["roofline"]["kernel"].endswith(", 1>" if e["dtype"] == "f64" else ", 0>")
        if e["dtype"] == "f32":
            assert e["cpu_harness"] > 0 and e["speedup_vs_cpu_harness"] > 0 and ed["cpu_harness"]["oracle_library"].startswith("liboracle")
    assert modes == [("closest", "f32"), ("triangles", "f32"), ("closest", "f64")]
    cb, cbd = out["cpu_baseline"], det["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] >= cb["value_median"] > 0 and len(cb["host_load_1m"]) == 2 and len(cb["sample"]) < 220
    assert cbd["build_threads"] >= 1 and cbd["build_ms"] <= cbd["build_ms_serial"] * 1.05 and set(cbd["legs"]) == {"free", "pinned"} and cb["threads"] in cbd["legs"]


@pytest.mark.parametrize("mode", ["closest", "triangles"])
def test_harness_as_the_headline(mode):
    """`bench.py --harness M`: the step is ray generation + build + flatten + walk + triangle stage (intersect_bh, testbase.rs:819-837); the line is
    tagged so that its profile is looked up under its own workload name"""
    out, det = _run(["--harness", mode, "--no-extra", "--no-cpu-baseline", "--pipeline-streams", "0"])
    assert out["workload_name"] == "cubes120k+" + mode and out["harness"] == mode and out["value"] > 0
    assert out["parity"]["equal"] is True and out["cpu_harness"]["value"] > 0
    assert "intersect_bh" in out["config"]["workload"] and out["phases_ms"]["ray_gen_ms"] > 0
    assert "step_excludes" not in out        # (those compare against the index step of `value`: not measured beside a harness headline)
