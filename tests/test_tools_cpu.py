"""tools/ holds developer diagnostics that only run on the GPU box; what can be checked everywhere is that they still parse
(VERDICT r2 hygiene: "~25 one-off diagnostics with no test") and that the two summarisers work on a synthetic counter dump."""
import glob
import json
import os
import py_compile
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))), ids=os.path.basename)
def test_tool_compiles(path):
    py_compile.compile(path, doraise=True)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh"))), ids=os.path.basename)
def test_tool_script_parses(path):
    subprocess.check_call(["bash", "-n", path])


def test_pmc_summary_on_a_synthetic_dump(tmp_path):
    """tools/pmc_summary.py turns rocprofv3 --pmc CSVs into bound.json — the file bench.py's `roofline` reads.  A synthetic pass
    (one kernel, known counters) must come out with the documented corrections: FETCH_SIZE KiB doubled, fractions of the peaks, the
    workload tag taken from the bench line of the profiled command."""
    d = tmp_path / "pmc_X" / "run"
    d.mkdir(parents=True)
    rows = [("FETCH_SIZE", 1000.0), ("WRITE_SIZE", 500.0), ("SQ_INSTS_VALU", 1.0e6), ("SQ_INSTS_LDS", 1.0e5), ("SQ_LDS_BANK_CONFLICT", 2.0e5),
            ("SQ_WAIT_ANY", 4.0e6), ("SQ_WAVE_CYCLES", 8.0e6)]
    with open(d / "out_counter_collection.csv", "w") as f:
        f.write("Kernel_Name,Counter_Name,Counter_Value\n")
        for c, v in rows:
            for _ in range(2):
                f.write(f'"void bvhgpu::k_traverse_wide<float, 0, 0, 1024, 8>(args)",{c},{v}\n')
    (tmp_path / "bench_under_rocprof.json").write_text(json.dumps({"workload_name": "standin-primary", "dtype": "f32", "config": {"rays_per_gpu": 10_000_000}}) + "\n")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(tmp_path)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    b = json.load(open(tmp_path / "bound.json"))
    assert (b["workload"], b["dtype"], b["rays_per_launch"]) == ("standin-primary", "f32", 10_000_000)
    k = b["kernels"][0]
    assert k["kernel"].startswith("bvhgpu::k_traverse_wide<float, 0, 0")
    assert k["hbm_bytes"] == 1000.0 * 1024 * 2 + 500.0 * 1024 and k["wait_frac"] == 0.5 and k["launches"] == 2


def test_prof_summary_on_a_synthetic_rocpd_database(tmp_path):
    """tools/prof_summary.py reads the `kernels` view of a rocprofv3 rocpd database and writes the per-kernel table committed under
    profiles/: calls, total / mean / min / max in µs (the view holds ns), share of the total, registers and launch geometry."""
    import sqlite3
    db = tmp_path / "out_results.db"
    con = sqlite3.connect(db)
    con.execute("create table kernels (name text, duration integer, vgpr_count integer, sgpr_count integer, lds_size integer, grid_x integer, workgroup_x integer)")
    con.executemany("insert into kernels values (?,?,?,?,?,?,?)",
                    [("void bvhgpu::k_traverse_wide<float, 0, 2, 1024, 8, 0>(args)", 120_000, 64, 96, 81312, 524288, 1024),
                     ("void bvhgpu::k_traverse_wide<float, 0, 2, 1024, 8, 0>(args)", 124_000, 64, 96, 81312, 524288, 1024),
                     ("void bvhgpu::k_level<float, false>(args)", 10_000, 88, 112, 13400, 61440, 256)])
    con.commit(); con.close()
    out = tmp_path / "kernel_stats.md"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), str(db), str(out), "synthetic"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    lines = out.read_text().splitlines()
    assert lines[0] == "# synthetic"
    rows = [ln for ln in lines if ln.startswith("| `bvhgpu::")]
    assert rows[0].startswith("| `bvhgpu::k_traverse_wide<float, 0, 2, 1024, 8, 0>` | 2 | 244.0 | 122.00 | 120.00 | 124.00 | 96.1 | 64 | 96 | 81312 | 524288 | 1024 |")
    assert rows[1].startswith("| `bvhgpu::k_level<float, false>` | 1 | 10.0 | 10.00 |")


def test_design_numbers_reads_the_committed_bench_line():
    """tools/design_numbers.py renders DESIGN.md §7 from a bench line: run on the newest committed line of a default bench run, it must name the
    headline, every extra config with its CSR-assembly share, the pure f64 walk and the CPU baseline."""
    path, j = _newest_default_line()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_numbers.py"), path], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert f"{j['value']:.0f} Mrays/s" in p.stdout and "CSR assembly" in p.stdout and "Pure f64 walk" in p.stdout and "CPU baseline" in p.stdout
    assert p.stdout.count("parity equal: true") + p.stdout.count("`equal: true`") >= 1 + len(j.get("extra_configs", []))


def _newest_default_line():
    def key(f):
        m = re.match(r"r(\d+)_v(\d+)_bench_default\.json", os.path.basename(f))
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_v*_bench_default.json")), key=key)
    assert lines, "no committed default bench line under profiles/"
    # round 6 on: the stdout line is compact (tests/test_bench_line_cpu.py) and every section is in the run's detail file beside it
    detail = lines[-1].replace("_bench_default.json", "_bench_detail.json")
    if os.path.exists(detail):
        d = json.loads(open(detail).read())
        line = json.loads(open(lines[-1]).read().strip().splitlines()[-1])
        assert d["value"] == line["value"] and d["roofline"]["frac"] == line["roofline"]["frac"], "detail file and stdout line are not of one run"
        return detail, d
    return lines[-1], json.loads(open(lines[-1]).read().strip().splitlines()[-1])


def test_committed_bench_line_follows_from_the_committed_profiles():
    """VERDICT r4 #8 — the evidence chain, checked on every CPU run.  The newest committed line of a default bench run
    (profiles/r<N>_v<M>_bench_default.json) must be reproducible from the profile files it cites:
      (a) roofline.kernel names a kernel that is a row of the cited round's kernel_stats.md (a template-argument change in traverse.hip
          can no longer silently drop the roofline),
      (b) roofline.frac recomputes from the cited bound.json's counters and the line's own kernel time, within 3 %,
      (c) the kernels of one step, at their profiled average durations, fit into the line's ms_per_step."""
    path, j = _newest_default_line()
    roof = j["roofline"]
    assert roof.get("frac") is not None and roof.get("source"), f"{path}: the headline carries no PMC-derived roofline"
    bound_file = roof["source"].split(":")[0]
    assert os.path.exists(os.path.join(ROOT, bound_file)), bound_file
    stats_file = bound_file.replace("_bound.json", "_kernel_stats.md")
    rows = {}
    for ln in open(os.path.join(ROOT, stats_file)):
        m = re.match(r"\| `([^`]+)` \| (\d+) \| [\d.]+ \| ([\d.]+) \|", ln)
        if m:
            rows[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    # (a)
    hit = [k for k in rows if k.startswith(roof["kernel"])]
    assert len(hit) == 1, (roof["kernel"], sorted(rows)[:5])
    # (b)
    bj = json.load(open(os.path.join(ROOT, bound_file)))
    c = next(k for k in bj["kernels"] if k["kernel"] == hit[0])
    secs = roof["kernel_ms"] * 1e-3
    scale = j["config"]["rays_per_gpu"] / bj["rays_per_launch"]      # (per-launch counters are per ray to first order: bench.py newest_bound)
    peaks = {"hbm": (c.get("hbm_bytes"), 8e12), "valu": (c.get("SQ_INSTS_VALU"), 1024 * 2.4e9 / 2),
             "lds": ((c.get("SQ_INSTS_LDS") or 0) * 4 + (c.get("SQ_LDS_BANK_CONFLICT") or 0), 256 * 2.4e9)}
    count, peak = peaks[roof["bound"]]
    frac = count * scale / secs / peak
    assert abs(frac - roof["frac"]) <= 0.03 * roof["frac"], (frac, roof["frac"])
    # (c): the walk runs once per step; every kernel that ran at least once per step is priced at calls-per-step x its average
    steps = rows[hit[0]][0]
    per_step_us = sum(avg * round(calls / steps) for calls, avg in rows.values() if calls >= 0.9 * steps)
    assert per_step_us <= j["ms_per_step"] * 1e3 * 1.03, (per_step_us, j["ms_per_step"])
    assert per_step_us >= 0.7 * j["ms_per_step"] * 1e3, "the profile and the line do not describe the same step"
