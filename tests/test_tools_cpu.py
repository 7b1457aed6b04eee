"""tools/ holds developer diagnostics that only run on the GPU box; what can be checked everywhere is that they still parse
(VERDICT r2 hygiene: "~25 one-off diagnostics with no test") and that the two summarisers work on a synthetic counter dump."""
import glob
import json
import os
import py_compile
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
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r4_*_bench_default.json")))
    assert lines, "no committed default bench line under profiles/"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_numbers.py"), lines[-1]], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    j = json.loads(open(lines[-1]).read().strip().splitlines()[-1])
    assert f"{j['value']:.0f} Mrays/s" in p.stdout and "CSR assembly" in p.stdout and "Pure f64 walk" in p.stdout and "CPU baseline" in p.stdout
    assert p.stdout.count("parity equal: true") + p.stdout.count("`equal: true`") >= 1 + len(j.get("extra_configs", []))
