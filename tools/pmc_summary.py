"""Summarise the --pmc passes of tools/profile_round.sh: per-kernel mean counter values per launch (pmc_summary.md on
stdout), and <out>/bound.json — for every kernel the counters per launch, the average duration of the kernel-trace pass
and the fraction of each resource's peak they amount to, which bench.py turns into `roofline` (binding resource, frac <= 1).

Resources (VERDICT r1 "make roofline honest"; peaks from MI355X_MICROARCH.md):
  hbm   (FETCH_SIZE * 2 + WRITE_SIZE) KiB per launch / t / 8 TB/s.  rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE counts
        128-byte requests as 64 bytes for wide reads ("HBM" section), so the read figure is doubled as that section
        prescribes; WRITE_SIZE is uncalibrated there and is taken as is
  valu  SQ_INSTS_VALU (wave64 instructions) * 2 cycles / (1024 SIMDs * 2.4 GHz) / t
  lds   (SQ_INSTS_LDS * 4 + SQ_LDS_BANK_CONFLICT) LDS-array cycles / (256 CUs * 2.4 GHz) / t
  wait  SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of the resident wave-time spent parked on s_waitcnt / barriers (not a throughput)
  lane utilisation is not a PMC figure: bench.py's STATS run reports device_steps / (64 * wave_steps)
usage: python tools/pmc_summary.py <out_dir> [kernel_trace.db]
"""
import collections
import csv
import glob
import json
import os
import sqlite3
import sys

out = sys.argv[1]
db_path = sys.argv[2] if len(sys.argv) > 2 else None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
kernels = sorted(agg, key=lambda k: -sum(sum(v) for v in agg[k].values()))
names = sorted({c for k in agg for c in agg[k]})
print("# PMC counters per launch (mean over the launches of the bench command)\n")
print("| kernel | launches | " + " | ".join(names) + " |")
print("|---|---:|" + "---:|" * len(names))
for k in kernels:
    n = max(len(v) for v in agg[k].values())
    print(f"| `{k}` | {n} | " + " | ".join(f"{sum(agg[k][c]) / len(agg[k][c]):.4g}" if agg[k][c] else "" for c in names) + " |")

dur = {}
if db_path and os.path.exists(db_path):
    db = sqlite3.connect(db_path)
    for name, cnt, avg, tot in db.execute("select name, count(*), avg(duration), sum(duration) from kernels group by name"):
        dur[name.split("(")[0].replace("void ", "")] = (cnt, avg / 1e3, tot / 1e3)

HBM, VALU, LDS = 8e12, 1024 * 2.4e9 / 2, 256 * 2.4e9


def mean(k, c):
    v = agg[k].get(c)
    return sum(v) / len(v) if v else None


rows = []
for k in kernels:
    fetch, write = mean(k, "FETCH_SIZE"), mean(k, "WRITE_SIZE")
    e = {"kernel": k, "launches": max(len(v) for v in agg[k].values())}
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_INSTS_VMEM_RD",
              "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"):
        e[c] = mean(k, c)
    e["fetch_bytes_corrected"] = None if fetch is None else fetch * 1024 * 2
    e["write_bytes"] = None if write is None else write * 1024
    e["hbm_bytes"] = None if fetch is None else e["fetch_bytes_corrected"] + (e["write_bytes"] or 0)
    e["wait_frac"] = round(e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"], 4) if e["SQ_WAIT_ANY"] and e["SQ_WAVE_CYCLES"] else None
    if k in dur:
        e["calls_in_trace"], e["avg_us"], e["total_us"] = dur[k][0], round(dur[k][1], 3), round(dur[k][2], 1)
        t = dur[k][1] * 1e-6
        if t > 0:
            if e["hbm_bytes"] is not None:
                e["hbm_frac"] = round(e["hbm_bytes"] / t / HBM, 4)
            if e["SQ_INSTS_VALU"] is not None:
                e["valu_frac"] = round(e["SQ_INSTS_VALU"] / t / VALU, 4)
            if e["SQ_INSTS_LDS"] is not None:
                e["lds_frac"] = round((e["SQ_INSTS_LDS"] * 4 + (e["SQ_LDS_BANK_CONFLICT"] or 0)) / t / LDS, 4)
            fr = {r: e[r + "_frac"] for r in ("hbm", "valu", "lds") if e.get(r + "_frac") is not None}
            if fr:
                e["bound"] = max(fr, key=fr.get)
    rows.append(e)
rows.sort(key=lambda e: -(e.get("total_us") or 0))
tag = {"workload": "cubes120k", "dtype": "f32", "rays_per_launch": 1_000_000}   # what the profiled bench command ran (bench.py matches on it)
try:
    bj = json.loads(open(os.path.join(out, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
    tag = {"workload": bj.get("workload_name", "cubes120k"), "dtype": bj.get("dtype", "f32"), "rays_per_launch": bj["config"]["rays_per_gpu"]}
except Exception:
    pass
json.dump(dict(tag, note=__doc__.split("usage:")[0].strip(), kernels=rows), open(os.path.join(out, "bound.json"), "w"), indent=1)
print("\n## resource fractions (counters per launch / average duration of the kernel-trace pass)\n")
print("| kernel | avg µs | hbm | valu | lds | wait | bound |")
print("|---|---:|---:|---:|---:|---:|---|")
for e in rows:
    if "avg_us" in e:
        print(f"| `{e['kernel']}` | {e['avg_us']} | {e.get('hbm_frac', '')} | {e.get('valu_frac', '')} | {e.get('lds_frac', '')} | "
              f"{e.get('wait_frac', '')} | {e.get('bound', '')} |")
