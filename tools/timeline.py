"""developer tool: per-dispatch timeline (start, duration, gap to the previous dispatch's end) of the n-th step found in a
rocprofv3 --kernel-trace database:  python tools/timeline.py <out_results.db> [step index, default 10]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = db.execute("select name,start,end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_prep" in r[0]]
i0, i1 = idx[k], idx[k + 1]
t0 = rows[i0][1]
prev = None
for n, s, e in rows[i0:i1]:
    short = n.split("(")[0].replace("void ", "").replace("bvhgpu::", "")[:44]
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{short:46s} start {(s - t0) / 1e3:8.2f}  dur {(e - s) / 1e3:7.2f}  gap {gap:6.2f}")
    prev = e
print(f"step: first start -> last end {(rows[i1 - 1][2] - t0) / 1e3:.2f} us; to next k_prep start {(rows[i1][1] - t0) / 1e3:.2f} us")
