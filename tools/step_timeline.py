"""print the kernel timeline of one steady-state bench step from a rocprofv3 rocpd database (tools/build_prof.sh)"""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_prep" in r[0]]
i0, i1 = idx[10], idx[11]
t0, prev = rows[i0][1], None
for r in rows[i0:i1]:
    name = r[0].split("(")[0].replace("void bvhgpu::", "")[:34]
    gap = (r[1] - prev) / 1e3 if prev else 0
    print(f"{(r[1] - t0) / 1e3:8.1f} us  +gap {gap:5.1f}  dur {(r[2] - r[1]) / 1e3:6.1f}  {name}")
    prev = r[2]
print("step span", (rows[i1][1] - t0) / 1e3)
