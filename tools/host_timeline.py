"""developer tool: kernels AND memory copies of one host-resident step on a common time axis, from a rocprofv3 --kernel-trace
--memory-copy-trace database:  python tools/host_timeline.py <results.db> [step index]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kern = db.execute("select name,start,end from kernels order by start").fetchall()
mc_tab = [t for t in tabs if t == "memory_copies"] or [t for t in tabs if "memory_copy" in t.lower() or "memory_copies" in t.lower()]
cols = [c[1] for c in db.execute(f"pragma table_info({mc_tab[0]})")]
name_col = "name" if "name" in cols else cols[0]
size_col = "size" if "size" in cols else None
q = f"select {name_col},start,end{',' + size_col if size_col else ''} from {mc_tab[0]} order by start"
copies = db.execute(q).fetchall()
ev = [("K " + n.split("(")[0].replace("void ", "").replace("bvhgpu::", "")[:40], s, e, 0) for n, s, e in kern]
ev += [("C " + str(c[0])[:30], c[1], c[2], c[3] if len(c) > 3 else 0) for c in copies]
ev.sort(key=lambda r: r[1])
idx = [i for i, r in enumerate(ev) if "k_prep" in r[0]]
i0, i1 = idx[k], idx[k + 1]
# the step's first event may be a copy just before k_prep (the AABB upload): start two events earlier
j0 = i0
while j0 > 0 and ev[j0 - 1][0].startswith("C ") and ev[i0][1] - ev[j0 - 1][1] < 200_000:
    j0 -= 1
t0 = ev[j0][1]
for n, s, e, sz in ev[j0:i1]:
    if ev[i1][1] - s < 0:
        break
    print(f"{n:44s} start {(s - t0) / 1e3:8.2f}  end {(e - t0) / 1e3:8.2f}  dur {(e - s) / 1e3:7.2f}" + (f"  {sz / 1e6:.2f} MB" if sz else ""))
print(f"step: {(ev[i1][1] - ev[i0][1]) / 1e3:.2f} us from k_prep to the next k_prep")
