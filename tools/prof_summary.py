"""Turn a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats -d DIR -o NAME) into the
per-kernel summary committed under profiles/.   python tools/prof_summary.py <results.db> <out.md> [title]"""
import sqlite3
import sys


def main():
    db_path, out_path = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else db_path
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name "
        "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out_path, "w") as f:
        f.write(f"# {title}\n\nsource: `rocprofv3 --kernel-trace --stats` (rocpd database → `kernels` view); durations in µs\n\n")
        f.write("| kernel | calls | total µs | avg µs | min µs | max µs | % | VGPR | SGPR | LDS B | max grid | wg |\n")
        f.write("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
        for n, c, s, a, mn, mx, vg, sg, lds, gx, wx in rows:
            short = n.split("(")[0].replace("void ", "")
            f.write(f"| `{short}` | {c} | {s / 1e3:.1f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | "
                    f"{100 * s / total:.1f} | {vg} | {sg} | {lds} | {gx} | {wx} |\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main()
