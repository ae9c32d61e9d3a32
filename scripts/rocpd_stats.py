#!/usr/bin/env python
"""Per-kernel summary (count / total / avg / share) from a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db in this ROCm).
    python scripts/rocpd_stats.py gpurun_out/prof/x_results.db [--top 25] > profiles/xxx_kernel_stats.txt
"""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 30
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % disp)]
    scol = [r[1] for r in c.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else scol[-1])
    q = ("select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, disp, sym, name_col))
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows)
    print("# %s : %d kernel symbols, %d dispatches, %.3f ms total GPU kernel time" %
          (db.split("/")[-1], len(rows), sum(r[1] for r in rows), total / 1e6))
    print("%-86s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for name, n, tot, mn, mx in rows[:top]:
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*$", "", name)
        print("%-86s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (name[:86], n, tot / 1e3, tot / n / 1e3, mn / 1e3, mx / 1e3,
                                                                100.0 * tot / total))


if __name__ == "__main__":
    main()
