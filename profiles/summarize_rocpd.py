#!/usr/bin/env python3
"""Per-kernel summary (calls, total / avg / min / max duration, share) from a rocprofv3 rocpd
SQLite file (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` -> DIR/NAME_results.db)."""
import sqlite3
import sys


def main(db, skip_first=0):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id "
                     f"order by d.start").fetchall()
    agg = {}
    for name, a, b in rows:
        name = name.split("(")[0].replace("void ", "")
        agg.setdefault(name, []).append(b - a)
    tot = sum(sum(v) for v in agg.values())
    print(f"{'kernel':60s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{name[:60]:60s} {len(v):6d} {sum(v)/1e6:10.3f} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} "
              f"{max(v)/1e3:10.2f} {100*sum(v)/tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
