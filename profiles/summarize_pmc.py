#!/usr/bin/env python3
"""Per-kernel mean of one PMC counter from a rocprofv3 rocpd SQLite file
(`rocprofv3 --pmc NAME --kernel-trace -d DIR -o X -- cmd`)."""
import sqlite3
import sys


def main(db):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tabs if x.startswith(p))  # noqa: E731
    pe, kd, ks, ip = t("rocpd_pmc_event"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_info_pmc")
    cols = [r[1] for r in c.execute(f"pragma table_info({pe})")]
    rows = c.execute(
        f"select s.kernel_name, i.name, e.value from {pe} e join {kd} d on e.event_id = d.event_id "
        f"join {ks} s on d.kernel_id = s.id join {ip} i on e.pmc_id = i.id").fetchall()
    agg = {}
    for name, ctr, v in rows:
        name = name.split("(")[0].replace("void ", "")
        agg.setdefault((name, ctr), []).append(v)
    print(f"{'kernel':58s} {'counter':12s} {'calls':>6s} {'mean':>16s}")
    for (name, ctr), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
        print(f"{name[:58]:58s} {ctr:12s} {len(v):6d} {sum(v)/len(v):16.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
