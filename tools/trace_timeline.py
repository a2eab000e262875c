"""Timeline of the last step in a rocprofv3 kernel trace (rocpd .db): start, gap to the previous kernel, duration.
   python tools/trace_timeline.py gpurun_out/prof_<tag>/trace [first-kernel-name]"""
import sys, sqlite3
sys.path.insert(0, 'tools')
import make_counters_json as M
db = M.db_of(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else 'k_sort1'
nth = int(sys.argv[3]) if len(sys.argv) > 3 else -1
c = sqlite3.connect(db); t = M.tables(c)
kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
names = [M.short(r[0].split('(')[0].replace('void ', '')) for r in rows]
idx = [i for i, n in enumerate(names) if n == first]
a = idx[nth]
b = idx[nth + 1] if nth != -1 and nth + 1 < len(idx) else len(rows)
t0 = rows[a][1]; prev = None
for i in range(a, b):
    s, e = rows[i][1], rows[i][2]
    gap = (s - prev) / 1e3 if prev else 0
    print(f"{(s-t0)/1e3:9.1f} us  +gap {gap:7.1f}  dur {(e-s)/1e3:8.1f}  {names[i][:40]}")
    prev = e
