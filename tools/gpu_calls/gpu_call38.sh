cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_emulall/trace -o emul -- python tools/emulate_ranks.py 1 2 4 8 > gpurun_out/c38_emul.log 2>&1
grep -v amdgpu gpurun_out/c38_emul.log | grep " ms " | tail -4
python - <<'PY'
import sys, sqlite3
sys.path.insert(0,'tools')
import make_counters_json as M
db=M.db_of('gpurun_out/prof_emulall/trace')
c=sqlite3.connect(db); t=M.tables(c)
kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
rows=list(c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
names=[M.short(r[0].split('(')[0].replace('void ','')) for r in rows]
for k in ('k_peaks_write','k_peak_short','k_sort1','k_runs_write','k_cand_hdr'):
    d=[(rows[i][2]-rows[i][1])/1e3 for i,n in enumerate(names) if n==k]
    print(k, [round(x,1) for x in d[::4]])
PY
