cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 900 tools/profile_round.sh r02i 2 > gpurun_out/c55_prof.log 2>&1
tail -2 gpurun_out/c55_prof.log
python tools/make_counters_json.py r02i 2 > gpurun_out/c55_counters.log 2>&1; tail -3 gpurun_out/c55_counters.log
cp profiles/r02_counters_config2.json gpurun_out/r02_counters_config2.json
cp profiles/r02i_config2_kernel_stats.txt profiles/r02i_config2_pmc.txt gpurun_out/ 2>/dev/null
( timeout -s KILL 400 python bench.py --steps 20 --warmup 5 ) > gpurun_out/c55_line_config2.json 2> gpurun_out/c55_line_config2.err
( timeout -s KILL 500 python bench.py --config 3 --steps 10 --warmup 3 ) > gpurun_out/c55_line_config3.json 2> gpurun_out/c55_line_config3.err
( timeout -s KILL 500 python bench.py --config 4 --steps 10 --warmup 3 ) > gpurun_out/c55_line_config4.json 2> gpurun_out/c55_line_config4.err
( timeout -s KILL 600 python bench.py --config 5 --steps 5 --warmup 2 ) > gpurun_out/c55_line_config5.json 2> gpurun_out/c55_line_config5.err
( timeout -s KILL 400 python bench.py --qval --steps 10 --warmup 3 --no-e2e ) > gpurun_out/c55_line_config2q.json 2> /dev/null
( GX_BENCH_FORCE_RCCL=1 timeout -s KILL 400 python bench.py --steps 10 --warmup 3 --no-e2e --no-cpu ) > gpurun_out/c55_line_config2_rccl1.json 2> gpurun_out/c55_line_config2_rccl1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c55_line_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), round(d["value"],1), {k: round(v,3) for k,v in d["phases_ms"].items()}, d.get("gate",{}).get("passed"), d.get("gate",{}).get("narrowpeak_diff"), round(d["roofline"]["frac"],3), d["config"].get("collectives"))
    except Exception as e: print(f, "ERR", e)
PY
