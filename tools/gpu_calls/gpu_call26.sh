cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/c26_pytest.log 2>&1
grep -E "passed|failed|^FAILED|Error" gpurun_out/c26_pytest.log | tail -12
( timeout -s KILL 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-e2e ) > gpurun_out/c26_bench_config2.json 2> gpurun_out/c26_bench_config2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c26_bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["phases_ms"].items()}, d["config"]["peaks"], d.get("gate",{}).get("passed"))
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
GX_PROF_PASSES=trace timeout -s KILL 400 tools/profile_round.sh r02d5 5 > gpurun_out/c26_prof5.log 2>&1; tail -3 gpurun_out/c26_prof5.log
GX_PROF_PASSES=trace timeout -s KILL 400 tools/profile_round.sh r02d3 3 > gpurun_out/c26_prof3.log 2>&1; tail -3 gpurun_out/c26_prof3.log
for t in r02d5 r02d3; do f=$(ls gpurun_out/prof_$t/trace/*kernel_stats.csv 2>/dev/null | head -1); echo "== $t $f"; head -30 "$f" | cut -d, -f1-6 | cut -c1-150; done
echo "---- emulate ranks"
timeout -s KILL 300 python tools/emulate_ranks.py 1 8 2>&1 | grep -v amdgpu.ids | tail -4
timeout -s KILL 120 python tools/host_floor.py 2>&1 | tail -1
