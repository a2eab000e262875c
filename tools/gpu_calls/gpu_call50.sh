cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/c50_pytest.log 2>&1
grep -E "passed|failed|^FAILED|Error" gpurun_out/c50_pytest.log | tail -12
( timeout -s KILL 600 python bench.py --config 5 --steps 4 --warmup 2 --no-e2e --no-cpu ) > gpurun_out/c50_bench_config5.json 2> gpurun_out/c50_bench_config5.err
( timeout -s KILL 600 python bench.py --config 3 --steps 5 --warmup 2 --no-e2e --no-cpu ) > gpurun_out/c50_bench_config3.json 2> gpurun_out/c50_bench_config3.err
( timeout -s KILL 400 python bench.py --qval --steps 10 --warmup 3 --no-e2e --no-cpu ) > gpurun_out/c50_bench_config2q.json 2> /dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c50_bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["phases_ms"].items()}, d["config"]["peaks"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
