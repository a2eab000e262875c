cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 300 python tools/diag_sort.py 2>&1 | grep -E "^run" > gpurun_out/c10_diag.log 2>&1
cat gpurun_out/c10_diag.log
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/c10_pytest.log 2>&1
tail -4 gpurun_out/c10_pytest.log
( timeout -s KILL 200 python bench.py --steps 20 --warmup 3 --no-cpu --no-e2e ) > gpurun_out/c10_bench2.json 2> gpurun_out/c10_bench2.err
( timeout -s KILL 300 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu --no-e2e ) > gpurun_out/c10_bench4.json 2> gpurun_out/c10_bench4.err
( timeout -s KILL 300 python bench.py --config 3 --steps 5 --warmup 2 --no-cpu --no-e2e ) > gpurun_out/c10_bench3.json 2> gpurun_out/c10_bench3.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c10_bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["phases_ms"].items()}, d["config"]["peaks"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
