cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/c15_pytest.log 2>&1
grep -E "passed|failed|Error|error" gpurun_out/c15_pytest.log | tail -5
( timeout -s KILL 200 python bench.py --steps 20 --warmup 3 --no-cpu --no-e2e ) > gpurun_out/c15_bench2.json 2> gpurun_out/c15_bench2.err
( GX_NO_LOOSE=1 timeout -s KILL 200 python bench.py --steps 20 --warmup 3 --no-cpu --no-e2e ) > gpurun_out/c15_bench2_noloose.json 2> /dev/null
( timeout -s KILL 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-e2e --qval ) > gpurun_out/c15_bench2q.json 2> /dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c15_bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["phases_ms"].items()}, d["config"]["peaks"])
    except Exception as e: print(f, "ERR", e, open("gpurun_out/c15_bench2.err").read()[-500:])
PY
