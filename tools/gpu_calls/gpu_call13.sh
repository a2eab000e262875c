cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 600 python -m pytest tests/test_hip_parity.py tests/test_hip_multirank.py -m gpu -q --timeout 300 2>&1 | grep -E "passed|failed"
for i in 1 2 3; do ( timeout -s KILL 200 python bench.py --steps 20 --warmup 3 --no-cpu --no-e2e ) > gpurun_out/c13_bench2_$i.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c13_bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["phases_ms"].items()}, d["config"]["peaks"])
    except Exception as e: print(f, "ERR", e)
PY
