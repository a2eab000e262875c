cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 300 python -m pytest tests/test_hip_parity.py -m gpu -q --timeout 300 -x 2>&1 | tail -2
( timeout -s KILL 200 python bench.py --steps 20 --warmup 3 --no-cpu --no-e2e ) > gpurun_out/c11_bench2.json 2> gpurun_out/c11_bench2.err
( timeout -s KILL 300 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu --no-e2e ) > gpurun_out/c11_bench4.json 2> gpurun_out/c11_bench4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c11_bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["phases_ms"].items()}, d["config"]["peaks"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
timeout -s KILL 900 tools/profile_round.sh r02b 2 > gpurun_out/c11_prof.log 2>&1
tail -3 gpurun_out/c11_prof.log
