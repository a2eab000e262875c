cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/c54_pytest.log 2>&1
grep -E "passed|failed|^FAILED|Error" gpurun_out/c54_pytest.log | tail -12
run() { tag=$1; shift; ( env "$@" timeout -s KILL 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-e2e ) > gpurun_out/c54_bench_$tag.json 2> gpurun_out/c54_bench_$tag.err; }
run dflt
( timeout -s KILL 600 python bench.py --config 4 --steps 5 --warmup 2 --no-e2e ) > gpurun_out/c54_bench_config4.json 2> gpurun_out/c54_bench_config4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c54_bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["phases_ms"].items()}, d["config"]["peaks"], d.get("gate",{}).get("passed"))
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
