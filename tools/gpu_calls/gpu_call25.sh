cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/c25_pytest.log 2>&1
grep -E "passed|failed|^FAILED|Error" gpurun_out/c25_pytest.log | tail -12
run() { tag=$1; shift; ( env "$@" timeout -s KILL 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-e2e ) > gpurun_out/c25_bench_$tag.json 2> gpurun_out/c25_bench_$tag.err; }
run dflt GX_DEBUG=1
for w in 18 20 22 24 26 28; do run wg$w GX_TILE_FAST_WG=$w; done
( timeout -s KILL 600 python bench.py --config 5 --steps 5 --warmup 2 --no-e2e ) > gpurun_out/c25_bench_config5.json 2> gpurun_out/c25_bench_config5.err
( timeout -s KILL 600 python bench.py --config 3 --steps 5 --warmup 2 --no-e2e --no-cpu ) > gpurun_out/c25_bench_config3.json 2> gpurun_out/c25_bench_config3.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c25_bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["phases_ms"].items()}, d["config"]["peaks"], d.get("gate",{}).get("passed"))
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
grep "k_tile workgroups" gpurun_out/c25_bench_dflt.err | head -2
echo "---- rccl exercise"
GX_BENCH_FORCE_RCCL=1 timeout -s KILL 200 python -X faulthandler bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --frags 2000000 > gpurun_out/c25_rccl.json 2> gpurun_out/c25_rccl.err; echo "exit $?"
tail -30 gpurun_out/c25_rccl.err; tail -c 600 gpurun_out/c25_rccl.json
echo "---- diag_rccl"
timeout -s KILL 300 python tools/diag_rccl.py > gpurun_out/c25_diag_rccl.log 2>&1; tail -40 gpurun_out/c25_diag_rccl.log
