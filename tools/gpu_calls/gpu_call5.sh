cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py -m gpu -q --timeout 600 -x > gpurun_out/c5_pytest.log 2>&1
tail -4 gpurun_out/c5_pytest.log
( timeout -s KILL 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-e2e ) > gpurun_out/c5_bench_fast.json 2> gpurun_out/c5_bench_fast.err
( GX_TILE_OLD=1 timeout -s KILL 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-e2e ) > gpurun_out/c5_bench_old.json 2> gpurun_out/c5_bench_old.err
for w in 8 12 15; do ( GX_TILE_FAST_WG=$w timeout -s KILL 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-e2e ) > gpurun_out/c5_bench_fast_wg$w.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c5_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["phases_ms"].items()}, d["config"]["peaks"])
    except Exception as e: print(f, "ERR", e)
PY
