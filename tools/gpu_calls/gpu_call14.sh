cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for pad in 0 4096 69632 2101248 1048576; do ( GX_ALLOC_PAD=$pad timeout -s KILL 200 python bench.py --steps 20 --warmup 3 --no-cpu --no-e2e ) > gpurun_out/c14_bench2_$pad.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c14_bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["phases_ms"].items()})
    except Exception as e: print(f, "ERR", e)
PY
