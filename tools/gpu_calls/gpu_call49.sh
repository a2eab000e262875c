cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run5() { tag=$1; shift; ( env "$@" timeout -s KILL 400 python bench.py --config 5 --steps 4 --warmup 2 --no-e2e --no-cpu ) > gpurun_out/c49_bench5_$tag.json 2> gpurun_out/c49_bench5_$tag.err; }
run5 dflt
run5 mnc8 GENRICH_AMD_LIB=genrich_amd/libgenrich_amd_mnc8.so
run5 mnc7 GENRICH_AMD_LIB=genrich_amd/libgenrich_amd_mnc7.so
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c49_bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["phases_ms"].items()}, d["config"]["peaks"])
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
