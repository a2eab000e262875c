cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GX_QT_MULTI=1 timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/c27_pytest.log 2>&1
grep -E "passed|failed|^FAILED|Error" gpurun_out/c27_pytest.log | tail -12
run5() { tag=$1; shift; ( env "$@" timeout -s KILL 400 python bench.py --config 5 --steps 4 --warmup 2 --no-e2e --no-cpu ) > gpurun_out/c27_bench5_$tag.json 2> gpurun_out/c27_bench5_$tag.err; }
run5 dflt
run5 mn512 GENRICH_AMD_LIB=genrich_amd/libgenrich_amd_mn512.so
run5 mn512c10 GENRICH_AMD_LIB=genrich_amd/libgenrich_amd_mn512c10.so
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c27_bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["phases_ms"].items()}, d["config"]["peaks"], d.get("gate",{}).get("passed"))
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
echo "---- emulate ranks under rocprof"
export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_emul8/trace -o emul -- python tools/emulate_ranks.py 8 > gpurun_out/c27_emul.log 2>&1
grep -v amdgpu.ids gpurun_out/c27_emul.log | tail -3
