cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/c28_pytest.log 2>&1
grep -E "passed|failed|^FAILED|Error" gpurun_out/c28_pytest.log | tail -12
GENRICH_AMD_LIB=genrich_amd/libgenrich_amd_pk128.so timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 -x -k "not fullsize" > gpurun_out/c28_pytest_pk128.log 2>&1
grep -E "passed|failed|^FAILED|Error" gpurun_out/c28_pytest_pk128.log | tail -5
run() { tag=$1; shift; ( env "$@" timeout -s KILL 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-e2e ) > gpurun_out/c28_bench_$tag.json 2> gpurun_out/c28_bench_$tag.err; }
run dflt
for v in pk64 pk128 pk256; do run $v GENRICH_AMD_LIB=genrich_amd/libgenrich_amd_$v.so; done
run5() { tag=$1; shift; ( env "$@" timeout -s KILL 400 python bench.py --config 5 --steps 4 --warmup 2 --no-e2e --no-cpu ) > gpurun_out/c28_bench5_$tag.json 2> gpurun_out/c28_bench5_$tag.err; }
run5 dflt
run5 mn512c8 GENRICH_AMD_LIB=genrich_amd/libgenrich_amd_mn512c8.so
run5 mn256c8 GENRICH_AMD_LIB=genrich_amd/libgenrich_amd_mn256c8.so
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c28_bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["phases_ms"].items()}, d["config"]["peaks"], d.get("gate",{}).get("passed"))
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
echo "---- emulate ranks"
for v in "" _pk128; do GENRICH_AMD_LIB=genrich_amd/libgenrich_amd$v.so timeout -s KILL 300 python tools/emulate_ranks.py 8 2>&1 | grep -v amdgpu.ids | tail -1; done
