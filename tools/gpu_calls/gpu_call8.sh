cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "" _poll1 _poll2; do
  echo "=== variant '$v'"
  GENRICH_AMD_LIB=genrich_amd/libgenrich_amd$v.so timeout -s KILL 300 python tools/diag_sort.py 2>&1 | grep -E "^run|^events"
done > gpurun_out/c8_diag.log 2>&1
cat gpurun_out/c8_diag.log
