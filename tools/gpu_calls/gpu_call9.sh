cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GX_DEBUG_SORT=1 timeout -s KILL 300 python tools/diag_sort.py 2>&1 | grep -E "^run|gave up|status 32" > gpurun_out/c9_diag.log 2>&1
cat gpurun_out/c9_diag.log
