cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 300 python tools/diag_pval.py > gpurun_out/c3_diag.log 2>&1
tail -5 gpurun_out/c3_diag.log
