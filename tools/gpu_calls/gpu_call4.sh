cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 600 python tools/diag_rccl.py > gpurun_out/c4_rccl.log 2>&1
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/c4_pytest.log 2>&1
tail -15 gpurun_out/c4_pytest.log
