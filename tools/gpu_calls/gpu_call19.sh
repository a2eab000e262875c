cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/c19_pytest.log 2>&1
grep -E "passed|failed|^FAILED" gpurun_out/c19_pytest.log | tail -8
