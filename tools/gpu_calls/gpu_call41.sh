cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 600 python -m pytest tests/test_hip_parity.py -m gpu -q --timeout 600 -x -k "guess_is_too_small" --tb=short 2>&1 | tail -40
