cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 600 python -m pytest tests/test_hip_multirank.py tests/test_hip_parity.py -m gpu -q --timeout 300 -k "rccl or calc_pval or fisher_vs or closed_form or crowded or saturation or end_before or atac_geometry" -s > gpurun_out/c2_pytest.log 2>&1
( timeout -s KILL 400 python bench.py --config 3 --steps 5 --warmup 2 --no-cpu --no-e2e ) > gpurun_out/c2_bench3.json 2> gpurun_out/c2_bench3.err
tail -3 gpurun_out/c2_pytest.log
