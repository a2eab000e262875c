cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout -s KILL 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -40 ) > gpurun_out/c1_pytest.log 2>&1
( timeout -s KILL 300 python bench.py --steps 10 --warmup 3 ) > gpurun_out/c1_bench2.json 2> gpurun_out/c1_bench2.err
( GENRICH_AMD_LIB=genrich_amd/libgenrich_amd_ept64.so timeout -s KILL 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-e2e ) > gpurun_out/c1_bench2_ept64.json 2> gpurun_out/c1_bench2_ept64.err
timeout -s KILL 900 tools/profile_round.sh r02a 2 > gpurun_out/c1_prof.log 2>&1
( timeout -s KILL 400 python bench.py --config 3 --steps 5 --warmup 2 ) > gpurun_out/c1_bench3.json 2> gpurun_out/c1_bench3.err
( timeout -s KILL 400 python bench.py --config 4 --steps 5 --warmup 2 ) > gpurun_out/c1_bench4.json 2> gpurun_out/c1_bench4.err
( timeout -s KILL 500 python bench.py --config 5 --steps 5 --warmup 2 ) > gpurun_out/c1_bench5.json 2> gpurun_out/c1_bench5.err
tail -5 gpurun_out/c1_pytest.log
