cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_c16
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c16/trace -o bench -- python bench.py --steps 4 --warmup 1 --no-cpu --no-e2e > gpurun_out/prof_c16/trace.log 2>&1
python profiles/summarize_rocpd.py $(find gpurun_out/prof_c16/trace -name "*.db" | head -1) | head -45
