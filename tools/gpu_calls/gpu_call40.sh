cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -s KILL 600 python -m pytest tests/test_hip_parity.py -m gpu -q --timeout 600 -x -k "chunked_bh or guess_is_too_small or phase_timers" 2>&1 | tail -5
timeout -s KILL 500 python tools/emulate_ranks.py 1 2 4 8 8:3 8:7 2>&1 | grep -v amdgpu.ids | grep " ms "
export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_emul8g/trace -o emul -- python tools/emulate_ranks.py 8 > gpurun_out/c40_emul.log 2>&1
python tools/trace_timeline.py gpurun_out/prof_emul8g/trace k_sort1 -2
