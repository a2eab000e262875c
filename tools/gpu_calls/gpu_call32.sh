cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for m in 0 8388608 67108864; do
GX_HPEAKS_MIN=$m timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_emul8e_$m/trace -o emul -- python tools/emulate_ranks.py 8 > gpurun_out/c32_emul_$m.log 2>&1
echo "== hpmin $m"; python tools/trace_timeline.py gpurun_out/prof_emul8e_$m/trace k_sort1 -2 | grep -E "k_peaks_write|k_peak_short|k_mail"
done
