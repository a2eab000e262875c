#!/bin/bash
# Runs ON THE GPU BOX (under gpurun): kernel trace + PMC passes of `bench.py --config C`, each in its own
# rocprofv3 run (counters never share a run with --stats / other trace domains), into gpurun_out/prof_<tag>/.
#   tools/profile_round.sh <tag> [config]      e.g. tools/profile_round.sh r02a 2     (GX_PROF_PASSES="trace rdsz": only those passes)
# Afterwards (anywhere): python tools/make_counters_json.py <tag> [config]  ->  profiles/
set -u
tag=$1; cfg=${2:-2}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
bcfg="--config $cfg"
[ "$cfg" = "2q" ] && bcfg="--config 2 --qval"   # (the headline with -q: the p + q scan)
cmd="python bench.py $bcfg --steps 2 --warmup 1 --no-cpu --no-e2e --no-materialised"
export GX_ROCTX=1   # the library brackets its phases with roctx ranges: the trace pass attributes every kernel to its phase
passes=" ${GX_PROF_PASSES:-trace fetch write rdsz wrsz sq1 sq2} "
run() {  # name, rocprofv3 options...
  local name=$1; shift
  [[ "$passes" == *" $name "* ]] || return 0
  timeout -s KILL 300 rocprofv3 "$@" -d $out/$name -o bench -- $cmd > $out/$name.log 2>&1
  echo "$name rc=$?"
}
run trace --kernel-trace --marker-trace --stats
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
# the L2's memory-side read requests by size (32 / 64 / 128 bytes): the bytes a kernel fetches without FETCH_SIZE's blanket
# factor (MI355X_MICROARCH.md: FETCH_SIZE tallies a 128-byte request at 64); the write side likewise
run rdsz --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run wrsz --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum
run sq1 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq2 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
ls -la $out
