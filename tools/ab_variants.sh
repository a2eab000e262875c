#!/bin/bash
# Runs ON THE GPU BOX: bench.py phases for the default library and the variants named (tools/build_variant.sh).
#   tools/ab_variants.sh <tag> <config> name1 name2 ...   ->  gpurun_out/ab_<tag>.txt
tag=$1; cfg=$2; shift 2
out=gpurun_out/ab_$tag.txt
: > $out
for v in default "$@"; do
  lib=genrich_amd/libgenrich_amd.so
  [ $v != default ] && lib=genrich_amd/libgenrich_amd_$v.so
  GENRICH_AMD_LIB=$lib timeout 300 python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu --no-e2e 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), json.dumps(d.get('phases_ms')))" >> $out 2>&1
done
cat $out
