#!/bin/bash
# instruction-cache behaviour of the two tick kernels (the RF one is 19 200 instructions long)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4ic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i "icache\|ifetch\|SQ_INST_LEVEL\|SQ_WAIT_INST\|SQC_" | head -40 > $OUT/avail.txt
cat $OUT/avail.txt | cut -c1-160
for M in krandomnodes bijection; do
  ARGS="--fanout-model $M --no-cpu-baseline --no-convergence --no-second-load --no-long-window --steps 20 --warmup 5"
  i=0
  for PMC in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"; do
    i=$((i+1))
    mkdir -p $OUT/$M
    timeout 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/$M/pmc_$i -o p -- python $ROOT/bench.py $ARGS > $OUT/$M/pmc_$i.log 2>&1 || echo "$M pmc pass $i failed: $PMC"
  done
  (cd $ROOT && python tools/pmc_summary.py $OUT/$M tick_kernel 20) | python -c "import json,sys; d=json.load(sys.stdin); print('$M', d.get('kernel_us_mean'), {k: round(v) for k,v in d['counters'].items()})"
done
