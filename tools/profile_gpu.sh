#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + separate PMC passes for bench.py.
# usage: tools/profile_gpu.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/{trace,pmc_*}/...
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-convergence $*"   # bench.py defaults: 320 pre-roll ticks, then 300 timed ticks after 60 warm-up ticks
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
grep '"metric"' $OUT/trace.log > $OUT/bench_traced.json
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_$i -o p -- python $ROOT/bench.py $ARGS > $OUT/pmc_$i.log 2>&1 || echo "pmc pass $i failed: $PMC" >> $OUT/errors.txt
done
# second counter set: the memory path (instruction cache, TCP<->TCC requests and latencies, TA/TCC busy, TCP and
# UTCL1 stalls) -> prof_<tag>/mem/pmc_*; summarise with tools/pmc_summary.py gpurun_out/prof_<tag>/mem tick_kernel 300
mkdir -p $OUT/mem
i=0
for PMC in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES" \
 "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
 "TCC_REQ_sum TCC_BUSY_sum TCC_TAG_STALL_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
 "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
 "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_LFIFO_FULL_sum" \
 "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" \
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/mem/pmc_$i -o p -- python $ROOT/bench.py $ARGS > $OUT/mem/pmc_$i.log 2>&1 || echo "mem pmc pass $i failed: $PMC" >> $OUT/errors.txt
done
find $OUT -name '*.csv' | head -30
