#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + separate PMC passes for bench.py.
# usage: tools/profile_gpu.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/{trace,pmc_*}/...
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-convergence $*"   # bench.py defaults: 300 timed ticks after 60 warm-up ticks
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
grep '"metric"' $OUT/trace.log > $OUT/bench_traced.json
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_$i -o p -- python $ROOT/bench.py $ARGS > $OUT/pmc_$i.log 2>&1 || echo "pmc pass $i failed: $PMC" >> $OUT/errors.txt
done
find $OUT -name '*.csv' | head -30
