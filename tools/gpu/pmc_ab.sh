#!/bin/bash
# usage: tools/gpu/pmc_ab.sh <tag> <lib.so> [bench args]: kernel trace + FETCH/WRITE/request counters of bench.py with that build
set -u
TAG=$1; LIB=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export SERF_SIM_LIB=$ROOT/$LIB
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-convergence --steps 100 --warmup 20 $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
grep '"metric"' $OUT/trace.log > $OUT/bench_traced.json
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_$i -o p -- python $ROOT/bench.py $ARGS > $OUT/pmc_$i.log 2>&1 || echo "pmc pass $i failed: $PMC" >> $OUT/errors.txt
done
python $ROOT/tools/pmc_summary.py $OUT tick_kernel 100 > $OUT/summary.json
python -c "
import json; d=json.load(open('$OUT/summary.json')); c=d['counters']
print('$TAG', 'us', round(d.get('kernel_us_mean',0),1), 'FETCHx2 MB', round(c.get('FETCH_SIZE',0)*2048/1e6,1), 'WRITE MB', round(c.get('WRITE_SIZE',0)*1024/1e6,1), 'RDREQ M', round(c.get('TCC_EA0_RDREQ_sum',0)/1e6,2), 'WRREQ M', round(c.get('TCC_EA0_WRREQ_sum',0)/1e6,2), 'hit', round(c.get('TCC_HIT_sum',0)/1e6,2), 'miss', round(c.get('TCC_MISS_sum',0)/1e6,2), 'VALU/wave', round(c.get('SQ_INSTS_VALU',0)/max(c.get('SQ_WAVES',1),1)), 'SALU/wave', round(c.get('SQ_INSTS_SALU',0)/max(c.get('SQ_WAVES',1),1)), 'wait_any', round(c.get('SQ_WAIT_ANY',0)/max(c.get('SQ_WAVE_CYCLES',1),1),3))
"
