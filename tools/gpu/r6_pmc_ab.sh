#!/bin/bash
# round 6: SQ / traffic counters of tick_kernel_rf, one handle against one rank of the sharded path (why is the same kernel slower there?)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in plain sharded; do
  ARGS="--fanout-model krandomnodes --no-cpu-baseline --no-convergence --no-second-load --no-long-window --steps 20 --warmup 5"
  [ $V = sharded ] && ARGS="$ARGS --force-sharded --exchange rccl --chunks 1"
  D=$OUT/$V
  mkdir -p $D
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o t -- python $ROOT/bench.py $ARGS > $D/trace.log 2>&1; echo "$V trace rc=$?"
  i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $D/pmc_$i -o p -- python $ROOT/bench.py $ARGS > $D/pmc_$i.log 2>&1 || echo "$V pmc pass $i failed: $PMC"
  done
  (cd $ROOT && python tools/pmc_summary.py $D tick_kernel 20 > $D/tick_kernel_pmc.json)
  find $D -name '*counter_collection.csv' -delete; find $D -name '*kernel_trace.csv' -delete
  python - <<PY
import json
q=json.load(open('$D/tick_kernel_pmc.json'))
print('$V', 'kernel us', q.get('kernel_us_mean'), 'hbm bytes/launch', q.get('hbm_bytes_per_launch'))
for k,v in sorted(q['counters'].items()): print('   ', k, '%.4g'%v)
PY
done
