#!/bin/bash
# round 4, call 8: where the balanced RF kernel spends its time (timing build) + PMC counters
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4c8
mkdir -p $OUT
cd $ROOT
SERF_RF_SYNC=1 timeout 300 python tools/tick_timing.py 1048576 --random-fanout > $OUT/timing_rf.txt 2>&1
cat $OUT/timing_rf.txt
B="--random-fanout --steps 100 --warmup 20 --no-second-load --no-cpu-baseline --no-convergence"
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  i=$((i+1))
  SERF_RF_SYNC=1 timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_$i -o p -- python $ROOT/bench.py $B > $OUT/pmc_$i.log 2>&1 || echo "pmc pass $i failed"
done
SERF_RF_SYNC=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py $B > $OUT/trace.log 2>&1
cd $ROOT
python tools/pmc_summary.py $OUT tick_kernel 100 > $OUT/pmc_summary.json
python - <<PY
import json
d=json.load(open("$OUT/pmc_summary.json"))
c=d["counters"]
print({k: round(v) for k,v in c.items()})
print("kernel_us_mean", d.get("kernel_us_mean"), "hbm bytes/launch", d.get("hbm_bytes_per_launch"), "read", d.get("hbm_read_bytes"), "write", d.get("hbm_write_bytes"))
PY
