#!/bin/bash
# round 5, call 5: the pack's runs and the sort one tick further ahead on shards: the sharded tests, then a kernel trace of one rank of the packed exchange over RCCL
mkdir -p gpurun_out/r5h
timeout 600 python -m pytest tests -m gpu -q -x -k "four_shards_on_one_gpu or rccl" > gpurun_out/r5h/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5h/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5h/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --force-sharded --fanout-model krandomnodes --exchange rccl --steps 20 --warmup 5 --no-cpu-baseline --no-second-load --no-convergence --no-long-window > $GRAFT_REPO_ROOT/gpurun_out/r5h/trace.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
grep '"metric"' gpurun_out/r5h/trace.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.3e'%d['value'],'ms/step',d['ms_per_step'],'kernel',d['roofline']['kernel_ms'],d['exchange']['exchange_ms'],d['exchange']['serial_ms_per_step'])"
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/r5h/trace/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
find gpurun_out/r5h -name '*kernel_trace.csv' -size +20M -delete
