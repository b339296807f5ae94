#!/bin/bash
# round 5, call 6: sender chunks of the packed exchange: the sharded tests, one rank over RCCL with 1 / 2 / 4 chunks
mkdir -p gpurun_out/r5i
timeout 900 python -m pytest tests -m gpu -q -x -k "four_shards_on_one_gpu or rccl" > gpurun_out/r5i/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5i/pytest.log
for C in 1 2 4; do
timeout 400 python bench.py --gpus 1 --force-sharded --fanout-model krandomnodes --exchange rccl --chunks $C --steps 20 --warmup 5 --no-cpu-baseline --no-second-load --no-convergence > gpurun_out/r5i/bench_rccl_rf_c$C.json 2> gpurun_out/r5i/bench_rccl_rf_c$C.err; echo "bench chunks $C rc=$?"
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r5i/bench_rccl_rf_c$C.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('chunks $C', 'value %.3e'%d['value'], 'long %.3e'%d.get('value_long_window',0), 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%r['kernel_ms'], {k:d['exchange'][k] for k in ('exchange_ms','serial_ms_per_step','overlapped_ms_per_step','bytes_per_gpu_per_tick')})
except Exception as e:
    print('unreadable',e); print(open('gpurun_out/r5i/bench_rccl_rf_c$C.err').read()[-1500:])
PY
done
