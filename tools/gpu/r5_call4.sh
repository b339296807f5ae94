#!/bin/bash
# round 5, call 4: second look only for records that can be retired, the pack with the cell asked for on spec: the tests of the paths touched, A/B, the one-rank packed exchange over RCCL
mkdir -p gpurun_out/r5g
timeout 900 python -m pytest tests -m gpu -q -x -k "random_fanout or rccl or decoder_rules or kRandomNodes or graph_build or third_model or at_size or 4mi or krandomnodes" > gpurun_out/r5g/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5g/pytest.log
timeout 600 python tools/ab.py --fanout-model krandomnodes --ticks 320 --rounds 2 serf_amd/csrc/libserf_sim_base.so serf_amd/csrc/libserf_sim_s128.so serf_amd/csrc/libserf_sim.so > gpurun_out/r5g/ab.log 2>&1; tail -8 gpurun_out/r5g/ab.log | cut -c1-300
timeout 400 python bench.py --gpus 1 --force-sharded --fanout-model krandomnodes --exchange rccl --steps 20 --warmup 5 --no-cpu-baseline --no-second-load > gpurun_out/r5g/bench_rccl_rf.json 2> gpurun_out/r5g/bench_rccl_rf.err; echo "bench one-rank rccl rf rc=$?"
python - <<PY
import json
for f in ('bench_rccl_rf',):
    try:
        d=json.loads(open('gpurun_out/r5g/%s.json'%f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f, 'value %.3e'%d['value'], 'long %.3e'%d.get('value_long_window',0), 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%r['kernel_ms'], {k:d['exchange'][k] for k in ('exchange_ms','serial_ms_per_step','overlapped_ms_per_step')})
    except Exception as e:
        print(f,'unreadable',e)
PY
