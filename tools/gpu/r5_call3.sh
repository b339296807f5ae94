#!/bin/bash
# round 5, call 3: stash size A/B (128 vs 208 entries: 8 vs 10 KiB of LDS per wave), bench.py end to end (both models), one rank of the packed exchange over RCCL
mkdir -p gpurun_out/r5f
timeout 600 python tools/ab.py --fanout-model krandomnodes --ticks 320 --rounds 2 serf_amd/csrc/libserf_sim_base.so serf_amd/csrc/libserf_sim_s128.so serf_amd/csrc/libserf_sim.so > gpurun_out/r5f/ab.log 2>&1; tail -8 gpurun_out/r5f/ab.log | cut -c1-400
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5f/bench.json 2> gpurun_out/r5f/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r5f/bench.err
timeout 400 python bench.py --gpus 1 --force-sharded --fanout-model krandomnodes --exchange rccl --steps 20 --warmup 5 --no-cpu-baseline --no-second-load > gpurun_out/r5f/bench_rccl_rf.json 2> gpurun_out/r5f/bench_rccl_rf.err; echo "bench one-rank rccl rf rc=$?"; tail -3 gpurun_out/r5f/bench_rccl_rf.err
python - <<PY
import json
for f in ('bench','bench_rccl_rf'):
    try:
        d=json.loads(open('gpurun_out/r5f/%s.json'%f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f, 'value %.3e'%d['value'], 'long %.3e'%d.get('value_long_window',0), 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%r['kernel_ms'], 'long kernel', d.get('long_window',{}).get('kernel_ms'), {k: ('%.3e'%v['value'], '%.4f'%v['kernel_ms']) for k,v in d.get('fanout_models',{}).items()}, d.get('exchange'))
    except Exception as e:
        print(f,'unreadable',e)
PY
