#!/bin/bash
# round 2, call 22: measurements of the sender-kept-packets kernel: GPU tests, smoke, bench (default and the driver's
# arguments), rocprofv3 stats + PMC with the driver's arguments, timing and ablation builds
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c22
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err; echo "bench20 rc=$?"
bash tools/profile_gpu.sh r02d --steps 20 --warmup 5 > $OUT/profile.log 2>&1; echo "profile rc=$?"
python tools/pmc_summary.py gpurun_out/prof_r02d tick_kernel 20 > $OUT/r02_tick_kernel_pmc.json
python tools/pmc_summary.py gpurun_out/prof_r02d/mem tick_kernel 20 > $OUT/r02_tick_kernel_pmc_memory_path.json
timeout 300 python tools/tick_timing.py > $OUT/tick_timing.txt 2>&1
timeout 900 python tools/ablate.py > $OUT/ablation.txt 2>&1
python -c "
import json
for f in ('bench_default','bench_20_5'):
    d=json.load(open('$OUT/%s.json'%f)); r=d['roofline']
    print(f, 'value %.3e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f [%.4f, %.4f]'%(r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_max']), 'frac %.3f'%r['frac'], 'drops', d['config']['model_bound_drops'], 'cpu', d.get('cpu_baseline',{}).get('value'))
d=json.load(open('$OUT/r02_tick_kernel_pmc.json')); print('profiled us', d.get('kernel_us_mean'), 'bytes', d.get('hbm_bytes_per_launch'), d.get('hbm_read_bytes'), d.get('hbm_write_bytes'))
"
