#!/bin/bash
# round 3, measurement call: the bench lines (driver arguments with the CPU baseline and the parity block; defaults), the
# rocprofv3 kernel trace of the driver's command, and the PMC passes the traffic figure comes from (separate passes, no
# other trace domain next to --pmc)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3m
mkdir -p $OUT
cd $ROOT
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err; echo "bench (driver args) rc=$?"
timeout 200 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-convergence --no-second-load --steps 20 --warmup 5"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1; echo "trace rc=$?"
grep '"metric"' $OUT/trace.log > $OUT/bench_traced.json
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_$i -o p -- python $ROOT/bench.py $ARGS > $OUT/pmc_$i.log 2>&1 || echo "pmc pass $i failed: $PMC"
done
cd $ROOT
python tools/pmc_summary.py $OUT tick_kernel 20 > $OUT/tick_kernel_pmc.json
python - <<PY
import json
for f in ('bench_20_5','bench_default','bench_traced'):
    try:
        d=json.loads(open('$OUT/%s.json'%f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f, 'value %.3e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%r['kernel_ms'], 'frac %.3f'%r['frac'], 'measured', (r.get('measured') or {}).get('frac'), 'drops', d['config']['model_bound_drops'],
              'parity', d.get('parity', {}).get('digest_match'), 'second', {k: d['second_load'][k] for k in ('value','kernel_ms','model_bound_drops')} if 'second_load' in d else None,
              'rounds', {k: d['rounds_to_99'][k] for k in ('median','p90','max','n')} if d.get('rounds_to_99') else None)
    except Exception as e:
        print(f, 'unreadable', e)
p=json.load(open('$OUT/tick_kernel_pmc.json'))
print('profiled kernel us', p.get('kernel_us_mean'), 'hbm bytes/launch', p.get('hbm_bytes_per_launch'), {k: round(v) for k, v in p['counters'].items()})
PY
