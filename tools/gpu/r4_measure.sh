#!/bin/bash
# round 4, measurement call: the driver's bench line (both fan-out models, CPU baseline, parity at both ends of the timed
# region), the default bench line, and for EACH model the rocprofv3 kernel trace and the PMC passes the traffic figures come
# from (separate passes, no other trace domain next to --pmc; --no-long-window: the last 20 tick launches are the timed ones)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4m
mkdir -p $OUT
cd $ROOT
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err; echo "bench (driver args) rc=$?"
timeout 400 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
timeout 400 python bench.py --gpus 1 --force-sharded --exchange rccl --chunks 2 --steps 20 --warmup 5 --no-cpu-baseline --no-second-load > $OUT/bench_one_rank_rccl.json 2> $OUT/bench_one_rank_rccl.err; echo "bench one rank over RCCL rc=$?"
timeout 400 python bench.py --gpus 1 --force-sharded --fanout-model krandomnodes --exchange rccl --steps 20 --warmup 5 --no-cpu-baseline --no-second-load > $OUT/bench_one_rank_rccl_krandomnodes.json 2> $OUT/bench_one_rank_rccl_krandomnodes.err; echo "bench one rank, random fan-out, over RCCL rc=$?"
cd /tmp && export TMPDIR=/tmp
for M in krandomnodes bijection; do
  ARGS="--fanout-model $M --no-cpu-baseline --no-convergence --no-second-load --no-long-window --steps 20 --warmup 5"
  mkdir -p $OUT/$M
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$M/trace -o t -- python $ROOT/bench.py $ARGS > $OUT/$M/trace.log 2>&1; echo "$M trace rc=$?"
  grep '"metric"' $OUT/$M/trace.log > $OUT/$M/bench_traced.json
  i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/$M/pmc_$i -o p -- python $ROOT/bench.py $ARGS > $OUT/$M/pmc_$i.log 2>&1 || echo "$M pmc pass $i failed: $PMC"
  done
  (cd $ROOT && python tools/pmc_summary.py $OUT/$M tick_kernel 20 > $OUT/$M/tick_kernel_pmc.json)
done
cd $ROOT
python - <<PY
import json
for f in ('bench_20_5','bench_default','bench_one_rank_rccl','bench_one_rank_rccl_krandomnodes'):
    try:
        d=json.loads(open('$OUT/%s.json'%f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f, d['config'].get('fanout_model'), 'value %.3e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%r['kernel_ms'], 'frac %.3f'%r['frac'], 'drops', d['config']['model_bound_drops'],
              'parity', d.get('parity', {}).get('digest_match'), {k: ('%.3e'%v['value'], '%.4f'%v['kernel_ms']) for k,v in d.get('fanout_models',{}).items()}, d.get('distributed'))
    except Exception as e:
        print(f, 'unreadable', e)
for m in ('krandomnodes','bijection'):
    p=json.load(open('$OUT/%s/tick_kernel_pmc.json'%m))
    print(m, 'profiled kernel us', p.get('kernel_us_mean'), 'hbm bytes/launch', p.get('hbm_bytes_per_launch'), {k: round(v) for k, v in p['counters'].items()})
PY
