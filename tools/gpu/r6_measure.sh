#!/bin/bash
# round 6, measurement call: the driver's bench line (both fan-out models, CPU baseline, parity at both ends of the timed region),
# the one-rank RCCL lines (both models: the packed slabs and the bijection's chunks), and for EACH model the rocprofv3 kernel trace
# and the PMC passes the traffic figures come from — over the driver's 20 timed launches AND over the 300 launches of the long
# window (ticks 345 .. 644: `--steps 300 --warmup 25 --no-long-window`, the very ticks bench.py's long_window times).
# Separate passes, no other trace domain next to --pmc.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6m
mkdir -p $OUT
cd $ROOT
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err; echo "bench (driver args) rc=$?"; cp bench_detail.json $OUT/bench_20_5_detail.json; wc -c $OUT/bench_20_5.json
timeout 400 python bench.py --gpus 1 --force-sharded --fanout-model krandomnodes --exchange rccl --chunks 1 --steps 20 --warmup 5 --no-cpu-baseline --no-second-load > $OUT/bench_one_rank_rccl_krandomnodes.json 2> $OUT/bench_one_rank_rccl_krandomnodes.err; echo "bench one rank, kRandomNodes packed slabs, over RCCL rc=$?"; cp bench_detail.json $OUT/bench_one_rank_rccl_krandomnodes_detail.json
timeout 400 python bench.py --gpus 1 --force-sharded --exchange rccl --chunks 2 --steps 20 --warmup 5 --no-cpu-baseline --no-second-load > $OUT/bench_one_rank_rccl.json 2> $OUT/bench_one_rank_rccl.err; echo "bench one rank, bijection, over RCCL rc=$?"; cp bench_detail.json $OUT/bench_one_rank_rccl_detail.json
cd /tmp && export TMPDIR=/tmp
for M in ${MODELS:-krandomnodes bijection}; do
  for W in short long; do
    if [ $W = short ]; then ARGS="--steps 20 --warmup 5"; K=20; else ARGS="--steps 300 --warmup 25"; K=300; fi
    ARGS="--fanout-model $M --no-cpu-baseline --no-convergence --no-second-load --no-long-window $ARGS"
    D=$OUT/${M}_$W
    mkdir -p $D
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o t -- python $ROOT/bench.py $ARGS > $D/trace.log 2>&1; echo "$M $W trace rc=$?"
    grep '"metric"' $D/trace.log | tail -1 > $D/bench_traced.json
    i=0
    for PMC in "FETCH_SIZE" "WRITE_SIZE" ${EXTRA_PMC:+"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"}; do
      i=$((i+1))
      [ $W = long ] && [ $i -gt 2 ] && continue
      timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $D/pmc_$i -o p -- python $ROOT/bench.py $ARGS > $D/pmc_$i.log 2>&1 || echo "$M $W pmc pass $i failed: $PMC"
    done
    (cd $ROOT && python tools/pmc_summary.py $D tick_kernel $K > $D/tick_kernel_pmc.json)
    # keep what is judged, drop the bulky per-dispatch CSVs (gpurun_out merges at most 64 MiB)
    find $D -name '*counter_collection.csv' -delete; find $D -name '*kernel_trace.csv' -delete
  done
done
# the second load (SURVEY 8d config 3 at the packet budget: 0.8 ops / tick, 16 records per packet): its own trace + traffic passes — the run is the
# headline's configuration with the second load's arguments, so that the LAST launches are the second load's timed ones
D=$OUT/second_load
mkdir -p $D
SARGS="--fanout-model krandomnodes --no-cpu-baseline --no-convergence --no-second-load --no-long-window --rate 0.8 --pkt-records 16 --ring-overflow 8 --preroll 160 --warmup 20 --steps 60"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -o t -- python $ROOT/bench.py $SARGS > $D/trace.log 2>&1; echo "second load trace rc=$?"
grep '"metric"' $D/trace.log | tail -1 > $D/bench_traced.json
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $D/pmc_$i -o p -- python $ROOT/bench.py $SARGS > $D/pmc_$i.log 2>&1 || echo "second load pmc pass $i failed: $PMC"
done
(cd $ROOT && python tools/pmc_summary.py $D tick_kernel 60 > $D/tick_kernel_pmc.json; python tools/pmc_summary.py $D deep_queue_kernel 60 > $D/deep_kernel_pmc.json)
find $D -name '*counter_collection.csv' -delete; find $D -name '*kernel_trace.csv' -delete
cd $ROOT
python - <<PY
import json
for f in ('bench_20_5','bench_one_rank_rccl_krandomnodes','bench_one_rank_rccl'):
    try:
        d=json.loads(open('$OUT/%s.json'%f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f, d['config'].get('fanout_model'), 'value %.3e'%d['value'], 'long %.3e'%d.get('value_long_window',0), 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%r['kernel_ms'], 'frac %.3f'%r['frac'], 'drops', d['config']['model_bound_drops'],
              'parity', d.get('parity', {}).get('digest_match'), {k: ('%.3e'%v['value'], '%.4f'%v['kernel_ms']) for k,v in d.get('fanout_models',{}).items()}, d.get('exchange',{}).get('exchange_ms'))
    except Exception as e:
        print(f, 'unreadable', e)
import glob
for p in sorted(glob.glob('$OUT/*/tick_kernel_pmc.json')):
    q=json.load(open(p))
    print(p.split('/')[-2], 'profiled kernel us', q.get('kernel_us_mean'), 'launches', q.get('launches'), 'hbm bytes/launch', q.get('hbm_bytes_per_launch'))
PY
