#!/bin/bash
# round 4: where rf_scatter's 48 us went (a record of the experiment: the variants were built with compile-time switches —
# -DRF_SPW / -DRFB / -DRF_ABLATE — that the source no longer has; results in profiles/r04_experiments.md).  Variant builds:
#   x_spw8k  8192 senders per workgroup (half the returning atomics, half the workgroups)
#   x_spw2k  2048 senders / 512 threads per workgroup (twice the atomics, twice the workgroups)
#   x_ab1    no global atomics (positions made up), x_ab2 no stores, x_ab3 neither
# then the level-1 bucket size / entry width of the shipped build (environment switches)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4sc
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_host_paths_gpu.py tests/test_third_model.py -m gpu -x -q 2>&1 | tail -3
B="--fanout-model krandomnodes --steps 20 --warmup 5 --no-second-load --no-cpu-baseline --no-convergence --no-long-window"
cd /tmp && export TMPDIR=/tmp
one() {  # name, env...
  local name=$1; shift
  env SERF_RF_SYNC=1 "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o t -- python $ROOT/bench.py $B > $OUT/$name.log 2>&1
  echo "== $name"; grep '"metric"' $OUT/$name.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.3e' % d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'parity', d.get('parity', {}).get('digest_match'))"
  grep "rf_\|tick_kernel" $OUT/$name/t_kernel_stats.csv | awk -F, '{print substr($1,1,40), $2, $4}'
}
one base
for V in spw8k spw2k ab1 ab2 ab3; do one $V SERF_SIM_LIB=$ROOT/serf_amd/csrc/libserf_sim_x_$V.so; done
one lb10 SERF_RF_LB=10
one lb9 SERF_RF_LB=9
one lb11w SERF_RF_WIDE=1
