#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4nt
mkdir -p $OUT
cd $ROOT
SERF_RF_SYNC=1 python tools/ab.py --fanout-model krandomnodes --rounds 2 serf_amd/csrc/libserf_sim.so serf_amd/csrc/libserf_sim_nt.so 2>&1 | grep "round\|us_per_tick_median"
cd /tmp && export TMPDIR=/tmp
ARGS="--fanout-model krandomnodes --no-cpu-baseline --no-convergence --no-second-load --no-long-window --steps 20 --warmup 5"
for L in libserf_sim.so libserf_sim_nt.so; do
  SERF_SIM_LIB=$ROOT/serf_amd/csrc/$L timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/$L/pmc_1 -o p -- python $ROOT/bench.py $ARGS > $OUT/$L.log 2>&1
  (cd $ROOT && python tools/pmc_summary.py $OUT/$L tick_kernel 20 | python -c "import json,sys; d=json.load(sys.stdin); print('$L', d['counters'])")
done
