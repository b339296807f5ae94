#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c10
mkdir -p $OUT
cd $ROOT
# configs[4] at one-GPU scale, SWIM layer ON: 1 % loss, a node crashes and comes back every 8 ticks (more subjects than view slots: recycling), push-pull on
timeout 900 python tools/convergence_hist.py --rumors 1000 --every 8 --churn-every 16 --down 15 --view-slots 256 --recycle-interval 25 --probe-interval 5 --push-pull-interval 150 --loss 0.01 --out $OUT/r02_convergence_hist_swim_churn.json > $OUT/hist1.log 2>&1; tail -2 $OUT/hist1.log
