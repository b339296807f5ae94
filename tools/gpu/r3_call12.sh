#!/bin/bash
# round 3, call 12: BASELINE configs[4] with the SWIM layer on, 16-record packets, join sync: 5 % churn at 256 Ki nodes
# (inside the model's capacity), and 1 Mi nodes with a fifth of the churn (what the 16-slot queue does in the Poisson tail)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c12
mkdir -p $OUT
cd $ROOT
timeout 400 python tools/config4_run.py --nodes 262144 --churn-every 24 --down 150 --rumors 1000 --pkt-records 16 --out $OUT/config4_256k.json 2>&1 | grep -v amdgpu.ids | tail -2
timeout 500 python tools/config4_run.py --nodes 1048576 --churn-frac 0.01 --churn-every 20 --down 160 --rumors 1000 --pkt-records 16 --out $OUT/config4_1m_fifth.json 2>&1 | grep -v amdgpu.ids | tail -2
