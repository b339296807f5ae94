#!/bin/bash
# round 3, call 36: rounds-to-99 % of 1 000 rumours at 1 Mi nodes under 1 % loss with memberlist's literal kRandomNodes ON THE GPU
# (round 2 had this histogram from the CPU oracle only), next to the bijection on the same schedule
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c36
mkdir -p $OUT
cd $ROOT
for m in rf bij; do
  F=""; [ $m = rf ] && F="--random-fanout"
  timeout 700 python tools/config4_run.py --nodes 1048576 --churn-frac 0.002 --churn-every 20 --down 160 --rumors 1000 --tcp-fallback --nacks $F --out $OUT/conv_1m_$m.json 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-500
done
