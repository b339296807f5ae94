#!/bin/bash
# round 3, call 27: BASELINE configs[4] at 256 Ki nodes with the reference's background tasks all on — Reconnector (30 s) and
# gossip_to_the_dead_time (30 s) next to the Reaper, the QueueChecker, push-pull and recycling — 5 % churn, 1 % loss
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c27
mkdir -p $OUT
cd $ROOT
timeout 600 python tools/config4_run.py --nodes 262144 --churn-every 24 --down 150 --rumors 1000 --pkt-records 16 --reconnect-interval 150 --gossip-to-the-dead 150 --out $OUT/config4_256k_reconnector.json 2>&1 | grep -v amdgpu.ids | tail -3
