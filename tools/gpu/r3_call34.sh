#!/bin/bash
# round 3, call 34: BASELINE configs[4] at 1 Mi nodes in full — 5 % churn (52 428 nodes crash, are declared failed, re-join),
# 1 % packet loss, memberlist's TCP fallback ping and nacks, Reconnector, gossip_to_the_dead, 16-record packets
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c34
mkdir -p $OUT
cd $ROOT
timeout 1500 python tools/config4_run.py --nodes 1048576 --churn-frac 0.05 --churn-every 12 --down 160 --rumors 1000 --pkt-records 16 --tcp-fallback --nacks --reconnect-interval 150 --gossip-to-the-dead 150 --out $OUT/config4_1m_churn5_tcp.json 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-900
