#!/bin/bash
# round 3, call 33: memberlist's TCP fallback ping and nack accounting (SIM_CF_TCP_FALLBACK, SIM_CF_NACKS) — the GPU sweeps that
# draw them, then BASELINE configs[4] at 1 Mi nodes with the fallback on (no false suspicions from packet loss)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c33
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "random_configurations or random_fanout or memberlist_flags or backend_is_hip" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $OUT/pytest.log
timeout 900 python tools/config4_run.py --nodes 1048576 --churn-frac 0.01 --churn-every 20 --down 160 --rumors 1000 --pkt-records 16 --tcp-fallback --nacks --reconnect-interval 150 --gossip-to-the-dead 150 --out $OUT/config4_1m_churn1_tcp.json 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-900
