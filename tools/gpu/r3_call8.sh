#!/bin/bash
# round 3, call 8: GPU suite on the SIM_S = 16 / deferred-suspicion build, then BASELINE configs[4] with the SWIM layer on:
# first at 64 Ki nodes (seconds), then at 1 Mi nodes (5 % of the nodes = 52 428 crash, are declared failed, re-join; 1 % loss)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c8
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log | cut -c1-200
timeout 200 python tools/config4_run.py --nodes 65536 --churn-every 20 --down 130 --rumors 200 --out $OUT/config4_64k.json 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python tools/config4_run.py --nodes 1048576 --churn-every 24 --down 160 --rumors 1000 --out $OUT/config4_1m.json 2>&1 | grep -v amdgpu.ids | tail -2
