#!/bin/bash
# round 3, call 9: GPU suite; configs[4] with 16-record packets: 64 Ki nodes, then a tenth of the 1 Mi run as a check
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c9
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log | cut -c1-200
timeout 200 python tools/config4_run.py --nodes 65536 --churn-every 20 --down 130 --rumors 200 --pkt-records 16 --out $OUT/config4_64k.json 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python tools/config4_run.py --nodes 1048576 --churn-frac 0.005 --churn-every 20 --down 160 --rumors 100 --pkt-records 16 --out $OUT/config4_1m_tenth.json 2>&1 | grep -v amdgpu.ids | tail -2
