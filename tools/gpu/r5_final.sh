#!/bin/bash
# round 5: the whole GPU suite, then the measurement call (tools/gpu/r5_measure.sh)
mkdir -p gpurun_out/r5m
timeout 1200 python -m pytest tests -m gpu -q -s > gpurun_out/r5m/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r5m/pytest.log
grep -E "bit-exact" gpurun_out/r5m/pytest.log | cut -c1-200
EXTRA_PMC=1 bash tools/gpu/r5_measure.sh
