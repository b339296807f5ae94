#!/bin/bash
# round 5, call 2: second look (tails) + copies skipped in tick_kernel_rf: GPU suite, A/B against the round-4 kernel, per-tick series
mkdir -p gpurun_out/r5e
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/r5e/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5e/pytest.log
timeout 600 python tools/ab.py --fanout-model krandomnodes --ticks 320 --rounds 2 serf_amd/csrc/libserf_sim_base.so serf_amd/csrc/libserf_sim.so > gpurun_out/r5e/ab.log 2>&1; tail -6 gpurun_out/r5e/ab.log
timeout 400 python tools/tick_series.py gpurun_out/r5e/tick_series.json > gpurun_out/r5e/series.log 2>&1; tail -2 gpurun_out/r5e/series.log
