#!/bin/bash
# round 5, call 1: in-tick duplicate elimination in tick_kernel_rf + 208-entry stash: GPU suite, then A/B against the round-4 kernel
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5b/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5b/pytest.log
timeout 600 python tools/ab.py --fanout-model krandomnodes --ticks 320 --rounds 2 serf_amd/csrc/libserf_sim_base.so serf_amd/csrc/libserf_sim.so > gpurun_out/r5b/ab.log 2>&1; tail -6 gpurun_out/r5b/ab.log
