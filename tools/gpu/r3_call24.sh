#!/bin/bash
# round 3, call 24: the Reconnector on the GPU — its own test, both configuration sweeps (it is drawn into them), the suites
# next to it; then the kernel A/B against the build before it
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c24
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "reconnector or sharded_kernel_four or backend_is_hip" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest.log
V=serf_amd/csrc/variants
timeout 600 python tools/ab.py --ticks 120 --rounds 2 $V/base.so serf_amd/csrc/libserf_sim.so > $OUT/ab.log 2>&1; echo "ab rc=$?"
grep -v amdgpu.ids $OUT/ab.log | tail -6
