#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/gpu/pmc_ab.sh base serf_amd/csrc/variants/base.so 2>&1 | tail -2
bash tools/gpu/pmc_ab.sh new serf_amd/csrc/libserf_sim.so 2>&1 | tail -2
