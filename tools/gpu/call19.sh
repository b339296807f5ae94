#!/bin/bash
# round 2, call 19: (1) rehearsal of the N > 1 bench on ONE GPU: two ranks on cuda:0, gloo standing in for RCCL (which
# refuses two ranks on one device) — the sharded HIP kernels, chunk-wise asynchronous all-to-alls, cross-shard
# push-pull, recycling and the JSON of bench.py --gpus 2 on hardware; (2) A/B: quad-cooperative cell stores vs plain ones
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c19
mkdir -p $OUT
cd $ROOT
timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --nodes-per-gpu 262144 --backend gloo --single-device > $OUT/bench_2ranks_one_gpu.json 2> $OUT/bench_2ranks_one_gpu.err; echo "rehearsal rc=$?"; tail -c 2500 $OUT/bench_2ranks_one_gpu.json; tail -5 $OUT/bench_2ranks_one_gpu.err
timeout 600 python tools/ab.py --ticks 120 --rounds 2 serf_amd/csrc/libserf_sim.so serf_amd/csrc/variants/nocoop.so > $OUT/ab.log 2>&1; echo "ab rc=$?"; tail -5 $OUT/ab.log
