#!/bin/bash
# round 3, call 26: the N > 1 bench line rehearsed on ONE GPU after the round's changes (two ranks on cuda:0, gloo standing in
# for RCCL, self-launched — no torch.distributed.run), and the one-process-per-shard parity check
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c26
mkdir -p $OUT
cd $ROOT
timeout 600 python bench.py --gpus 2 --backend gloo --single-device --nodes-per-gpu 262144 --steps 40 --warmup 10 --no-cpu-baseline > $OUT/bench_2ranks.json 2> $OUT/bench_2ranks.err; echo "bench 2 ranks rc=$?"
tail -c 1500 $OUT/bench_2ranks.json; echo; tail -5 $OUT/bench_2ranks.err
timeout 600 python tools/shard_procs_check.py > $OUT/shard_procs.txt 2>&1; echo "shard procs rc=$?"; tail -6 $OUT/shard_procs.txt
