#!/bin/bash
# round 2, call 23: seeded sweep over the configuration space (HIP vs oracle, digest every tick) + the bench line once more
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c23
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "random_configurations" > $OUT/pytest_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -12 $OUT/pytest_fuzz.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$OUT/bench_20_5.json')); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['config']['departure_from_survey_8d'][:60])"
