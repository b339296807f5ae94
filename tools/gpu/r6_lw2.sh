#!/bin/bash
# round 6: the long window with every plane mapped at creation (SERF_SIM_EAGER=1) against planes on demand — what the mapping of a
# chunk of ring planes (hipMemCreate / hipMemMap / hipMemSetAccess + a 128 MiB fill, twice) inside the timed ticks costs
cd ${GRAFT_REPO_ROOT:-.}
B="python bench.py --fanout-model krandomnodes --no-cpu-baseline --no-convergence --no-second-load --steps 20 --warmup 5"
for V in "lazy" "eager" "lazy" "eager"; do
  if [ $V = eager ]; then export SERF_SIM_EAGER=1; else unset SERF_SIM_EAGER; fi
  $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); lw=d['long_window']; print('[$V]', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], 'long', d['value_long_window'], lw['ms_per_step'], lw['roofline'].get('kernel_ms'))"
done
