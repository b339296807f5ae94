#!/bin/bash
# round 6: what the long window carries beyond the tick kernel — recycling passes, the push-pull batch, the operations
cd ${GRAFT_REPO_ROOT:-.}
[ "${TESTS:-1}" = 1 ] && python -m pytest tests -q -m gpu -x -k "recycl or lazy or shard or bounds or soak or at_size" 2>&1 | tail -2
B="python bench.py --fanout-model krandomnodes --no-cpu-baseline --no-convergence --no-second-load --steps 20 --warmup 5"
for V in "" "--recycle-interval 0" ""; do
  $B $V 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); lw=d['long_window']; print('[$V]', d['value'], d['ms_per_step'], 'long', d['value_long_window'], lw['ms_per_step'], lw['roofline']['achieved'])"
done
