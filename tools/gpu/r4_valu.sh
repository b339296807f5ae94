#!/bin/bash
# VALU instructions per wave of ONE tick-kernel launch with parts of the tick left out (libserf_sim_ablate.so): where the
# instructions are.  One process per mask; the ablated launch is the last tick-kernel dispatch of the process.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4valu
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for M in krandomnodes bijection; do
  for MASK in 0 1 2 4 8 16 32 64 65 3; do
    D=$OUT/$M/m$MASK
    mkdir -p $D
    timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d $D -o p -- python $ROOT/tools/ablate.py 1048576 --fanout-model $M masks=$MASK ticks=333 > $D.log 2>&1
    python - <<PY
import csv, glob
f = glob.glob("$D/**/p_counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "tick_kernel" in r["Kernel_Name"]]
last = max(int(r["Dispatch_Id"]) for r in rows)
c = {r["Counter_Name"]: float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) == last}
w = c.get("SQ_WAVES", 1.0)
print("$M mask $MASK:", {k: round(v / w, 1) for k, v in c.items() if k != "SQ_WAVES"}, "us", [l.split()[-2] for l in open("$D.log") if l.rstrip().endswith(" us")])
PY
  done
done
