#!/bin/bash
# Run on the GPU box: the calibration kernels plain (known bytes, durations), then under rocprofv3 with
# FETCH_SIZE and WRITE_SIZE in separate passes (+ the request counters).  -> gpurun_out/calib/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/calib
mkdir -p $OUT
BIN=$ROOT/tools/calib/hbm_calib
[ -x $BIN ] || hipcc --offload-arch=gfx950 -O3 -o $BIN $ROOT/tools/calib/hbm_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
timeout 120 $BIN 3 > $OUT/known.json 2> $OUT/known.err
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_$i -o p -- $BIN 1 > $OUT/pmc_$i.log 2>&1 || echo "calib pmc pass $i failed: $PMC" >> $OUT/errors.txt
done
python $ROOT/tools/calib/calib_summary.py $OUT > $OUT/calibration.json
cat $OUT/calibration.json
