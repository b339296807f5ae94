#!/bin/bash
# GPU call 1 of round 2: parity of the new kernel, counter calibration, A/B of the builds, steady-state bench.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c1
mkdir -p $OUT
cd $ROOT
rocm-smi --showmeminfo vram --showuse > $OUT/smi.txt 2>&1
nproc > $OUT/host.txt; free -g >> $OUT/host.txt
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/rc.txt
echo "== pytest" ; timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/rc.txt; tail -5 $OUT/pytest.log
echo "== ab" ; timeout 600 python tools/ab.py --ticks 120 --rounds 3 serf_amd/csrc/variants/base.so serf_amd/csrc/libserf_sim.so serf_amd/csrc/variants/occ5.so > $OUT/ab.log 2>&1; echo "ab rc=$?" | tee -a $OUT/rc.txt; tail -4 $OUT/ab.log
echo "== calib" ; timeout 900 bash tools/calib/run_calib.sh > $OUT/calib.log 2>&1; echo "calib rc=$?" | tee -a $OUT/rc.txt
echo "== bench default" ; timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?" | tee -a $OUT/rc.txt
echo "== bench driver args" ; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err; echo "bench20 rc=$?" | tee -a $OUT/rc.txt
echo "== timing" ; timeout 300 python tools/tick_timing.py > $OUT/tick_timing.txt 2>&1; echo "timing rc=$?" | tee -a $OUT/rc.txt
cat $OUT/rc.txt; head -c 1500 $OUT/bench_default.json; echo; cat $OUT/tick_timing.txt | tail -14
