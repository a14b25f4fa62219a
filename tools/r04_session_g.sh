#!/bin/bash
set -o pipefail
OUT=gpurun_out/r4i; mkdir -p $OUT
N=sc_gameengine_amd/libsc_tick.so
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
for W in config5 config3dyn config3; do
timeout -k 10 300 python tools/ab_step.py --workload $W --rounds 4 --burst 300 build_ab/base.so $N build_ab/bcast0.so build_ab/bcast8.so build_ab/nosweep.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.log || exit 1
done
timeout -k 10 300 python tools/stress_broadphase.py > $OUT/stress_broadphase.log 2>&1; tail -1 $OUT/stress_broadphase.log
timeout -k 10 300 python tools/stress_lazy.py > $OUT/stress_lazy.log 2>&1; tail -1 $OUT/stress_lazy.log
timeout -k 10 300 python tools/stress_tiles.py > $OUT/stress_tiles.log 2>&1; tail -1 $OUT/stress_tiles.log
timeout -k 10 300 python tools/stress_tick.py > $OUT/stress_tick.log 2>&1; tail -1 $OUT/stress_tick.log
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counter_names.txt; wc -l $OUT/sq_counter_names.txt
