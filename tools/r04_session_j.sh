#!/bin/bash
set -o pipefail
OUT=gpurun_out/r4j; mkdir -p $OUT
N=sc_gameengine_amd/libsc_tick.so
timeout -k 10 900 python -m pytest tests/test_gpu_broadphase.py tests/test_gpu_tiles.py tests/test_gpu_stress.py tests/test_gpu_bench_rehearsal.py -x -q -m gpu > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
for W in config5 config3dyn; do
timeout -k 10 300 python tools/ab_step.py --workload $W --rounds 4 --burst 300 build_ab/base.so $N build_ab/nosweep.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.log || exit 1
done
for W in config5 config3dyn; do
bash tools/sq_breakdown.sh $N $W $OUT/sq_new_$W
bash tools/sq_breakdown.sh build_ab/base.so $W $OUT/sq_base_$W
done
