#!/bin/bash
set -o pipefail
OUT=gpurun_out/r4v; mkdir -p $OUT
N=sc_gameengine_amd/libsc_tick.so
timeout -k 10 600 python -m pytest tests/test_gpu_broadphase.py tests/test_gpu_fuzz.py tests/test_gpu_stress.py tests/test_gpu_tiles.py -q -x 2>&1 | tail -1
for W in config5 config3dyn config3; do
timeout -k 10 300 python tools/ab_step.py --workload $W --rounds 4 --burst 300 build_ab/prev.so $N 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.log || exit 1
done
for T in stress_broadphase stress_lazy stress_tick; do timeout -k 10 300 python3 tools/$T.py > $OUT/$T.log 2>&1; echo "$T: $(tail -1 $OUT/$T.log)"; done
