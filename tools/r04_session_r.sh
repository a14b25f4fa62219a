#!/bin/bash
set -o pipefail
OUT=gpurun_out/r4r; mkdir -p $OUT
N=sc_gameengine_amd/libsc_tick.so
export SC_TICK_LAX_BIND=1
for W in config3 config3dyn config5; do
timeout -k 10 300 python tools/ab_step.py --workload $W --rounds 4 --burst 300 build_ab/base.so $N 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.log || exit 1
done
