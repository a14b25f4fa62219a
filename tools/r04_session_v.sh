#!/bin/bash
set -o pipefail
OUT=gpurun_out/r4v; mkdir -p $OUT
N=sc_gameengine_amd/libsc_tick.so
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
export SC_TICK_LAX_BIND=1
for W in config5 config3dyn config3; do
timeout -k 10 300 python tools/ab_step.py --workload $W --rounds 4 --burst 300 build_ab/base.so $N "$N@196608" "$N@327680" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.log || exit 1
done
