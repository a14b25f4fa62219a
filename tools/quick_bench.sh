#!/bin/bash
# parity tests + bench (no rocprof): the inner loop while tuning kernels
set -o pipefail
TAG=${1:-quick}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log || exit 1
timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tee $OUT/bench.log
timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --stages xform,cull 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bench.log
