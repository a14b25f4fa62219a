#!/bin/bash
# parity tests + A/B (no rocprof): the inner loop while tuning kernels.  Usage: quick_bench.sh TAG [variants]
set -o pipefail
TAG=${1:-quick}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
grep -q passed $OUT/pytest.log && ! grep -q failed $OUT/pytest.log || exit 1
timeout -k 10 300 python tools/ab.py ${2:-0,1} 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.log
