#!/bin/bash
set -o pipefail
OUT=gpurun_out/r4w; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
for T in stress_tick stress_lazy stress_broadphase; do timeout -k 10 300 python3 tools/$T.py > $OUT/$T.log 2>&1; echo "$T: $(tail -1 $OUT/$T.log)"; done
