#!/bin/bash
set -o pipefail
OUT=gpurun_out/r4y; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_tiles_edge.py tests/test_gpu_comm.py -q -x > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
for F in 0 4; do timeout -k 10 300 python tools/pipeline_check.py --only $F --row 1 --steps 400 2>&1 | grep "^{" | tee -a $OUT/pipeline.log || exit 1; done
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python3 tools/pipeline_check.py --only 0 --row 1 --steps 300 > $OUT/prof.log 2>&1 || exit 1
head -7 $(find $OUT/prof -name "*kernel_stats.csv" | head -1) | cut -c1-150
