#!/bin/bash
# PMC traffic session (counters only: never combined with sys/hip/hsa tracing).
# Separate passes for FETCH_SIZE and WRITE_SIZE (TCC slots: 3 + 2 do not fit one pass).
set -o pipefail
TAG=${1:-pmc}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o $OUT/pmc_calib || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/calib_$C -o calib -- $OUT/pmc_calib > $OUT/calib_$C.log 2>&1 || { tail -5 $OUT/calib_$C.log; exit 1; }
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/tick_$C -o tick -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --sample 0 > $OUT/tick_$C.log 2>&1 || { tail -5 $OUT/tick_$C.log; exit 1; }
  echo "pass $C done"
done
rm -f $OUT/pmc_calib
find $OUT -name "*counter_collection.csv" | head
python3 tools/pmc_parse.py $OUT | tee $OUT/pmc_summary.txt
