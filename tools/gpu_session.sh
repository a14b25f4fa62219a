#!/bin/bash
# One GPU-box session: parity tests, smoke, a short bench, and a rocprofv3 kernel trace.
# Usage (from the repo root, through gpurun): bash tools/gpu_session.sh [tag]
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocminfo ==" > $OUT/env.txt; (rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; lscpu | grep "Model name") >> $OUT/env.txt 2>&1
echo "== pytest -m gpu ==" | tee $OUT/pytest.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tee -a $OUT/pytest.log || exit 1
echo "== smoke ==" | tee $OUT/smoke.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee -a $OUT/smoke.log || exit 1
echo "== bench ==" | tee $OUT/bench.log
timeout -k 10 600 python bench.py --steps 200 --warmup 20 2>&1 | tee -a $OUT/bench.log || exit 1
echo "== rocprofv3 kernel trace ==" | tee $OUT/prof.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o tick -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/prof_bench.log 2>&1 || { tail -20 $OUT/prof_bench.log; exit 1; }
find $OUT/prof -name "*stats*" | head; for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -12 $f; done | tee -a $OUT/prof.log
