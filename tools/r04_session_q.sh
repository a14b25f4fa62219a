#!/bin/bash
# round 4: homeCount word merge + fixed-slot border messages -- tests, A/B against the round-3 build, the loop-back tile step
set -o pipefail
OUT=gpurun_out/r4q; mkdir -p $OUT
N=sc_gameengine_amd/libsc_tick.so
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
export SC_TICK_LAX_BIND=1
for W in config3 config3dyn config5; do
timeout -k 10 300 python tools/ab_step.py --workload $W --rounds 4 --burst 300 build_ab/base.so $N 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.log || exit 1
done
unset SC_TICK_LAX_BIND
for F in 0 4; do timeout -k 10 300 python tools/pipeline_check.py --only $F --steps 400 2>&1 | grep -v amdgpu.ids | tee -a $OUT/pipeline.log || exit 1; done
timeout -k 10 300 python tools/pipeline_check.py --only 0 --row 1 --steps 400 2>&1 | grep -v amdgpu.ids | tee -a $OUT/pipeline.log || exit 1
timeout -k 10 300 python tools/pipeline_check.py --only 0 --workload config5 --steps 400 2>&1 | grep -v amdgpu.ids | tee -a $OUT/pipeline.log || true
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o pc -- python $GRAFT_REPO_ROOT/tools/pipeline_check.py --only 0 --steps 300 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r4q/prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
for T in stress_tiles stress_lazy; do timeout -k 10 300 python tools/$T.py > $OUT/$T.log 2>&1; tail -1 $OUT/$T.log; done
