#!/bin/bash
set -o pipefail
OUT=gpurun_out/r4s; mkdir -p $OUT
N=sc_gameengine_amd/libsc_tick.so
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
export SC_TICK_LAX_BIND=1
for W in config3 config3dyn config5; do
timeout -k 10 300 python tools/ab_step.py --workload $W --rounds 4 --burst 300 build_ab/base.so $N 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.log || exit 1
done
unset SC_TICK_LAX_BIND
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_config3.json 2> $OUT/bench_config3.err || { tail -5 $OUT/bench_config3.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4s/bench_config3.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"], {k:v for k,v in d.items() if k in ("end_of_tick","secondary")})
PY
