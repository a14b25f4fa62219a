#!/bin/bash
set -o pipefail
OUT=gpurun_out/r4m; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
timeout -k 10 300 ./tests/host/test_tile_host 16 20 > $OUT/tile_host.log 2>&1; tail -3 $OUT/tile_host.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -c 600 $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4m/bench_default.json") if l.startswith("{")][-1])
print(json.dumps({k:d[k] for k in ("value","ms_per_step")}), json.dumps(d["parity_in_run"]), json.dumps(d["config"]["learn_tick"]))
print(json.dumps(d["cpu_baseline"])[:300]); print(json.dumps(d.get("secondary"))[:1500]); print(json.dumps(d["roofline"])[:900])
PY
