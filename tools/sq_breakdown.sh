#!/bin/bash
# Where the end-of-tick kernel's cycles go: per-pipe active cycles, waits, LDS conflicts (PMC passes of <= 4 SQ counters each, counters only).
#   tools/sq_breakdown.sh <lib.so> <workload> <outdir>
set -o pipefail
LIB=$(realpath $1); W=$2; OUT=$3; mkdir -p $OUT
export TMPDIR=/tmp SC_TICK_LIB=$LIB SC_TICK_LAX_BIND=1
P=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH"; do
  P=$((P+1))
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$P -o sq -- python3 bench.py --workload $W --steps 40 --warmup 10 --no-cpu-baseline --no-parity > $OUT/pass$P.log 2>&1 || { tail -5 $OUT/pass$P.log; exit 1; }
done
python3 - $OUT <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("k_xform_cull", "k_compact_pairs"):
            if k in r["Kernel_Name"]:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: round(sum(v[len(v)//4:]) / len(v[len(v)//4:])) for c, v in sorted(cs.items())} for k, cs in acc.items()}
json.dump(res, open(os.path.join(out, "sq.json"), "w"), indent=1)
print(json.dumps(res.get("k_compact_pairs", {})))
PY
