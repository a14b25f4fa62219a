#!/bin/bash
# Round-2 evidence session (run through gpurun from the repo root): bench lines, rocprofv3 kernel stats per workload, and the
# PMC traffic passes (counters only, separate passes: never combined with tracing domains other than --kernel-trace).
set -o pipefail
TAG=${1:-r02}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; lscpu | grep "Model name") > $OUT/env.txt 2>&1
for W in config3 config3dyn config5; do
  timeout -k 10 400 python3 bench.py --workload $W --steps 200 --warmup 20 $( [ $W = config3 ] || echo --no-cpu-baseline ) > $OUT/bench_$W.json 2> $OUT/bench_$W.err || { tail -5 $OUT/bench_$W.err; exit 1; }
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$W -o tick -- python3 bench.py --workload $W --steps 100 --warmup 10 --no-cpu-baseline > $OUT/prof_$W.log 2>&1 || { tail -5 $OUT/prof_$W.log; exit 1; }
  cp $(find $OUT/prof_$W -name "*kernel_stats.csv" | head -1) $OUT/${W}_kernel_stats.csv
  echo "== $W =="; python3 -c "import json; d=json.load(open('$OUT/bench_$W.json')); print(round(d['value']/1e9,2), 'G ent/s', round(d['ms_per_step']*1e3,1), 'us/step', d['parity_in_run'].get('ok'), 'pairs', d['config']['pairs'], 'frac', round(d['roofline']['frac'],3))"
  head -6 $OUT/${W}_kernel_stats.csv | cut -c1-150
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o $OUT/pmc_calib || exit 1
for W in config3 config5; do
  mkdir -p $OUT/pmc_$W
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$W/calib_$C -o calib -- $OUT/pmc_calib > $OUT/pmc_$W/calib_$C.log 2>&1 || { tail -5 $OUT/pmc_$W/calib_$C.log; exit 1; }
    timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$W/tick_$C -o tick -- python3 bench.py --workload $W --steps 60 --warmup 10 --no-cpu-baseline --no-parity > $OUT/pmc_$W/tick_$C.log 2>&1 || { tail -5 $OUT/pmc_$W/tick_$C.log; exit 1; }
  done
  PMC_WORKLOAD=$W python3 tools/pmc_parse.py $OUT/pmc_$W > $OUT/pmc_$W/summary.txt && cp $OUT/pmc_$W/pmc_traffic.json $OUT/pmc_traffic_$W.json
  python3 -c "import json; d=json.load(open('$OUT/pmc_traffic_$W.json')); print('$W', {k: round(v.get('hbm_bytes_per_launch', 0)/1e6, 1) for k, v in d['kernels'].items()})"
done
rm -f $OUT/pmc_calib
# what one GPU can show of the tiled step: loop-back RCCL, in order / pipelined depth 2-4, 16 and 4 operations per group, captured in order
timeout -k 10 300 python3 tools/pipeline_check.py > $OUT/tile_step_16ops.log 2>&1; tail -1 $OUT/tile_step_16ops.log > $OUT/tile_step_16ops.json
timeout -k 10 300 python3 tools/pipeline_check.py --row 1 > $OUT/tile_step_4ops.log 2>&1; tail -1 $OUT/tile_step_4ops.log > $OUT/tile_step_4ops.json
timeout -k 10 300 python3 tools/pipeline_check.py --graph 1 --only 0 > $OUT/tile_step_graph.log 2>&1; tail -1 $OUT/tile_step_graph.log > $OUT/tile_step_graph.json
timeout -k 10 300 python3 tools/crowd_probe.py > $OUT/crowd.log 2>&1; tail -2 $OUT/crowd.log > $OUT/crowded_sectors.json
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_tile -o pipe -- python3 tools/pipeline_check.py --steps 60 --only 4 --row 1 > $OUT/trace_tile.log 2>&1 && python3 tools/trace_timeline.py $OUT/trace_tile 400 | sed -n 1,60p > $OUT/tile_step_timeline_4ops.txt
cat $OUT/tile_step_16ops.json $OUT/tile_step_4ops.json $OUT/tile_step_graph.json $OUT/crowded_sectors.json | cut -c1-400
