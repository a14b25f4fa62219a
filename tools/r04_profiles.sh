#!/bin/bash
# Round-4 evidence session (through gpurun from the repo root), in two parts so that each fits one call:
#   A: bench lines per workload (200 steps), rocprofv3 --kernel-trace --stats of `bench.py --profile-run` per workload (the timed pass
#      and nothing else), the default 20-step line the driver runs, the trace of a full default run split per pass;
#   B: PMC traffic per workload (FETCH_SIZE / WRITE_SIZE in separate passes, counters + kernel trace only, calibrated on known-byte
#      kernels), the tiled step on the loop-back exchange (in order / pipelined, 4 and 16 operations), the runtime's own dispatches
#      of that step attributed to the HIP calls behind them, the stress tools.
set -o pipefail
PART=${1:-A}; TAG=${2:-r04}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [ $PART = A ]; then
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; lscpu | grep "Model name") > $OUT/env.txt 2>&1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_config3_driver_20_steps.json 2> $OUT/bench_driver.err || { tail -5 $OUT/bench_driver.err; exit 1; }
timeout -k 10 500 python3 bench.py --steps 200 --warmup 20 > $OUT/bench_config3.json 2> $OUT/bench_config3.err || { tail -5 $OUT/bench_config3.err; exit 1; }
for W in config3dyn config5; do
  timeout -k 10 400 python3 bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_$W.json 2> $OUT/bench_$W.err || { tail -5 $OUT/bench_$W.err; exit 1; }
done
for W in config3 config3dyn config5; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$W -o tick -- python3 bench.py --workload $W --steps 400 --warmup 10 --profile-run > $OUT/prof_$W.log 2>&1 || { tail -5 $OUT/prof_$W.log; exit 1; }
  cp $(find $OUT/prof_$W -name "*kernel_stats.csv" | head -1) $OUT/${W}_kernel_stats.csv
  echo "== $W =="; python3 -c "import json; d=json.load(open('$OUT/bench_$W.json')); r=d['roofline']; e=r.get('end_of_tick_kernel',{}); print(round(d['value']/1e9,2), 'G ent/s', round(d['ms_per_step']*1e3,1), 'us/step', d['parity_in_run'].get('ok'), 'pairs', d['config']['pairs'], 'frac', round(r['frac'],3), 'k1 us', round(r['avg_launch_ms']*1e3,2), 'eot us', round(e.get('avg_launch_ms',0)*1e3,2), 'eot frac', round(e.get('frac',0),3))"
  head -5 $OUT/${W}_kernel_stats.csv | cut -c1-160
done
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_full_config3 -o tick -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $OUT/trace_full_config3.log 2>&1 || { tail -5 $OUT/trace_full_config3.log; exit 1; }
python3 tools/trace_passes.py $OUT/trace_full_config3 --warmup 10 --steps 100 --bench-line $OUT/trace_full_config3.log > $OUT/config3_kernel_passes.json && head -30 $OUT/config3_kernel_passes.json
rm -rf $OUT/prof_config3 $OUT/prof_config3dyn $OUT/prof_config5 $OUT/trace_full_config3
fi
if [ $PART = B ]; then
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o $OUT/pmc_calib || exit 1
for W in config3 config3dyn config5; do
  mkdir -p $OUT/pmc_$W
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$W/calib_$C -o calib -- $OUT/pmc_calib > $OUT/pmc_$W/calib_$C.log 2>&1 || { tail -5 $OUT/pmc_$W/calib_$C.log; exit 1; }
    timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$W/tick_$C -o tick -- python3 bench.py --workload $W --steps 60 --warmup 10 --profile-run > $OUT/pmc_$W/tick_$C.log 2>&1 || { tail -5 $OUT/pmc_$W/tick_$C.log; exit 1; }
  done
  PMC_WORKLOAD=$W python3 tools/pmc_parse.py $OUT/pmc_$W > $OUT/pmc_$W/summary.txt && cp $OUT/pmc_$W/pmc_traffic.json $OUT/pmc_traffic_$W.json
  python3 -c "import json; d=json.load(open('$OUT/pmc_traffic_$W.json')); print('$W', {k: round(v.get('hbm_bytes_per_launch', 0)/1e6, 1) for k, v in d['kernels'].items()})"
  rm -rf $OUT/pmc_$W
done
rm -f $OUT/pmc_calib
timeout -k 10 300 python3 tools/pipeline_check.py > $OUT/tile_step_16ops.log 2>&1 || { tail -5 $OUT/tile_step_16ops.log; exit 1; }
tail -1 $OUT/tile_step_16ops.log > $OUT/tile_step_16ops.json
timeout -k 10 300 python3 tools/pipeline_check.py --row 1 > $OUT/tile_step_4ops.log 2>&1 || { tail -5 $OUT/tile_step_4ops.log; exit 1; }
tail -1 $OUT/tile_step_4ops.log > $OUT/tile_step_4ops.json
cat $OUT/tile_step_16ops.json $OUT/tile_step_4ops.json | cut -c1-900
for F in 0 4; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tile_$F -o t -- python3 tools/pipeline_check.py --only $F --row 1 --steps 300 > $OUT/prof_tile_$F.log 2>&1 || { tail -5 $OUT/prof_tile_$F.log; exit 1; }
  cp $(find $OUT/prof_tile_$F -name "*kernel_stats.csv" | head -1) $OUT/tile_step_4ops_$([ $F = 0 ] && echo inorder || echo pipelined)_kernel_stats.csv
  rm -rf $OUT/prof_tile_$F
done
head -8 $OUT/tile_step_4ops_inorder_kernel_stats.csv | cut -c1-140
timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $OUT/attr -o t -- python3 tools/pipeline_check.py --only 0 --row 1 --steps 200 > $OUT/attr.log 2>&1 || { tail -5 $OUT/attr.log; exit 1; }
python3 tools/attribute_dispatches.py $OUT/attr --steps 230 > $OUT/tile_step_runtime_dispatches.json; head -60 $OUT/tile_step_runtime_dispatches.json
rm -rf $OUT/attr
for T in stress_tiles stress_lazy stress_broadphase stress_tick stress_traffic; do timeout -k 10 300 python3 tools/$T.py > $OUT/$T.log 2>&1; echo "$T: $(tail -1 $OUT/$T.log)"; done
fi
