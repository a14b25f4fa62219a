#!/bin/bash
# Round-3 evidence session (run through gpurun from the repo root).  What VERDICT r02 item 1 asked for:
#   * bench lines per workload (the default line carries the config3dyn / config5 secondary legs);
#   * rocprofv3 --kernel-trace --stats of `bench.py --profile-run`: warm-up + the timed steps and nothing else (no event timing,
#     no every-launch pass, no end-to-end pass), so <workload>_kernel_stats.csv IS the timed pass;
#   * the same trace of the full default run split per pass (tools/trace_passes.py);
#   * PMC traffic per workload (FETCH_SIZE / WRITE_SIZE in separate passes, counters only, calibrated on known-byte kernels).
set -o pipefail
TAG=${1:-r03}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; lscpu | grep "Model name") > $OUT/env.txt 2>&1
timeout -k 10 500 python3 bench.py --steps 200 --warmup 20 > $OUT/bench_config3.json 2> $OUT/bench_config3.err || { tail -5 $OUT/bench_config3.err; exit 1; }
for W in config3dyn config5; do
  timeout -k 10 400 python3 bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_$W.json 2> $OUT/bench_$W.err || { tail -5 $OUT/bench_$W.err; exit 1; }
done
for W in config3 config3dyn config5; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$W -o tick -- python3 bench.py --workload $W --steps 400 --warmup 10 --profile-run > $OUT/prof_$W.log 2>&1 || { tail -5 $OUT/prof_$W.log; exit 1; }
  cp $(find $OUT/prof_$W -name "*kernel_stats.csv" | head -1) $OUT/${W}_kernel_stats.csv
  echo "== $W =="; python3 -c "import json; d=json.load(open('$OUT/bench_$W.json')); print(round(d['value']/1e9,2), 'G ent/s', round(d['ms_per_step']*1e3,1), 'us/step', d['parity_in_run'].get('ok'), 'pairs', d['config']['pairs'], 'frac', round(d['roofline']['frac'],3), 'k1 us', round(d['roofline']['avg_launch_ms']*1e3,2))"
  head -5 $OUT/${W}_kernel_stats.csv | cut -c1-160
done
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_full_config3 -o tick -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $OUT/trace_full_config3.log 2>&1 || { tail -5 $OUT/trace_full_config3.log; exit 1; }
python3 tools/trace_passes.py $OUT/trace_full_config3 --warmup 10 --steps 100 > $OUT/config3_kernel_passes.json && head -40 $OUT/config3_kernel_passes.json
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o $OUT/pmc_calib || exit 1
for W in config3 config3dyn config5; do
  mkdir -p $OUT/pmc_$W
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$W/calib_$C -o calib -- $OUT/pmc_calib > $OUT/pmc_$W/calib_$C.log 2>&1 || { tail -5 $OUT/pmc_$W/calib_$C.log; exit 1; }
    timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$W/tick_$C -o tick -- python3 bench.py --workload $W --steps 60 --warmup 10 --profile-run > $OUT/pmc_$W/tick_$C.log 2>&1 || { tail -5 $OUT/pmc_$W/tick_$C.log; exit 1; }
  done
  PMC_WORKLOAD=$W python3 tools/pmc_parse.py $OUT/pmc_$W > $OUT/pmc_$W/summary.txt && cp $OUT/pmc_$W/pmc_traffic.json $OUT/pmc_traffic_$W.json
  python3 -c "import json; d=json.load(open('$OUT/pmc_traffic_$W.json')); print('$W', {k: round(v.get('hbm_bytes_per_launch', 0)/1e6, 1) for k, v in d['kernels'].items()})"
done
rm -f $OUT/pmc_calib
# crowded sectors: the all-district probe and a district of 512 crowded sectors inside the 1M world (round 2's one-wave search,
# 2.06 ms on this probe, left the source with the round-3 clean-up: profiles/r02/crowded_sectors.json and the history hold it)
(timeout -k 10 200 python3 tools/crowd_probe.py; timeout -k 10 200 python3 tools/crowd_probe2.py) 2>&1 | grep -v amdgpu.ids > $OUT/crowded_sectors.json; cat $OUT/crowded_sectors.json
# home slots on / off in one process, three workloads
for W in config3 config3dyn config5; do timeout -k 10 250 python3 tools/ab_step.py --rounds 3 --workload $W sc_gameengine_amd/libsc_tick.so@2 sc_gameengine_amd/libsc_tick.so 2>&1 | grep -v amdgpu.ids >> $OUT/ab_home_slots_final.log; done; cat $OUT/ab_home_slots_final.log
# lazy / unchanged records on / off (SC_TICK_VARIANT bit 5) in one process, three workloads
for W in config3 config3dyn config5; do timeout -k 10 250 python3 tools/ab_step.py --rounds 3 --workload $W sc_gameengine_amd/libsc_tick.so@32 sc_gameengine_amd/libsc_tick.so 2>&1 | grep -v amdgpu.ids >> $OUT/ab_lazy_on_off_final.log; done; cat $OUT/ab_lazy_on_off_final.log
# what one GPU can show of the tiled step: loop-back RCCL, in order / pipelined, 16 and 4 operations per group
timeout -k 10 300 python3 tools/pipeline_check.py > $OUT/tile_step_16ops.log 2>&1; tail -1 $OUT/tile_step_16ops.log > $OUT/tile_step_16ops.json
timeout -k 10 300 python3 tools/pipeline_check.py --row 1 > $OUT/tile_step_4ops.log 2>&1; tail -1 $OUT/tile_step_4ops.log > $OUT/tile_step_4ops.json
cat $OUT/tile_step_16ops.json $OUT/tile_step_4ops.json | cut -c1-600
