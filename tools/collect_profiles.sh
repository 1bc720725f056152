#!/bin/bash
# Collects the round's judged artifacts on the GPU box into gpurun_out/r06/ (copied to profiles/ afterwards).
#   bash tools/collect_profiles.sh [part ...]     parts: trace pmc bench tools ab (default: all)
# Every command runs under its own `timeout` (an intermittent device dead-lock cost this round 25 GPU-minutes once).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${DM_ROUND:-r06}
mkdir -p $O
PARTS="${@:-trace pmc bench tools ab}"
cd /tmp && export TMPDIR=/tmp
export DM_BENCH_WATCHDOG=0     # profiled runs measure in-process (no child process from a process that holds the GPU)
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has trace; then
  # 1. kernel trace + stats of the default bench command
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
  f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
  s=$(find $O/kt -name "*kernel_stats.csv" | head -1)
  cp $s $O/detmatch_bench_kernel_stats.csv
  python3 $R/tools/steady_profile.py $f --marker ema_f32 --steps 8 --skip-last 2 --top 70 > $O/detmatch_step_steady.txt
  python3 $R/tools/step_breakdown.py $f --steps 8 --skip-last 2 > $O/step_breakdown.txt
  python3 $R/tools/dconv_calls.py $f --skip-last 2 > $O/dense_conv_launch_shapes.txt
  python3 $R/tools/aten_big_dispatches.py $f --marker ema_f32 --skip-last 2 --top 40 > $O/aten_big_dispatches.txt
  rm -rf $O/kt
fi
if has pmc; then
  # 2. PMC passes (separate runs, counters only)
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1
  done
  python3 $R/tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_spconv.json > $O/pmc_traffic.log 2>&1
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
fi
cd $R
if has bench; then
  timeout 300 python3 bench.py > $O/bench_detmatch.json 2> $O/bench_detmatch.err
  DM_CONV_MATH=bf16 timeout 120 python3 bench.py --no-cpu-baseline > $O/bench_detmatch_mixed_precision.json 2>/dev/null
  DM_BENCH_PROFILE=waymo timeout 150 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_waymo_fp32.json 2>/dev/null
  DM_BENCH_PROFILE=waymo DM_CONV_MATH=bf16 timeout 150 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_waymo_mixed_precision.json 2>/dev/null
  DM_BENCH_WORKLOAD=pvrcnn timeout 120 python3 bench.py --no-cpu-baseline > $O/bench_pvrcnn.json 2>/dev/null
  DM_BENCH_WORKLOAD=confthr timeout 120 python3 bench.py --no-cpu-baseline > $O/bench_confthr.json 2>/dev/null
fi
if has tools; then
  timeout 200 python3 tools/launch_census.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > $O/launch_census.txt
  timeout 100 python3 tools/phase_timeline.py 2>&1 | grep -v "amdgpu.ids" > $O/phase_timeline.txt
  timeout 100 python3 tools/cpu_vs_gpu_bound.py 2>&1 | grep "detmatch:\|CPU ms" > $O/host_vs_device.txt
  DM_CHAIN=0 timeout 100 python3 tools/cpu_vs_gpu_bound.py 2>&1 | grep "detmatch:\|CPU ms" | sed "s/^/DM_CHAIN=0 /" >> $O/host_vs_device.txt
  timeout 100 python3 tools/find_syncs.py 2>&1 | grep -v "amdgpu.ids" > $O/host_syncs.txt
  timeout 100 python3 tools/steady_timeline.py 7 2>&1 | grep -v "amdgpu.ids" > $O/steady_timeline.txt
fi
if has ab; then
  # 3. round-5 A/Bs (same box, alternated)
  rm -f $O/ab_step_variants.txt
  for i in ${AB_ROUNDS:-1 2}; do
    for v in "default:A=1" "op_by_op:DM_CHAIN=0" "one_lane_glue:DM_TWO_LANES=0" "op_by_op_one_lane(r4):DM_CHAIN=0 DM_TWO_LANES=0" "no_trunk2d_chain:DM_CHAIN_OFF=trunk2d"; do
      n=${v%%:*}; e=${v#*:}
      env $e timeout 100 python3 bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-26s run $i  %.1f ms/step  roofline kernel %.1f us  order %s' % ('$n', d['ms_per_step'], r.get('avg_us') or 0, d['config'].get('stream_order')))" >> $O/ab_step_variants.txt || echo "$n run $i FAILED" >> $O/ab_step_variants.txt
    done
  done
fi
ls -la $O
