#!/bin/bash
# Collects the round's judged artifacts on the GPU box into gpurun_out/r04/ (copied to profiles/ afterwards).
#   bash tools/collect_profiles.sh [part ...]     parts: trace pmc bench tools ab (default: all)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
PARTS="${@:-trace pmc bench tools ab}"
cd /tmp && export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has trace; then
  # 1. kernel trace + stats of the default bench command
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
  f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
  s=$(find $O/kt -name "*kernel_stats.csv" | head -1)
  cp $s $O/detmatch_bench_kernel_stats.csv
  python3 $R/tools/steady_profile.py $f --marker ema_f32 --steps 8 --top 70 > $O/detmatch_step_steady.txt
  python3 $R/tools/step_breakdown.py $f --steps 8 > $O/step_breakdown.txt
  python3 $R/tools/dconv_calls.py $f > $O/dense_conv_launch_shapes.txt
  rm -rf $O/kt
fi
if has pmc; then
  # 2. PMC passes (separate runs, counters only; bounded: a counter pass that wedges must not eat the budget)
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1
  done
  python3 $R/tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_spconv.json > $O/pmc_traffic.log 2>&1
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
fi
cd $R
if has bench; then
  python3 bench.py > $O/bench_detmatch.json 2> $O/bench_detmatch.err
  DM_CONV_MATH=bf16 python3 bench.py --no-cpu-baseline > $O/bench_detmatch_mixed_precision.json 2>/dev/null
  DM_BENCH_PROFILE=waymo python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_waymo_fp32.json 2>/dev/null
  DM_BENCH_PROFILE=waymo DM_CONV_MATH=bf16 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_waymo_mixed_precision.json 2>/dev/null
  DM_BENCH_WORKLOAD=pvrcnn python3 bench.py --no-cpu-baseline > $O/bench_pvrcnn.json 2>/dev/null
  DM_BENCH_WORKLOAD=confthr python3 bench.py --no-cpu-baseline > $O/bench_confthr.json 2>/dev/null
fi
if has tools; then
  python3 tools/launch_census.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > $O/launch_census.txt
  python3 tools/phase_timeline.py 2>&1 | grep -v "amdgpu.ids" > $O/phase_timeline.txt
  python3 tools/bench_fps.py 2>&1 | grep -v "amdgpu.ids" > $O/fps.txt
  (cd tools && python3 bench_dense_conv_math.py 2>&1 | grep -v "amdgpu.ids" > $O/dense_conv_math_modes.txt)
  python3 tools/bench_spconv_layers.py 2>&1 | grep -v "amdgpu.ids" > $O/spconv_layers.txt
  python3 tools/bench_dense_wgrad.py 2>&1 | grep -v "amdgpu.ids" > $O/dense_wgrad_fused_vs_per_tap.txt
  python3 tools/bench_tall_wgrad.py 2>&1 | grep -v "amdgpu.ids" > $O/tall_skinny_wgrad_blas_vs_own.txt
  bash tools/pmc_step_lds.sh r04_lds > /dev/null 2>&1; cp $R/gpurun_out/r04_lds/lds_conflicts.txt $O/lds_conflicts.txt
  python3 tools/cpu_vs_gpu_bound.py 2>&1 | grep "detmatch:\|CPU ms" > $O/host_vs_device.txt
  python3 tools/find_syncs.py 2>&1 | grep -v "amdgpu.ids" > $O/host_syncs.txt
  tools/launch_cost_bin > $O/launch_cost.txt 2>&1
  python3 tools/host_cost_probe.py 2>&1 | grep -v "amdgpu.ids" >> $O/launch_cost.txt
fi
if has ab; then
  # 3. round-4 A/Bs (same box, alternated)
  rm -f $O/ab_step_variants.txt
  for i in 1 2 3; do
    for v in "default:A=1" "per_tap_dense_wgrad:DM_FP32_CONV=fp32_split_tapwise_wgrad" "no_weight_planes:DM_DCONV_PLANES=0" "per_layer_sparse_wgrad:DM_SPCONV_WGRAD_BATCH=0" "no_collect_early:DM_COLLECT_EARLY=0" "hipgraph_sections:DM_HIPGRAPH=1" "branches:DM_TWO_LANES=1" "pairs:DM_LANE_MODE=pairs"; do
      n=${v%%:*}; e=${v#*:}
      env $e python3 bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); o=d['roofline']['other_kernels']; w=d['roofline']['all_spconv']['wgrad']; print('%-26s run $i  %.1f ms/step  roofline kernel %.1f us  dense fwd+dgrad %.2f ms  sparse wgrad %.0f us (%d launches)' % ('$n', d['ms_per_step'], d['roofline'].get('avg_us') or 0, o['dense_conv.fwd+dgrad']['ms_per_step'], w['us_per_step'], w['launches_per_step']))" >> $O/ab_step_variants.txt
    done
  done
fi
ls -la $O
