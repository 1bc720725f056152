#!/bin/bash
# Collects the round's judged artifacts on the GPU box into gpurun_out/r03/ (copied to profiles/ afterwards).
#   bash tools/collect_profiles.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the default bench command
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
s=$(find $O/kt -name "*kernel_stats.csv" | head -1)
cp $s $O/detmatch_bench_kernel_stats.csv
python3 $R/tools/steady_profile.py $f --marker ema_f32 --steps 8 --top 60 > $O/detmatch_step_steady.txt
python3 $R/tools/dconv_calls.py $f > $O/dense_conv_launch_shapes.txt
rm -rf $O/kt
# 2. PMC passes (separate runs, counters only)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1
done
python3 $R/tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_spconv.json > $O/pmc_traffic.log 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cd $R
# 3. plain runs
python3 bench.py > $O/bench_detmatch.json 2> $O/bench_detmatch.err
DM_CONV_MATH=bf16 python3 bench.py --no-cpu-baseline > $O/bench_detmatch_mixed_precision.json 2>/dev/null
DM_BENCH_PROFILE=waymo python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_waymo_fp32.json 2>/dev/null
DM_BENCH_PROFILE=waymo DM_CONV_MATH=bf16 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_waymo_mixed_precision.json 2>/dev/null
DM_BENCH_WORKLOAD=pvrcnn python3 bench.py --no-cpu-baseline > $O/bench_pvrcnn.json 2>/dev/null
DM_BENCH_WORKLOAD=confthr python3 bench.py --no-cpu-baseline > $O/bench_confthr.json 2>/dev/null
python3 tools/launch_census.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > $O/launch_census.txt
python3 tools/phase_timeline.py 2>&1 | grep -v "amdgpu.ids" > $O/phase_timeline.txt
python3 tools/bench_fps.py 2>&1 | grep -v "amdgpu.ids" > $O/fps.txt
(cd tools && python3 bench_dense_conv_math.py 2>&1 | grep -v "amdgpu.ids" > $O/dense_conv_math_modes.txt)
python3 tools/bench_spconv_layers.py 2>&1 | grep -v "amdgpu.ids" > $O/spconv_layers.txt
python3 tools/cpu_vs_gpu_bound.py 2>&1 | tail -1 > $O/host_vs_device.txt
python3 tools/find_syncs.py 2>&1 | grep -v "amdgpu.ids" > $O/host_syncs.txt
# 4. round-3 A/Bs (same box, alternated)
for i in 1 2; do
  for v in "default:A=1" "hipgraph_teacher_trunk:DM_HIPGRAPH=1" "separate_2d_trunks:DM_SHARE_2D_TRUNK=0" "no_lookahead:DM_LOOKAHEAD=0" "fp32_mfma_dense:DM_FP32_CONV=fp32_mfma" "split_no_patch:DM_FP32_CONV=fp32_split_nopatch" "branches:DM_TWO_LANES=1"; do
    n=${v%%:*}; e=${v#*:}
    env $e python3 bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-26s run $i  %.1f ms/step  roofline kernel %.1f us' % ('$n', d['ms_per_step'], d['roofline'].get('avg_us') or 0))" >> $O/ab_step_variants.txt
  done
done
ls -la $O
