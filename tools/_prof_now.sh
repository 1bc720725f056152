R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; rm -rf $O; mkdir -p $O; cd $R
python3 bench.py > $O/bench_detmatch.json 2> $O/bench_detmatch.err
DM_CONV_MATH=bf16 python3 bench.py --no-cpu-baseline > $O/bench_detmatch_mixed_precision.json 2>/dev/null
DM_BENCH_PROFILE=waymo python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_waymo_fp32.json 2>/dev/null
DM_BENCH_PROFILE=waymo DM_CONV_MATH=bf16 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_waymo_mixed_precision.json 2>/dev/null
DM_BENCH_WORKLOAD=pvrcnn python3 bench.py --no-cpu-baseline > $O/bench_pvrcnn.json 2>/dev/null
DM_BENCH_WORKLOAD=confthr python3 bench.py --no-cpu-baseline > $O/bench_confthr.json 2>/dev/null
python3 tools/phase_timeline.py 2>&1 | grep -v "amdgpu.ids" > $O/phase_timeline.txt
python3 tools/cpu_vs_gpu_bound.py 2>&1 | tail -1 > $O/host_vs_device.txt
python3 - <<'PY'
import json
for f in ['bench_detmatch','bench_detmatch_mixed_precision','bench_waymo_fp32','bench_waymo_mixed_precision','bench_pvrcnn','bench_confthr']:
    d=json.loads(open('gpurun_out/r03f/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline'].get('traffic_source','')[:30])
PY
head -3 $O/phase_timeline.txt
