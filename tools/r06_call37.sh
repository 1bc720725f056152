#!/bin/bash
# long runs of the other BASELINE configurations on the final order (stability, not speed)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_long; mkdir -p $O
p() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: %d steps, %.2f ms/step' % (d['steps'], d['ms_per_step']))"; }
DM_BENCH_PROFILE=waymo timeout 300 python3 bench.py --no-cpu-baseline --steps 400 --warmup 5 2>/dev/null | tail -1 | p waymo_fp32 | tee $O/long.txt
DM_BENCH_PROFILE=waymo DM_CONV_MATH=bf16 timeout 300 python3 bench.py --no-cpu-baseline --steps 400 --warmup 5 2>/dev/null | tail -1 | p waymo_mixed | tee -a $O/long.txt
DM_CONV_MATH=bf16 timeout 300 python3 bench.py --no-cpu-baseline --steps 600 --warmup 5 2>/dev/null | tail -1 | p kitti_mixed | tee -a $O/long.txt
DM_BENCH_WORKLOAD=confthr timeout 300 python3 bench.py --no-cpu-baseline --steps 800 --warmup 5 2>/dev/null | tail -1 | p confthr | tee -a $O/long.txt
DM_BENCH_WORKLOAD=pvrcnn timeout 300 python3 bench.py --no-cpu-baseline --steps 1500 --warmup 5 2>/dev/null | tail -1 | p pvrcnn | tee -a $O/long.txt
