R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_frcnn_gpu.py -m gpu -q -k "roi or frcnn or faster" 2>&1 | tail -1
for i in 1 2; do
DM_ROI_NARROW=0 python3 tools/bench_kernels.py 2>&1 | grep -i "roi_align_fpn forward" | sed 's/^/old /'
DM_ROI_NARROW=1 python3 tools/bench_kernels.py 2>&1 | grep -i "roi_align_fpn forward" | sed 's/^/new /'
done
for i in 1 2; do
DM_BENCH_WORKLOAD=confthr python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('confthr', d['ms_per_step'])"
DM_BENCH_PROFILE=waymo python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('waymo', d['ms_per_step'])"
done
