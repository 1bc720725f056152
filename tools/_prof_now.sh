R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2 3; do
DM_BN_DEFER=0 python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('nodefer', d['ms_per_step'])"
python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('defer  ', d['ms_per_step'])"
done
timeout 1500 python -m pytest tests/test_ssl_gpu.py tests/test_ops_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed"
