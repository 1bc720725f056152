R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2 3; do python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
