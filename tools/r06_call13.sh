#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_call13
mkdir -p $O
timeout 900 python -m pytest tests/test_ssl_gpu.py -q -m gpu -x -k "early or lanes or soak" 2>&1 | tail -4
for round in 1 2 3; do
  for v in "inside:A=1" "after:DM_2D_INSIDE_3D=0"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs DM_BENCH_WATCHDOG=0 timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name round $round  %.2f ms/step' % d['ms_per_step'])"
  done
done
timeout 200 python tools/phase_timeline.py 2>&1 | grep -v amdgpu.ids > $O/phase_timeline.txt; grep -E "HardPseudo|Bboxes3DTo2D|backward\+clip|host issued" $O/phase_timeline.txt
