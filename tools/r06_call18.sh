#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_ssl_gpu.py tests/test_chain_gpu.py tests/test_multirank_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|^FAILED|Error" | tail -5
for round in 1 2 3; do
  for v in "trunk_on_2d_lane:A=1" "trunk_on_main:DM_TRUNK_ON_2D_LANE=0"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs DM_BENCH_WATCHDOG=0 timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name round $round  %.2f ms/step' % (d['ms_per_step']))"
  done
done
timeout 200 python tools/phase_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_phase_timeline_trunk.txt; head -8 gpurun_out/r06_phase_timeline_trunk.txt; grep -E "HardPseudo|Opd_SimpleTest|backward\+clip" gpurun_out/r06_phase_timeline_trunk.txt
