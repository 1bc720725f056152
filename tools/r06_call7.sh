#!/bin/bash
# Round 6, call 7: scheduling A/B on one box (2D module in front of its 3D neighbour; geometry look-ahead with the
# deferred read-backs), the off-main-lane linear test, host profile.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_call7
mkdir -p $O
echo "== test"
timeout 600 python -m pytest tests/test_blas_turn_gpu.py -x -q -m gpu 2>&1 | tail -12
echo "== A/B (bench.py --steps 30 --warmup 6 --no-cpu-baseline, alternated)"
for round in 1 2; do
  for v in "default:A=1" "2d_after_3d:DM_2D_FIRST=0" "lookahead:DM_LOOKAHEAD=1" "lookahead_2d_after:DM_LOOKAHEAD=1 DM_2D_FIRST=0" "stepwise_geometry:DM_DEFER_GEOMETRY=0"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs DM_BENCH_WATCHDOG=0 timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > $O/ab_${name}_$round.json 2> $O/ab_${name}_$round.err
    python - "$name" "$round" <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r06_call7/ab_%s_%s.json'%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
    print('%-22s round %s  %.2f ms/step  %s' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['config'].get('pseudo_labels_per_step')))
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'FAILED', e)
PY
  done
done
echo "== host profile"
timeout 400 python tools/host_profile.py > $O/host_profile.txt 2>&1; head -70 $O/host_profile.txt | cut -c1-150
