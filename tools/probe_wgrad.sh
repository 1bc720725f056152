#!/bin/bash
# weight-gradient time of the 3 x 3 layers with the default library and with timing-probe builds (tools/build_alt.sh)
cd $GRAFT_REPO_ROOT/tools
for v in default "$@"; do
  echo "== $v"
  if [ $v = default ]; then python3 bench_dense_wgrad.py 2>&1 | grep -v amdgpu.ids | cut -c1-110
  else DM_LIB_PATH=$GRAFT_REPO_ROOT/tools/altlib/lib_$v.so python3 bench_dense_wgrad.py 2>&1 | grep -v amdgpu.ids | cut -c1-110; fi
done
