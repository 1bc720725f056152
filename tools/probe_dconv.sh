#!/bin/bash
# forward time of the dense-conv layers with the default library and with timing-probe builds (tools/build_alt.sh)
cd $GRAFT_REPO_ROOT/tools
for v in default "$@"; do
  echo "== $v"
  if [ $v = default ]; then python3 bench_dense_conv_math.py 2>&1 | grep -v amdgpu.ids | cut -c1-80
  else DM_LIB_PATH=$GRAFT_REPO_ROOT/tools/altlib/lib_$v.so python3 bench_dense_conv_math.py 2>&1 | grep -v amdgpu.ids | cut -c1-80; fi
done
