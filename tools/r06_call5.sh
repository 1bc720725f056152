#!/bin/bash
# Round 6, call 5: why an initialised RCCL communicator costs 30 ms per iteration (hardware-queue census), the suite, the timeline.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_call5
mkdir -p $O
echo "== full GPU suite (batched FPS, lazy glue, token)"
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12
echo "== nccl one rank, three lanes, 100 iterations: hardware queues"
for q in 2 4 5 6 8 12; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node 1 tools/lane_soak.py run 100 2>&1 | grep "steps ok" | sed "s/^/nccl GPU_MAX_HW_QUEUES=$q: /"
done
for q in 4 6; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/lane_soak.py run 100 2>&1 | grep "steps ok" | sed "s/^/no process group GPU_MAX_HW_QUEUES=$q: /"
done
NCCL_MAX_NCHANNELS=2 timeout 300 python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node 1 tools/lane_soak.py run 100 2>&1 | grep "steps ok" | sed "s/^/nccl NCCL_MAX_NCHANNELS=2: /"
echo "== timeline"
timeout 300 python tools/phase_timeline.py > $O/phase_timeline.txt 2>&1; head -45 $O/phase_timeline.txt
