#!/bin/bash
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/r06_pack2; mkdir -p $O
for cap in 256 1024 4096; do
  cd /tmp
  DM_PACK_BLOCKS_CAP=$cap timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$cap -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/b$cap.json 2>/dev/null < /dev/null
  s=$(find $O/kt$cap -name "*kernel_stats.csv" 2>/dev/null | head -1)
  if [ -n "$s" ]; then echo "cap $cap: $(grep -i 'dconv_pack_batch' "$s" | cut -d, -f2-4)"; fi
  rm -rf $O/kt$cap
done | tee $O/pack_stats.txt
cd $R
export DM_BENCH_WATCHDOG=0
for round in 1 2; do for cap in 256 1024 4096; do DM_PACK_BLOCKS_CAP=$cap timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null < /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cap $cap round $round  %.2f ms' % d['ms_per_step'])"; done; done | tee $O/bench.txt
