R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; rm -rf $O; mkdir -p $O; cd $R
for i in 1 2 3 4 5; do
  for v in "default:A=1" "hipgraph:DM_HIPGRAPH=1" "nopatch:DM_FP32_CONV=fp32_split_nopatch"; do
    n=${v%%:*}; e=${v#*:}
    env $e python3 bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s %d %.2f' % ('$n', $i, d['ms_per_step']))" >> $O/ab.txt
  done
done
python3 - <<'PY'
import collections,statistics
d=collections.defaultdict(list)
for l in open('gpurun_out/r3y/ab.txt'):
    n,i,v=l.split(); d[n].append(float(v))
for n,v in d.items(): print(n, ' '.join('%.1f'%x for x in v), '| median %.1f mean %.1f' % (statistics.median(v), statistics.mean(v)))
PY
