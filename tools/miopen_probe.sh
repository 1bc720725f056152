for mode in default bench; do
  if [ $mode = bench ]; then export DM_CUDNN_BENCHMARK=1; else unset DM_CUDNN_BENCHMARK; fi
  echo "== $mode"; python bench.py --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | cut -c1-140
done
