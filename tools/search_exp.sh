#!/bin/bash
cd "$(dirname "$0")/.."
for t in 0 1; do
  if [ $t = 1 ]; then export FDCM_SEARCH_TILED=1; else unset FDCM_SEARCH_TILED; fi
  echo "== tiled=$t"
  timeout 300 python tools/fuzz_parity.py 40 81 2>&1 | tail -1
  timeout 300 python tools/run_config.py --config 2p --check none --reps 9 --search | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  search alone:', d['search'])"
  for i in 1 2; do python3 bench.py --steps 100 --warmup 10 --cpu-sample 0 --single-frames 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  pipelined: %.1f M/s %.3f ms/step; search kernel in region %.3f ms' % (d['value']/1e6, d['ms_per_step'], d['in_timed_region']['search_kernel_ms']))"; done
done
