#!/bin/bash
cd "$(dirname "$0")/.."
for cfg in 2 3; do
  for s in 1 2 4 8; do
    echo -n "config $cfg S=$s: "; FDCM_K2_SEGMENTS=$s timeout 300 python tools/run_config.py --config $cfg --check none --reps 9 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pass2_ms %.3f kernels_ms %.3f' % (d['stage_ms']['pass2_ms'], d['kernels_ms']))"
  done
  echo -n "config $cfg legacy: "; FDCM_K2_LEGACY=1 timeout 300 python tools/run_config.py --config $cfg --check none --reps 9 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pass2_ms %.3f kernels_ms %.3f' % (d['stage_ms']['pass2_ms'], d['kernels_ms']))"
done
