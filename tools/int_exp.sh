#!/bin/bash
cd "$(dirname "$0")/.."
for cfg in ${INT_CFGS:-2 3}; do for o in ${INT_MODES:-0 1 2}; do
  echo -n "config $cfg only_mode=$o: "; FDCM_INT_ONLY=$o timeout 300 python tools/run_config.py --config $cfg --check none --reps 9 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('integral_ms %.3f' % d['stage_ms']['integral_ms'])"
done; done
