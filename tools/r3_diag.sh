#!/bin/bash
# round-3 diagnostics of the L2 sweep at config 3 (and 2): per-wave debug stamps + per-chunk costs
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
{
for cfg in 3 2; do
  echo "== config $cfg timing"; timeout 600 python tools/run_config.py --config $cfg --check none --reps 7
  echo "== config $cfg debug S=4"; FDCM_K2_DEBUG=1 timeout 300 python tools/run_config.py --config $cfg --check none --reps 2 2>&1 | grep "k2 debug" | tail -12
  echo "== config $cfg cost"; FDCM_K2_DUMP_COST=gpurun_out/cost_c$cfg.bin timeout 300 python tools/run_config.py --config $cfg --check none --reps 3 | tail -1
done
} > gpurun_out/r3_diag.log 2>&1
cat gpurun_out/r3_diag.log
