#!/bin/bash
# per-kernel times of the build at configs 2 and 3 (rocprofv3 --kernel-trace --stats), run through gpurun
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
for cfg in 2 3; do
  rm -rf gpurun_out/prof/c$cfg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/c$cfg -- python3 tools/run_config.py --config $cfg --check none --reps 20 > gpurun_out/prof/c$cfg.log 2>&1
  f=$(find gpurun_out/prof/c$cfg -name "*kernel_stats.csv" | head -1)
  echo "== config $cfg: $f"; cut -d, -f1-8 "$f" | head -14
done
