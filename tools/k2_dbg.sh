#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
{
echo "== fuzz default"; timeout 600 python tools/fuzz_parity.py 100 51 2>&1 | tail -2
echo "== fuzz S=2 forced redo"; FDCM_K2_SEGMENTS=7 FDCM_K2_FORCE_REDO=5 timeout 600 python tools/fuzz_parity.py 80 52 2>&1 | tail -2
for cfg in 2 3; do
  echo "== config $cfg parity + timing"; timeout 600 python tools/run_config.py --config $cfg --check full --reps 7
  for s in ${K2_DBG_SEGS:-4 8}; do
    echo "== config $cfg debug S=$s"; FDCM_K2_SEGMENTS=$s FDCM_K2_DEBUG=1 timeout 300 python tools/run_config.py --config $cfg --check none --reps 2 2>&1 | grep "k2 debug" | tail -2
  done
done
mkdir -p gpurun_out/prof
for cfg in 2 3; do
  rm -rf gpurun_out/prof/c$cfg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/c$cfg -- python3 tools/run_config.py --config $cfg --check none --reps 20 > gpurun_out/prof/c$cfg.log 2>&1
  f=$(find gpurun_out/prof/c$cfg -name "*kernel_stats.csv" | head -1)
  echo "== config $cfg: $f"; python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    print(f"  {row['Name'][:70]:70s} calls {row['Calls']:>4s} avg_us {float(row['AverageNs'])/1e3:9.1f}")
PY
done
} > gpurun_out/k2_dbg.log 2>&1
cat gpurun_out/k2_dbg.log
