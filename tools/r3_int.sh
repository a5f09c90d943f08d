#!/bin/bash
# round-3: steep line integral with 60 / 124 / 252 own chains per block (FDCM_INT_XC=64/128/256); parity + timing
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
{
for xc in 128 64 256; do
  echo "== fuzz XC=$xc"; FDCM_INT_XC=$xc timeout 600 python tools/fuzz_parity.py 40 71 2>&1 | tail -2
  for cfg in 2 3; do
    echo "== config $cfg XC=$xc"; FDCM_INT_XC=$xc timeout 600 python tools/run_config.py --config $cfg --check full --reps 9 | cut -c1-330
  done
  echo "== config 5 XC=$xc"; FDCM_INT_XC=$xc timeout 900 python tools/run_config.py --config 5 --check none --reps 5 | cut -c1-330
done
for cfg in 2 3 5; do echo "== config $cfg default"; timeout 900 python tools/run_config.py --config $cfg --check none --reps 7 | cut -c1-330; done
} > gpurun_out/r3_int.log 2>&1
cat gpurun_out/r3_int.log
