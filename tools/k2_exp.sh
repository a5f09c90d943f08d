#!/bin/bash
cd "$(dirname "$0")/.."
echo "== fuzz"; timeout 300 python tools/fuzz_parity.py 80 61 2>&1 | tail -1
echo "== fuzz S=5 redo"; FDCM_K2_SEGMENTS=5 FDCM_K2_FORCE_REDO=4 timeout 300 python tools/fuzz_parity.py 60 62 2>&1 | tail -1
echo "== fuzz S=2"; FDCM_K2_SEGMENTS=2 timeout 300 python tools/fuzz_parity.py 60 63 2>&1 | tail -1
FDCM_K2_SEGMENTS=4 FDCM_K2_DEBUG=1 timeout 120 python tools/run_config.py --config 2 --check none --reps 3 2>&1 | grep "k2 debug" | tail -3 | cut -c1-420
timeout 120 python tools/run_config.py --config 2 --check full --reps 9
FDCM_K2_SEGMENTS=4 timeout 300 python tools/run_config.py --config 3 --check full --reps 5
FDCM_K2_SEGMENTS=8 timeout 300 python tools/run_config.py --config 3 --check none --reps 5
