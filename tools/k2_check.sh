#!/bin/bash
# K2 parity + timing on the GPU box (run through gpurun).  Every step has its own timeout.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== staged + config builds (default segments)"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "staged or kat or config2_build or config3 or golden or config1" 2>&1 | tail -5
echo "== fuzz default"; timeout 600 python tools/fuzz_parity.py 150 21 2>&1 | tail -2
for s in 1 2 3 4; do echo "== fuzz FDCM_K2_SEGMENTS=$s"; FDCM_K2_SEGMENTS=$s timeout 600 python tools/fuzz_parity.py 100 3$s 2>&1 | tail -2; done
echo "== fuzz forced redo"; FDCM_K2_FORCE_REDO=3 timeout 600 python tools/fuzz_parity.py 60 41 2>&1 | tail -2
echo "== fuzz legacy"; FDCM_K2_LEGACY=1 timeout 600 python tools/fuzz_parity.py 40 42 2>&1 | tail -2
echo "== timing config 2"; timeout 300 python tools/run_config.py --config 2 --check full --reps 7
echo "== timing config 2 legacy"; FDCM_K2_LEGACY=1 timeout 300 python tools/run_config.py --config 2 --check none --reps 7
echo "== timing config 3"; timeout 600 python tools/run_config.py --config 3 --check full --reps 5
echo "== timing config 3 legacy"; FDCM_K2_LEGACY=1 timeout 300 python tools/run_config.py --config 3 --check none --reps 5
} > gpurun_out/k2_check.log 2>&1
tail -60 gpurun_out/k2_check.log
