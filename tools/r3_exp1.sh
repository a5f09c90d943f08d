#!/bin/bash
# round-3 experiment: segments x addend waves x launch order at config 3 (throughput bound) and 2 (latency bound)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
{
for cfg in 3 2; do
for s in 2 3 4; do for aw in 1 2 4; do for lpt in 1 0; do
  echo "== config $cfg S=$s AW=$aw LPT=$lpt"; FDCM_K2_LPT=$lpt FDCM_K2_SEGMENTS=$s FDCM_K2_AW=$aw timeout 300 python tools/run_config.py --config $cfg --check none --reps 9 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   pass2 %.3f total %.3f' % (d['stage_ms']['pass2_ms'], d['kernels_ms']))"
done; done; done; done
} > gpurun_out/r3_exp1.log 2>&1
cat gpurun_out/r3_exp1.log
