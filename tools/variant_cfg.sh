#!/bin/bash
# build times at configs 2 and 3 + the pipelined bench for variants/lib_*.so; run through gpurun
cd "$(dirname "$0")/.."
cp openfdcm_amd/libfdcm_hip.so /tmp/lib_orig.so
for v in variants/lib_*.so; do
  cp $v openfdcm_amd/libfdcm_hip.so
  echo "== $v"
  timeout 120 python tools/fuzz_parity.py 40 181 2>&1 | tail -1
  for c in 2 3; do python tools/run_config.py --config $c --check none --reps 9 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  config', d['config'], 'pass2 %.3f total %.3f' % (d['stage_ms']['pass2_ms'], d['kernels_ms']))"; done
  for i in 1 2; do python bench.py --steps 200 --warmup 20 --cpu-sample 0 --single-frames 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  bench %.2f M/s, %.3f ms/step' % (d['value']/1e6, d['ms_per_step']))"; done
done
cp /tmp/lib_orig.so openfdcm_amd/libfdcm_hip.so
