#!/bin/bash
# times variants/lib_*.so (drop-in builds of libfdcm_hip.so) on the integral and the bench; run through gpurun
cd "$(dirname "$0")/.."
cp openfdcm_amd/libfdcm_hip.so /tmp/lib_orig.so
for v in variants/lib_*.so; do
  cp $v openfdcm_amd/libfdcm_hip.so
  echo "== $v"
  INT_CFGS="${INT_CFGS:-2 3}" INT_MODES="${INT_MODES:-0 1}" bash tools/int_exp.sh
  [ -n "$VARIANT_BENCH" ] && python bench.py --steps 200 --warmup 20 --cpu-sample 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['value']/1e6,2), 'M/s', d['ms_per_step'], 'single', d['single_frame_ms'])"
done
cp /tmp/lib_orig.so openfdcm_amd/libfdcm_hip.so
