#!/bin/bash
# times variants/lib_*.so (drop-in builds of libfdcm_hip.so): run through gpurun.  VARIANT_INT=1: the line integral
# alone; default: bench.py (pipelined rate, blocking frame, blocking search span)
cd "$(dirname "$0")/.."
cp openfdcm_amd/libfdcm_hip.so /tmp/lib_orig.so
for v in variants/lib_*.so; do
  cp $v openfdcm_amd/libfdcm_hip.so
  echo "== $v"
  [ -n "$VARIANT_INT" ] && INT_CFGS="${INT_CFGS:-2 3}" INT_MODES="${INT_MODES:-0 1 2}" bash tools/int_exp.sh
  for i in 1 2; do python bench.py --steps 200 --warmup 20 --cpu-sample 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  bench %.2f M/s, %.3f ms/step, blocking frame %.3f ms, blocking search kernels %.3f ms' % (d['value']/1e6, d['ms_per_step'], d['single_frame_ms'], d['roofline_search']['avg_launch_ms']))"; done
done
cp /tmp/lib_orig.so openfdcm_amd/libfdcm_hip.so
