#!/bin/bash
# search kernel span per process for variants/lib_*.so; run through gpurun
cd "$(dirname "$0")/.."
cp openfdcm_amd/libfdcm_hip.so /tmp/lib_orig.so
for v in variants/lib_*.so; do
  cp $v openfdcm_amd/libfdcm_hip.so
  echo "== $v"
  timeout 120 python tools/fuzz_parity.py 30 91 2>&1 | tail -1
  for n in 0 4 0 4 0 4; do python tools/search_cold.py $n | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ', d['rebuild + search'])"; done
done
cp /tmp/lib_orig.so openfdcm_amd/libfdcm_hip.so
