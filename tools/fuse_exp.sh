#!/bin/bash
cd "$(dirname "$0")/.."
echo "== fuzz unfused blocking"; FDCM_K2_UNFUSED=1 timeout 300 python tools/fuzz_parity.py 40 91 | tail -1
for v in fused three; do
  if [ $v = three ]; then export FDCM_K2_UNFUSED=1; else unset FDCM_K2_UNFUSED; fi
  for i in 1 2 3; do python3 bench.py --steps 150 --warmup 10 --cpu-sample 0 --single-frames 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  pipelined $v: %.1f M/s %.3f ms/step' % (d['value']/1e6, d['ms_per_step']))"; done
done
unset FDCM_K2_UNFUSED
python3 bench.py --steps 20 --warmup 5 --cpu-sample 50 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['ms_per_step'], d['single_frame_ms'], d['roofline']['frac'], d['roofline']['stages']['pass2']['ms'], d['parity_gate'])"
