#!/bin/bash
# Is the driver's command (20 steps) within 10 % of the 200-step figure?  Run through gpurun.
cd "$(dirname "$0")/.."
for sf in ${SFS:-16}; do
  echo "== set-up frames $sf"
  for i in 1 2 3 4 5 6; do
    BENCH_SETUP_FRAMES=$sf python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --single-frames 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('  20 steps: %.1f M/s  %.3f ms/step  p50 %.2f p95 %.2f max %.2f' % (d['value']/1e6, d['ms_per_step'], d['frame_latency_ms']['p50'], d['frame_latency_ms']['p95'], d['frame_latency_ms']['max']))"
  done
done
python3 bench.py --gpus 1 --steps 200 --warmup 10 --cpu-sample 0 --single-frames 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(' 200 steps: %.1f M/s  %.3f ms/step  p50 %.2f p95 %.2f max %.2f' % (d['value']/1e6, d['ms_per_step'], d['frame_latency_ms']['p50'], d['frame_latency_ms']['p95'], d['frame_latency_ms']['max']))"
