#!/bin/bash
# kernel durations with the frame pipeline running (4 frames in flight); run through gpurun
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=8 TMPDIR=/tmp
rm -rf gpurun_out/pprof; mkdir -p gpurun_out/pprof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pprof/r -- python3 bench.py --cpu-sample 0 --single-frames 0 --steps 200 --warmup 20 > gpurun_out/pprof/r.log 2>&1
f=$(find gpurun_out/pprof/r -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
tot = 0
rows = list(csv.DictReader(open(sys.argv[1])))
for row in rows:
    print(f"  {row['Name'][:64]:64s} calls {row['Calls']:>5s} avg_us {float(row['AverageNs'])/1e3:8.1f} total_ms {float(row['TotalDurationNs'])/1e6:8.1f}")
PY
tail -1 gpurun_out/pprof/r.log | cut -c1-200
