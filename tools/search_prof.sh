#!/bin/bash
# k_search alone (blocking frames) under rocprofv3 --kernel-trace; run through gpurun
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/sprof
python tools/run_config.py --config 2p --check none --reps 9 --search | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('search alone:', d['search'])"
rm -rf gpurun_out/sprof/r
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sprof/r -- python3 tools/run_config.py --config 2p --check none --reps 20 --search > gpurun_out/sprof/r.log 2>&1
f=$(find gpurun_out/sprof/r -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    print(f"  {row['Name'][:70]:70s} calls {row['Calls']:>4s} avg_us {float(row['AverageNs'])/1e3:9.1f}")
PY
