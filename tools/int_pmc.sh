#!/bin/bash
# memory-side counters of k_integral (separate rocprofv3 --pmc passes), run through gpurun
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/ipmc
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1)); rm -rf gpurun_out/ipmc/p$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/ipmc/p$i -- python3 tools/run_config.py --config ${INT_PMC_CFG:-2} --check none --reps 3 > gpurun_out/ipmc/p$i.log 2>&1
  f=$(find gpurun_out/ipmc/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][-40:]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    n[(k, row["Counter_Name"])] += 1
for k in acc:
    if any(t in k for t in ("k_integral", "k_propagate", "k_search")):
        print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
  [ -z "$f" ] && tail -3 gpurun_out/ipmc/p$i.log
done
