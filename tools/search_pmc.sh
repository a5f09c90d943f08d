#!/bin/bash
# SQ counters of k_search at config 2' (separate rocprofv3 --pmc passes), run through gpurun
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/spmc
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_IFETCH" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf gpurun_out/spmc/p$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/spmc/p$i -- python3 tools/run_config.py --config 2p --check none --reps 3 --search > gpurun_out/spmc/p$i.log 2>&1
  f=$(find gpurun_out/spmc/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][-40:]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    n[(k, row["Counter_Name"])] += 1
for k in acc:
    if "k_search" in k:
        print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
  [ -z "$f" ] && tail -2 gpurun_out/spmc/p$i.log
done
