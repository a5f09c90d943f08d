#!/bin/bash
# SQ counters of the K2 kernels at config 2 (separate rocprofv3 --pmc passes), run through gpurun
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > gpurun_out/pmc/sq_counters.txt
wc -l gpurun_out/pmc/sq_counters.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_IFETCH" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1)); rm -rf gpurun_out/pmc/p$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc/p$i -- python3 tools/run_config.py --config ${K2_PMC_CFG:-2} --check none --reps 3 > gpurun_out/pmc/p$i.log 2>&1
  f=$(find gpurun_out/pmc/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][-40:]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    n[(k, row["Counter_Name"])] += 1
for k in acc:
    if any(t in k for t in ("k_env", "k_addend", "k_fill", "k_sweep", "k_integral", "k_propagate")):
        print(k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()})
PY
done
