#!/bin/bash
# Where the wave cycles of each kernel go (SQ counters) and its L2 hit rate (TCC counters), one frame in flight.
# usage (inside gpurun): bash tools/pmc_sq.sh <out dir under gpurun_out>
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${1:-pmc_sq}
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
cd /tmp
pass() {  # pass <name> <counters...> : one rocprofv3 --pmc pass of the one-frame bench
  local name=$1; shift
  rm -rf /tmp/sq_$name
  rocprofv3 --kernel-trace --pmc "$@" -d /tmp/sq_$name -o sq --output-format csv -- python3 $R/bench.py --frames 1 --steps 5 --warmup 2 --cpu-sample 0 --single-frames 0 --api-frames 0 > /dev/null 2>&1
  cp "$(find /tmp/sq_$name -name '*counter_collection.csv' | head -1)" $OUT/$name.csv
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM
pass sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
FDCM_REPO=$R python3 - $OUT <<'PY'
import csv, sys, os, collections, json
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(os.listdir(out)):
    if not f.endswith('.csv'): continue
    for r in csv.DictReader(open(os.path.join(out, f))):
        k = r['Kernel_Name'].split('(')[0]
        res[k][r['Counter_Name']].append(float(r['Counter_Value']))
summ = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
import hashlib
so = os.path.join(os.environ.get('FDCM_REPO', '.'), 'openfdcm_amd', 'libfdcm_hip.so')
doc = {"source": "rocprofv3 --kernel-trace --pmc <set> (one pass per set) -- python3 bench.py --frames 1 --steps 5 --warmup 2 --cpu-sample 0 --single-frames 0 --api-frames 0; averages per launch",
       "so_sha256_16": hashlib.sha256(open(so, 'rb').read()).hexdigest()[:16] if os.path.exists(so) else None, "kernels": summ}
json.dump(doc, open(os.path.join(out, 'summary.json'), 'w'), indent=1)
for k, d in summ.items():
    if d.get('SQ_WAVE_CYCLES', 0) < 1e5: continue
    print(k, {c: round(v) for c, v in d.items()})
PY
