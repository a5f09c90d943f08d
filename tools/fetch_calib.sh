#!/bin/bash
# Calibrate FETCH_SIZE per access pattern (run inside gpurun): builds tools/fetch_calib.hip, profiles it, writes
# gpurun_out/fetch_calib.json = {pattern: {needed_bytes, fetch_size_bytes, factor = needed / counter}}
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $R/tools/fetch_calib.hip || exit 1
/tmp/fetch_calib > /tmp/fetch_calib_needed.json || exit 1
for c in FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; do
  rm -rf /tmp/fc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/fc_$c -o fc --output-format csv -- /tmp/fetch_calib > /dev/null 2>&1
done
python3 - <<'PY' > $R/gpurun_out/fetch_calib.json
import collections, csv, glob, json
need = json.load(open("/tmp/fetch_calib_needed.json"))
out = {}
for c in ["FETCH_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"]:
    fs = glob.glob(f"/tmp/fc_{c}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if k.startswith("calib_") and k != "calib_flush":
            out.setdefault(k, {})[c] = sum(v) / len(v)
for k, d in out.items():
    d["needed_bytes"] = need["needed_bytes"][k] + need["index_bytes"].get(k, 0)
    if "FETCH_SIZE" in d:
        d["fetch_size_bytes"] = d["FETCH_SIZE"] * 1024
        d["factor_needed_over_counter"] = d["needed_bytes"] / d["fetch_size_bytes"]
print(json.dumps(out, indent=1))
PY
cat $R/gpurun_out/fetch_calib.json
