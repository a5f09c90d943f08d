#!/bin/bash
# HBM traffic of k_integral by class of slice (inside gpurun): the lab build's FDCM_INT_ONLY=shallow|steep runs only the
# slices whose sweep goes along x (one wave per 58 chains, straight 1 KB loads) or along y (LDS tiles); FETCH_SIZE and
# WRITE_SIZE in separate rocprofv3 passes.  Prints KiB per launch; V of config 3 = 983 040 KiB (half of the slices per class).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export FDCM_LIBRARY=$PWD/openfdcm_amd/libfdcm_hip_lab.so
CFG=${INT_CFG:-3}
mkdir -p gpurun_out/isplit
for only in all shallow steep; do
  if [ $only = all ]; then unset FDCM_INT_ONLY; else export FDCM_INT_ONLY=$only; fi
  for set in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/isplit/${only}_$set; rm -rf $d
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 tools/run_config.py --config $CFG --check none --reps 3 > $d.log 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" "$only" "$set" <<'PY'
import csv, sys, collections
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "k_integral" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[3]]
print(f"k_integral {sys.argv[2]:8s} {sys.argv[3]:10s} KiB per launch: {sum(v) / max(1, len(v)):12.0f}  ({len(v)} launches)")
PY
    [ -z "$f" ] && tail -3 $d.log
  done
done
