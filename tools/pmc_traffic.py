#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc runs (FETCH_SIZE and WRITE_SIZE, separate passes as MI355X_MICROARCH.md
prescribes) of `python3 bench.py` into profiles/<name>.json: HBM bytes per launch and kernel.

Units and gfx950 correction (MI355X_MICROARCH.md, section HBM): both counters are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so it is doubled
(`fetch_corrected`); WRITE_SIZE is exact.  For the gather-dominated k_search the doubling is not
calibrated and the raw value is kept beside it.

usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [command line that was profiled]
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("fdcm::"):
            continue
        fr, wr = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
        out[k] = {"fetch_raw_bytes": fr, "fetch_corrected_bytes": 2 * fr, "write_bytes": wr,
                  "hbm_bytes_per_launch": 2 * fr + wr}
    import hashlib
    import os
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "openfdcm_amd", "libfdcm_hip.so")
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- " + (sys.argv[4] if len(sys.argv) > 4 else
                         "python3 bench.py --frames 1 --steps 5 --warmup 2 --cpu-sample 0 --single-frames 0"),
               "so_sha256_16": hashlib.sha256(open(so, "rb").read()).hexdigest()[:16],
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
