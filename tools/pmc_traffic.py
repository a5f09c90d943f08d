#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc runs (FETCH_SIZE and WRITE_SIZE, separate passes as MI355X_MICROARCH.md
prescribes) of `python3 bench.py` into profiles/<name>.json: HBM bytes per launch and kernel.

Units and gfx950 correction (MI355X_MICROARCH.md, section HBM): both counters are in KiB; WRITE_SIZE is exact.
FETCH_SIZE = memory-side read requests x 64 B, and a request is 64 B or 128 B: calibrated on this GPU with
tools/fetch_calib.hip (profiles/r03_fetch_calib.json) --
  * consecutive addresses, 4 B or 16 B per lane (>= 128 B contiguous per wave operation): 128-B requests, the
    counter reads exactly 1/2 of the bytes -> x2 (the guide's rule);
  * 4-byte gathers that touch one 64-B sector of a 128-B line: 64-B requests, the counter is exact -> x1;
  * 4-byte gathers that touch both sectors of a 128-B line in one wave operation: one 128-B request -> x2.
There is no counter that separates the two request sizes, so every kernel gets a pattern class: "stream" kernels
(whole-unit loads of >= 128 contiguous bytes per wave operation) are doubled; for "gather" / "mixed" kernels the truth
lies between the raw value (all requests 64 B) and twice it (all 128 B), both are reported and `fetch_corrected_bytes`
is the upper bound.

usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [command line that was profiled]
"""
import collections
import csv
import glob
import json
import sys


def device_code_hash(path):
    """sha256[:16] of the library's .hip_fatbin section (the gfx950 code objects); the same function as in bench.py."""
    import hashlib
    import struct
    data = open(path, "rb").read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        return None
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)

    def sh(i):
        name, typ, flags, addr, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
        return name, off, size
    _, stroff, strsize = sh(shstrndx)
    for i in range(shnum):
        name, off, size = sh(i)
        end = data.index(b"\0", stroff + name)
        if data[stroff + name:end] == b".hip_fatbin":
            return hashlib.sha256(data[off:off + size]).hexdigest()[:16]
    return None


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
    # access pattern of each kernel's HBM reads (see the module docstring)
    patterns = {"k_search": "gather", "k_evaluate": "gather", "k_sweep": "mixed",
                "k_pass2_l2": "mixed", "k_topk": "mixed"}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("fdcm::"):
            continue
        fr, wr = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
        pat = next((v for n, v in patterns.items() if n in k), "stream")
        out[k] = {"read_pattern": pat, "fetch_raw_bytes": fr, "fetch_lower_bytes": fr if pat != "stream" else 2 * fr,
                  "fetch_corrected_bytes": 2 * fr, "write_bytes": wr, "hbm_bytes_per_launch": 2 * fr + wr,
                  "hbm_bytes_per_launch_lower": (fr if pat != "stream" else 2 * fr) + wr}
    import hashlib
    import os
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "openfdcm_amd", "libfdcm_hip.so")
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- " + (sys.argv[4] if len(sys.argv) > 4 else
                         "python3 bench.py --frames 1 --steps 5 --warmup 2 --cpu-sample 0 --single-frames 0"),
               "so_sha256_16": hashlib.sha256(open(so, "rb").read()).hexdigest()[:16],
               "device_sha256_16": device_code_hash(so),  # of the .hip_fatbin section: host-only changes of the library keep the summary valid
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
