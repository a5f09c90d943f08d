#!/usr/bin/env python3
"""k_search on BASELINE config 2' with the scene lines of the SEARCH restricted to one direction (diagnostic tool).
The volume is the config's own; the candidates' translations walk along the scene lines they are aligned to, so the scene
handed to the search decides the direction of the walk: 'x' = lines within 15 degrees of the x axis (steps with
|savx| = 1), 'y' = within 15 degrees of the y axis (|savy| = 1), 'd' = diagonal (30 - 60 degrees), 'all' = the config's scene.
Prints one JSON line: kernel time of the search, candidates, translations evaluated.
  usage: search_dirs.py x|y|d|all [reps]"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def directed_scene(S, N, seed, lo_deg, hi_deg):
    from openfdcm_amd.synthetic import SplitMix64, _line
    rng = SplitMix64(seed)
    out = np.zeros((4, N), dtype=np.float64)
    hi = S - 1.0
    for i in range(N):
        while True:
            cx, cy = rng.u01() * hi, rng.u01() * hi
            length = S / 32.0 + rng.u01() * (S / 4.0 - S / 32.0)
            angle = math.radians(lo_deg + rng.u01() * (hi_deg - lo_deg))
            x1, y1, x2, y2 = _line(cx, cy, length, angle)
            if 0.0 <= x1 <= hi and 0.0 <= y1 <= hi and 0.0 <= x2 <= hi and 0.0 <= y2 <= hi:
                out[:, i] = (x1, y1, x2, y2)
                break
    return out.astype(np.float32)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    from openfdcm_amd import synthetic, _capi
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, search_raw
    cfg = dict(synthetic.CONFIGS["2p"])
    scene = synthetic.scene(cfg["S"], cfg["scene_lines"], 1)
    tmpls = synthetic.templates(cfg["T"], cfg["n"], cfg["S"], 2)
    dev = DeviceFeatureMap.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"])
    ts = DeviceTemplates(tmpls)
    rng = {"x": (-15.0, 15.0), "y": (75.0, 105.0), "d": (30.0, 60.0)}
    sc = scene if which == "all" else directed_scene(cfg["S"], cfg["scene_lines"], 7, *rng[which])
    ks, n, ev = [], 0, 0
    for _ in range(reps):
        got = search_raw(dev, ts, sc, 4, 4, _capi.BATCH_OPTIMIZE, 10)
        st = dev.search_timing()
        ks.append(st["kernel_ms"]); n = len(got); ev = st["evaluations"]
    print(json.dumps({"scene": which, "search_kernels_ms": float(np.median(ks)), "matches": n, "evaluations": int(ev),
                      "us_per_1k_evaluations": float(np.median(ks)) * 1e3 / (ev / 1e3)}))


if __name__ == "__main__":
    main()
