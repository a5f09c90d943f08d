import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openfdcm_amd import synthetic
from openfdcm_amd.engine import DeviceFeatureMap
from oracle import oracle as O
"""Which build stage first differs from the oracle: sampled slices of a BASELINE config after stage 1 (transforms), 2
(propagation), 3 (line integral); diagnostic (found the 16-byte-store hazard at config 5).  usage: stage_parity.py <config> [depth] [distance 0|1|2]"""
name = sys.argv[1]
cfg = dict(synthetic.CONFIGS[name])
depth = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["depth"]
if len(sys.argv) > 3:
    cfg["distance"] = int(sys.argv[3])
scene = synthetic.scene(cfg["S"], cfg["scene_lines"], 1)
for stop in (1, 2, 3):
    dev = DeviceFeatureMap.build(scene, depth=depth, coeff=5.0, padding=1.0, distance=cfg["distance"], stop_after=stop)
    orc = O.build(scene, depth=depth, coeff=5.0, padding=1.0, distance=cfg["distance"], nthreads=os.cpu_count(), stop_after=stop)
    bad = {}
    for k in sorted(set(np.linspace(0, depth - 1, 5).astype(int))):
        a, b = dev.slice(int(k)), orc.slice(int(k))
        d = a.view(np.uint32) != b.view(np.uint32)
        if d.any():
            ys, xs = np.nonzero(d)
            bad[int(k)] = (int(d.sum()), int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max()))
            if stop == 1 and len(bad) == 1:
                print("  slice", k, "x%4 hist", np.bincount(xs % 4, minlength=4), "first:", [(int(x), int(y), float(a[y, x]), float(b[y, x])) for y, x in list(zip(ys, xs))[:12]])
                rows = np.unique(ys)
                print("  rows", len(rows), rows[:20], "cols per row of first row:", xs[ys == rows[0]][:40])
    print("stop", stop, "depth", depth, "bad slices (count, xmin, xmax, ymin, ymax):", bad)
    dev.close()
