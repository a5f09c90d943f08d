#!/usr/bin/env python3
"""Which kernel of the build is slower when the scene changes (diagnostic)?  Per-stage HIP-event times of config 2 builds of
scene B right after scene B (same) and right after scene A (new), for a few pairs of seeds; and of B after a jittered B."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openfdcm_amd import synthetic  # noqa: E402
from openfdcm_amd.engine import DeviceFeatureMap  # noqa: E402

cfg = dict(synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "2"])
scenes = [synthetic.scene(cfg["S"], cfg["scene_lines"], s) for s in (1, 2, 3, 4)]
dev = DeviceFeatureMap.build(scenes[0], depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"])
keys = ("pass1_ms", "pass2_ms", "propagate_ms", "integral_ms", "span_ms")
res = {}
for bi, B in enumerate(scenes):
    A = scenes[(bi + 1) % 4]
    same, new = [], []
    for _ in range(12):
        dev.rebuild(B); dev.rebuild(B); dev.rebuild(B)
        same.append(dev.build_timing())
        dev.rebuild(A); dev.rebuild(B)
        new.append(dev.build_timing())
    res[f"seed{bi + 1}"] = {"same": {k: round(float(np.median([t[k] for t in same])), 4) for k in keys},
                            "after_another": {k: round(float(np.median([t[k] for t in new])), 4) for k in keys}}
print(json.dumps(res, indent=1))
