#!/usr/bin/env python3
"""Search kernel span at config 2' after a rebuild (the volume was just written, the caches hold the build's last
traffic) and without one (the previous search's gathers are still cached); with extra feature maps allocated first
to see whether the placement of the volume matters.  Run on the GPU box."""
import json
import sys
import os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openfdcm_amd import synthetic, _capi
from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, search_raw

cfg = dict(synthetic.CONFIGS["2p"])
scene = synthetic.scene(cfg["S"], cfg["scene_lines"], 1)
tset = DeviceTemplates(synthetic.templates(cfg["T"], cfg["n"], cfg["S"], 2))
extra = [DeviceFeatureMap.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"])
         for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0)]
dev = DeviceFeatureMap.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"])
out = {}
for name, rebuild in (("search only", False), ("rebuild + search", True), ("search only again", False)):
    ts = []
    for _ in range(15):
        if rebuild:
            dev.rebuild(scene)
        search_raw(dev, tset, scene, 4, 4, _capi.BATCH_OPTIMIZE, 10)
        ts.append(dev.search_timing()["kernel_ms"])
    out[name] = [round(float(np.median(ts)), 3), round(float(np.min(ts)), 3), round(float(np.max(ts)), 3)]
print(json.dumps(out))
