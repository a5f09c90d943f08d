#!/usr/bin/env python3
"""Lab build: phase times of k_sweep_balanced for one scene of a config (FDCM_SWEEP_LAB=1 output of the last build).
usage: FDCM_LIBRARY=.../libfdcm_hip_lab.so FDCM_SWEEP_LAB=1 python tools/lab_probe.py <config> <seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openfdcm_amd import synthetic
from openfdcm_amd.engine import DeviceFeatureMap
cfg = dict(synthetic.CONFIGS[sys.argv[1]])
scene = synthetic.scene(cfg["S"], cfg["scene_lines"], int(sys.argv[2]))
dev = DeviceFeatureMap.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"])
for _ in range(2):
    dev.rebuild(scene)
sys.stderr.write("==== last build\n")
dev.rebuild(scene)
print(dev.build_timing())
