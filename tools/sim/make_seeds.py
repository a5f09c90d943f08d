#!/usr/bin/env python3
"""Seed images of a BASELINE config for tools/sim/k2_sim.cpp (diagnostic; uses the oracle as the rasteriser)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from openfdcm_amd import synthetic
from oracle import oracle as O
cfg = dict(synthetic.CONFIGS[sys.argv[1]])
scene = synthetic.scene(cfg["S"], cfg["scene_lines"], int(sys.argv[3]) if len(sys.argv) > 3 else 1)
orc = O.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=1, nthreads=8, stop_after=1)
with open(sys.argv[2], "wb") as f:
    np.array([orc.depth, orc.W, orc.H], dtype=np.int32).tofile(f)
    for k in range(orc.depth):
        (orc.slice(k).T == 0).astype(np.uint8).tofile(f)  # slice(k) is (H, W); file wants [x][y]
print("wrote", sys.argv[2])
