#!/usr/bin/env python3
"""Soak of the L2 sweep's dynamic cuts at full size: scenes of many seeds at config 2 (and a few at config 3), each built
several times on one handle (every build is cut differently: the cuts depend on timing) and compared with the oracle's
volume bit for bit.  usage: soak_volumes.py [first seed = 5] [seeds = 12] [builds per scene = 3] [config3 seeds = 2]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openfdcm_amd import _capi, synthetic  # noqa: E402
from openfdcm_amd.engine import DeviceFeatureMap  # noqa: E402
from oracle import oracle as O  # noqa: E402

s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 12
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n3 = int(sys.argv[4]) if len(sys.argv) > 4 else 2
nt = min(64, os.cpu_count() or 1)
t0 = time.time()
total = taken = 0
for cfgname, seeds, dists in (("2", range(s0, s0 + ns), (O.L2, O.L2_SQUARED)), ("3", range(s0, s0 + n3), (O.L2_SQUARED,))):
    cfg = dict(synthetic.CONFIGS[cfgname])
    for dist in dists:
        dev = None
        for seed in seeds:
            scene = synthetic.scene(cfg["S"], cfg["scene_lines"], seed)
            orc = O.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=dist, nthreads=nt)
            want = orc.volume().view(np.uint32)
            for r in range(reps):
                if dev is None:
                    dev = DeviceFeatureMap.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=dist)
                else:
                    dev.rebuild(scene)
                got = dev.volume().view(np.uint32)
                assert got.shape == want.shape and np.array_equal(got, want), (cfgname, dist, seed, r, int(np.sum(got != want)))
                total += 1
        c = ctypes.c_int64()
        _capi.check(_capi.lib().fdcm_selftest_sweep_steals(dev._h, ctypes.byref(c)))
        taken += c.value
        dev.close()
print(f"{total} full-size builds identical to the oracle, {taken} ranges taken over, {time.time() - t0:.0f} s")
