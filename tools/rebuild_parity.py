#!/usr/bin/env python3
"""One feature-map handle rebuilt over a sequence of scenes (same size, changing content; then a different size and
back): every volume bit for bit against the oracle.  Exercises what only a reused handle has: buffers that are not
reallocated, the cleared-in-place seed bitmap, and the launch order taken from the previous build's chunk times
(FDCM_SWEEP_ORDER=1 forces it at small sizes).  usage: rebuild_parity.py [n_rebuilds] [seed]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openfdcm_amd import synthetic  # noqa: E402
from openfdcm_amd.engine import DeviceFeatureMap  # noqa: E402
from oracle import oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
for dist, depth in ((O.L2, 12), (O.L2_SQUARED, 7), (O.L1, 9)):
    dev = None
    for i in range(n):
        S = 320 if i % 5 != 3 else 192  # the fourth build changes the size: the history is dropped and taken up again
        scene = synthetic.scene(S, int(rng.integers(6, 60)), int(rng.integers(1, 1 << 30)))
        if dev is None:
            dev = DeviceFeatureMap.build(scene, depth=depth, coeff=5.0, padding=1.0, distance=dist)
        else:
            dev.rebuild(scene)
        orc = O.build(scene, depth=depth, coeff=5.0, padding=1.0, distance=dist, nthreads=8)
        a, b = dev.volume(), orc.volume()
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), ("volume", dist, i, S)
    dev.close()
print(f"{3 * n} rebuilds identical to the oracle")
