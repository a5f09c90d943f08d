#!/usr/bin/env python3
"""Randomised parity run on the GPU: random scenes, depths, distances, paddings, optimisers and template
sets against the CPU oracle (volume bit for bit, match list bit for bit).  Diagnostic; imports oracle/."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from openfdcm_amd import synthetic, _capi
from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, search_raw
from oracle import oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed0)
t0 = time.time()
steals = 0  # ranges the L2 sweep's waves took over from slower ones (dynamic cuts), summed over the cases
import ctypes
for case in range(n_cases):
    S = int(rng.choice([33, 64, 97, 150, 256, 300, 511, 640]))
    n_lines = int(rng.integers(2, 120))
    depth = int(rng.choice([1, 2, 7, 16, 30, 45, 60]))
    dist = int(rng.integers(0, 3))
    padding = float(rng.choice([1.0, 1.3, 2.2]))
    coeff = float(rng.choice([0.5, 5.0, 50.0]))
    scene = np.array(synthetic.scene(S, n_lines, int(rng.integers(1, 1 << 30))), dtype=np.float32)
    if rng.random() < 0.3:  # axis-aligned and degenerate lines, duplicates
        k = int(rng.integers(0, scene.shape[1]))
        scene[:, k] = [scene[0, k], scene[1, k], scene[0, k], scene[1, k] + 20]
        scene = np.concatenate([scene, scene[:, :2], np.array([[5], [5], [5], [5]], np.float32)], axis=1)
    T = int(rng.integers(1, 60))
    tmpls = []
    for t in range(T):
        n = int(rng.choice([0, 1, 2, 5, 13, 32, 40]))
        tmpls += synthetic.templates(1, n, S, int(rng.integers(1, 1 << 30))) if n else [np.zeros((4, 0), np.float32)]
    kind = int(rng.integers(0, 3))
    batch = int(rng.choice([1, 3, 10, 25])) if kind != 0 else 1
    maxT, maxS = int(rng.integers(1, 6)), int(rng.integers(1, 6))
    dev = DeviceFeatureMap.build(scene, depth=depth, coeff=coeff, padding=padding, distance=dist)
    orc = O.build(scene, depth=depth, coeff=coeff, padding=padding, distance=dist, nthreads=8)
    a, b = dev.volume(), orc.volume()
    assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), ("volume", case, S, n_lines, depth, dist, padding)
    got = search_raw(dev, DeviceTemplates(tmpls), scene, maxT, maxS, kind, batch)
    want = O.search(orc, tmpls, scene, maxT, maxS, kind=kind, batch=batch, nthreads=8).astype(_capi.MATCH_DTYPE)
    assert got.tobytes() == want.tobytes(), ("matches", case, S, n_lines, depth, dist, kind, batch, maxT, maxS, len(got), len(want))
    if rng.random() < 0.5:  # the handle rebuilt with another scene of another size (the ranges are cut anew by whoever runs dry first)
        scene2 = np.array(synthetic.scene(int(rng.choice([64, 200, 333])), int(rng.integers(2, 80)), int(rng.integers(1, 1 << 30))), dtype=np.float32)
        dev.rebuild(scene2)
        orc2 = O.build(scene2, depth=depth, coeff=coeff, padding=padding, distance=dist, nthreads=8)
        a, b = dev.volume(), orc2.volume()
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), ("rebuilt volume", case, depth, dist, padding)
    c = ctypes.c_int64()
    _capi.check(_capi.lib().fdcm_selftest_sweep_steals(dev._h, ctypes.byref(c)))
    steals += c.value
    dev.close()
print(f"{n_cases} random cases identical to the oracle (seed {seed0}), {steals} ranges taken over, {time.time() - t0:.0f} s")
