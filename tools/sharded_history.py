#!/usr/bin/env python3
"""A frame slot of the sharded engine keeps the L2 sweep's launch-order history although its buffers are reserved
before every frame (fdcm_sharded_submit -> run_build(reserve_only)); and a slot whose first frame cannot be built
(feature size above 16384) serves the next frame.  Run with FDCM_SWEEP_ORDER=1 so that these small builds take a launch
order at all; prints one line per check and exits non-zero on a failure (tests/test_gpu_parity.py)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from openfdcm_amd import _capi, synthetic  # noqa: E402
from openfdcm_amd.engine import ShardedEngine  # noqa: E402
from oracle import oracle as O  # noqa: E402


def counts():
    h, p = C.c_int64(), C.c_int64()
    _capi.check(_capi.lib().fdcm_selftest_sweep_order_counts(C.byref(h), C.byref(p)))
    return h.value, p.value


S = 256
tmpls = synthetic.templates(24, 11, S, 5)
scenes = [synthetic.scene(S, 40 + 5 * i, 70 + i) for i in range(3)]
wants = []
for sc in scenes:
    orc = O.build(sc, depth=12, coeff=5.0, padding=1.0, distance=O.L2, nthreads=4)
    wants.append(np.asarray(O.search(orc, tmpls, sc, 4, 4, kind=O.BATCH_OPTIMIZE, batch=10, nthreads=4)))

eng = ShardedEngine(tmpls, n_devices=1, depth=12, coeff=5.0, padding=1.0, distance=O.L2)
h0, p0 = counts()
for i in range(7):
    got = eng.search(scenes[i % 3], 4, 4, _capi.BATCH_OPTIMIZE, 10)
    assert got.tobytes() == wants[i % 3].tobytes(), f"frame {i} differs from the oracle"
h1, p1 = counts()
print(f"7 frames of one slot: {p1 - p0} launch order(s) from the proxy, {h1 - h0} from the previous build")
assert (p1 - p0, h1 - h0) == (1, 6), "the slot lost its cost history between frames"

# a first frame that cannot be built, then a normal one on the same (empty) handle
eng2 = ShardedEngine(tmpls, n_devices=1, depth=12, coeff=5.0, padding=1.0, distance=O.L2)
huge = np.array([[0.0], [0.0], [30000.0], [100.0]], dtype=np.float32)
try:
    eng2.search(huge, 4, 4)
    raise SystemExit("a feature size above 16384 was accepted")
except _capi.FdcmError as e:
    assert "16384" in str(e), str(e)
    print("unbuildable first frame reported:", str(e)[:90])
got = eng2.search(scenes[1], 4, 4, _capi.BATCH_OPTIMIZE, 10)
assert got.tobytes() == wants[1].tobytes(), "the frame after the failed one differs from the oracle"
got = eng2.search(scenes[2], 4, 4, _capi.BATCH_OPTIMIZE, 10)
assert got.tobytes() == wants[2].tobytes()
print("frames after the failed one identical to the oracle")
eng.close()
eng2.close()
print("sharded history ok")
