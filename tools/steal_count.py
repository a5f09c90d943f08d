import sys, ctypes, json
sys.path.insert(0, "/root/repo")
import numpy as np
from openfdcm_amd import synthetic, _capi
from openfdcm_amd.engine import DeviceFeatureMap
cfgname = sys.argv[1]
cfg = dict(synthetic.CONFIGS[cfgname])
scenes = [synthetic.scene(cfg["S"], cfg["scene_lines"], s) for s in (1, 2, 3, 4)]
dev = DeviceFeatureMap.build(scenes[0], depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"])
c = ctypes.c_int64()
out = []
for sc in scenes:
    ts = []
    _capi.check(_capi.lib().fdcm_selftest_sweep_steals(dev._h, ctypes.byref(c))); c0 = c.value
    for _ in range(6):
        dev.rebuild(sc); ts.append(dev.build_timing()["pass2_ms"])
    _capi.check(_capi.lib().fdcm_selftest_sweep_steals(dev._h, ctypes.byref(c)))
    out.append((round(float(np.median(ts)), 4), (c.value - c0) // 6))
print(cfgname, out, "mean", round(sum(o[0] for o in out) / 4, 4))
