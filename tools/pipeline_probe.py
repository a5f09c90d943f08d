#!/usr/bin/env python3
"""Probe: steady-state frame time of the native frame pipeline for 1..F slots versus the blocking
rebuild -> search step; checks that every frame returns the blocking call's matches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openfdcm_amd import synthetic, _capi
from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, FramePipeline, search_raw

cfg = dict(synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "2p"])
maxF = int(sys.argv[2]) if len(sys.argv) > 2 else 4
scene = synthetic.scene(cfg["S"], cfg["scene_lines"], 1)
rec = _capi.as_records(scene)
tmpls = synthetic.templates(cfg["T"], cfg["n"], cfg["S"], 2)
tset = DeviceTemplates(tmpls)
fm = DeviceFeatureMap.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"])
K = 60
for _ in range(5):
    fm.rebuild(scene); ref = search_raw(fm, tset, scene, 4, 4, _capi.BATCH_OPTIMIZE, 10)
t0 = time.perf_counter()
for _ in range(K):
    fm.rebuild(scene); m = search_raw(fm, tset, scene, 4, 4, _capi.BATCH_OPTIMIZE, 10)
print("blocking    ms/frame %.4f  matches %d" % ((time.perf_counter() - t0) / K * 1e3, len(m)))
for F in range(1, maxF + 1):
    pipe = FramePipeline(tset, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"], slots=F)
    def run(n, check=False):
        tickets = []
        for k in range(n):
            tickets.append(pipe.submit(rec, prepared=True))
            if len(tickets) == F:
                r = pipe.wait(tickets.pop(0))
                if check: assert r.tobytes() == ref.tobytes()
        while tickets:
            r = pipe.wait(tickets.pop(0))
            if check: assert r.tobytes() == ref.tobytes()
    run(3 * F, check=True)
    t0 = time.perf_counter(); run(K)
    print("pipeline F=%d ms/frame %.4f" % (F, (time.perf_counter() - t0) / K * 1e3))
    pipe.close()
