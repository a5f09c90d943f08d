import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, "/root/repo")
import numpy as np
from openfdcm_amd import synthetic, _capi
from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, FramePipeline, search_raw
rng = np.random.default_rng(3)
S0 = 300
tmpls = synthetic.templates(60, 14, S0, 5) + synthetic.templates(20, 3, S0, 6)
tset = DeviceTemplates(tmpls)
# frames of different sizes / line counts / emptiness: slots must regrow and shrink correctly
scenes = []
for i in range(40):
    S = int(rng.integers(64, 700)); n = int(rng.integers(0, 80))
    scenes.append(synthetic.scene(S, max(n, 2), 100 + i) if n > 0 else np.zeros((4, 0), np.float32))
ref = []
fm = None
for sc in scenes:
    fm = DeviceFeatureMap.build(sc, depth=17, coeff=5.0, padding=1.3, distance=_capi.L2)
    ref.append(np.array(search_raw(fm, tset, sc, 3, 5, _capi.BATCH_OPTIMIZE, 7), copy=True))
    fm.close()
for F in (1, 3, 5):
    pipe = FramePipeline(tset, depth=17, coeff=5.0, padding=1.3, distance=_capi.L2, max_tmpl_lines=3, max_scene_lines=5,
                         optimizer=_capi.BATCH_OPTIMIZE, batch_size=7, slots=F)
    for rep in range(3):
        tickets, got = [], []
        for sc in scenes:
            if len(tickets) == F: got.append(np.array(pipe.wait(tickets.pop(0)), copy=True))
            tickets.append(pipe.submit(sc))
        while tickets: got.append(np.array(pipe.wait(tickets.pop(0)), copy=True))
        bad = [i for i, (a, b) in enumerate(zip(got, ref)) if a.tobytes() != b.tobytes()]
        assert not bad, (F, rep, bad[:5])
    pipe.close()
    print("slots", F, "ok:", len(scenes) * 3, "frames identical to the blocking calls; matches per frame", [len(r) for r in ref[:8]])
