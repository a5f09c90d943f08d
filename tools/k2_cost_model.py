#!/usr/bin/env python3
"""Per-chunk sweep times of a config (FDCM_K2_DUMP_COST) against features of the chunk's seed picture, to choose the
launch-order estimate of k_cost.  Run on the GPU box: python tools/k2_cost_model.py 3"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfgname = sys.argv[1] if len(sys.argv) > 1 else "3"
dump = "/tmp/k2_cost.bin"
os.environ["FDCM_K2_DUMP_COST"] = dump
from openfdcm_amd import synthetic  # noqa: E402
from openfdcm_amd.engine import DeviceFeatureMap  # noqa: E402
from oracle import oracle as O  # noqa: E402
cfg = dict(synthetic.CONFIGS[cfgname])
scene = synthetic.scene(cfg["S"], cfg["scene_lines"], 1)
dev = DeviceFeatureMap.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"])
dev.rebuild(scene)
cost = np.fromfile(dump, dtype=np.int32).astype(np.float64) / 100.0  # us
orc = O.build(scene, depth=cfg["depth"], coeff=5.0, padding=1.0, distance=cfg["distance"], nthreads=64, stop_after=1)
vol = orc.volume()  # [k][x][y], 0 at seeds
m, W, H = vol.shape
nch = (H + 63) // 64
assert cost.size == m * nch, (cost.size, m, nch)
seed = vol == 0
feats = []
for k in range(m):
    col_has = seed[k].any(axis=1)                     # seeded columns of the slice
    n_seeded = int(col_has.sum())
    ys = [np.flatnonzero(seed[k, x]) for x in range(W)]
    for c in range(nch):
        y0, y1 = c * 64, min(H, c * 64 + 64) - 1
        inside = seed[k, :, y0:y1 + 1].any(axis=1)
        n_far = int((col_has & ~inside).sum())
        # distance from the chunk to the nearest seed of each seeded column (0 when inside)
        lb = []
        for x in np.flatnonzero(col_has):
            yy = ys[x]
            d = np.where(yy < y0, y0 - yy, np.where(yy > y1, yy - y1, 0))
            lb.append(d.min())
        lb = np.array(lb, dtype=np.float64) if lb else np.zeros(1)
        n_in_px = int(seed[k, :, y0:y1 + 1].sum())      # seed pixels inside the chunk (~ owner entries of its rows)
        rows_px = seed[k, :, y0:y1 + 1].sum(axis=0)
        feats.append((n_seeded, n_far, lb.mean(), np.minimum(lb, 256).sum(), (lb > 64).sum(), (lb > 128).sum(), (lb > 256).sum(),
                      int(inside.sum()), n_in_px, int(rows_px.max())))
F = np.array(feats, dtype=np.float64)
names = ["n_seeded", "n_far", "mean_lb", "sum_min(lb,256)", "n(lb>64)", "n(lb>128)", "n(lb>256)", "n_inside_cols", "n_inside_px", "max_row_px"]
print("chunks", cost.size, "cost us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (cost.mean(), *np.percentile(cost, [50, 90, 99]), cost.max()))
for i, n in enumerate(names):
    r = np.corrcoef(F[:, i], cost)[0, 1]
    print("  corr(cost, %-16s) = %.3f" % (n, r))
A = np.column_stack([F, np.ones(len(F))])
coef, *_ = np.linalg.lstsq(A, cost, rcond=None)
pred = A @ coef
print("  least squares:", dict(zip(names + ["1"], np.round(coef, 4))), "corr %.3f" % np.corrcoef(pred, cost)[0, 1])
for cols in ([0, 1], [0, 3], [0, 1, 3], [0, 4, 5, 6], [0, 1, 7], [0, 1, 8], [0, 1, 9], [0, 1, 7, 9]):
    A2 = np.column_stack([F[:, cols], np.ones(len(F))])
    c2, *_ = np.linalg.lstsq(A2, cost, rcond=None)
    print("  fit on", [names[j] for j in cols], np.round(c2, 4), "corr %.3f" % np.corrcoef(A2 @ c2, cost)[0, 1])
top = np.argsort(-cost)[:12]
for t in top:
    print("   chunk %5d slice %3d c %2d cost %7.1f us feats %s" % (t, t // nch, t % nch, cost[t], np.round(F[t], 1)))
