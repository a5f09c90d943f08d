"""Deterministic synthetic scenes and templates (SURVEY.md section 8d / BASELINE.md section 4).

SplitMix64 stream, u01() = (next() >> 40) * 2^-24; geometry in float64, rounded once to float32.
Draw order per line: centre x, centre y, length, angle.
"""
import math

import numpy as np

_M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _M64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def u01(self):
        return (self.next() >> 40) * (2.0 ** -24)


def _line(cx, cy, length, angle):
    dx, dy = 0.5 * length * math.cos(angle), 0.5 * length * math.sin(angle)
    return cx - dx, cy - dy, cx + dx, cy + dy


def scene(S, N, seed):
    """(4, N) float32.  Two anchor lines force the bounding box to [0, S-1]^2, so with padding 1.0
    the feature size is exactly S x S and the scene translation is (0, 0)."""
    rng = SplitMix64(seed)
    out = np.zeros((4, N), dtype=np.float64)
    out[:, 0] = (0.0, 0.0, S / 8.0, 0.0)
    if N > 1:
        out[:, 1] = (S - 1.0, S - 1.0, S - 1.0 - S / 8.0, S - 1.0)
    hi = S - 1.0
    for i in range(2, N):
        while True:
            cx, cy = rng.u01() * hi, rng.u01() * hi
            length = S / 32.0 + rng.u01() * (S / 4.0 - S / 32.0)
            angle = rng.u01() * math.pi
            x1, y1, x2, y2 = _line(cx, cy, length, angle)
            if 0.0 <= x1 <= hi and 0.0 <= y1 <= hi and 0.0 <= x2 <= hi and 0.0 <= y2 <= hi:
                out[:, i] = (x1, y1, x2, y2)
                break
    return out.astype(np.float32)


def templates(T, n, S, seed):
    """List of T (4, n) float32 arrays in object-local coordinates."""
    rng = SplitMix64(seed)
    res = []
    for _ in range(T):
        t = np.zeros((4, n), dtype=np.float64)
        for i in range(n):
            cx, cy = rng.u01() * (S / 8.0), rng.u01() * (S / 8.0)
            length = S / 64.0 + rng.u01() * (S / 16.0 - S / 64.0)
            angle = rng.u01() * math.pi
            t[:, i] = _line(cx, cy, length, angle)
        res.append(t.astype(np.float32))
    return res


# BASELINE.json configs
CONFIGS = {
    "2": dict(S=1024, scene_lines=200, depth=30, distance=0, T=100, n=32),
    "2p": dict(S=1024, scene_lines=200, depth=30, distance=0, T=1000, n=32),
    "3": dict(S=2048, scene_lines=400, depth=60, distance=1, T=1000, n=32),
    "4": dict(S=2048, scene_lines=400, depth=60, distance=1, T=8000, n=32),
    "5": dict(S=4096, scene_lines=800, depth=180, distance=2, T=16000, n=32),
}


def make_config(name, T=None):
    c = dict(CONFIGS[name])
    if T is not None:
        c["T"] = T
    sc = scene(c["S"], c["scene_lines"], 1)
    tm = templates(c["T"], c["n"], c["S"], 2)
    return c, sc, tm
