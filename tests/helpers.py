"""Shared test helpers: restatements of the reference's test utilities and the synthetic generators.

create_lines / make_rotation follow tests/test-utils/include/test-utils/utils.h:38-94 and
tests/python/test_matching.py:5-41 of the reference (inputs of its end-to-end tests).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

f32 = np.float32


def make_rotation(angle):
    s, c = f32(np.sin(f32(angle))), f32(np.cos(f32(angle)))
    return np.array([[c, -s], [s, c]], dtype=np.float32)


def rotate_about(line, rot, pt):
    """core::rotate(line, rotation, rot_point), math.h:372-378."""
    line = np.asarray(line, dtype=np.float32).reshape(4, -1)
    pt = np.asarray(pt, dtype=np.float32)
    t = pt - rot @ pt
    pts = line.reshape(2, -1, order="F")
    out = (rot @ pts + t[:, None]).astype(np.float32)
    return out.reshape(4, -1, order="F")


def create_lines(n, length):
    """tests::createLines (utils.h:74-91): a fan of n lines from the origin, log-spaced angles in [2pi, 4pi]."""
    log_start = f32(np.log10(f32(2 * np.pi)))
    log_end = f32(np.log10(f32(4 * np.pi)))
    step = f32((log_end - log_start) / f32(n - 1))
    out = np.zeros((4, n), dtype=np.float32)
    for i in range(n):
        ang = f32(np.power(10.0, float(f32(log_start + f32(i) * step))))
        r = make_rotation(ang)
        out[2, i] = r[0, 0] * f32(length)
        out[3, i] = r[1, 0] * f32(length)
    return out


def apply_transform(lines, T):
    lines = np.asarray(lines, dtype=np.float32)
    T = np.asarray(T, dtype=np.float32)
    pts = lines.reshape(2, -1, order="F")
    return (T[:, :2] @ pts + T[:, 2:3]).astype(np.float32).reshape(4, -1, order="F")
