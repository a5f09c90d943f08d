#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the numpy restatement (oracle/pyoracle.py).

The reference binary cannot be built in this image (Eigen etc. are network dependencies), so the
golden vectors are produced by the slow, independent numpy restatement; the C++ oracle and the HIP
path are both checked against them (tests/test_golden.py).  obj_04/* are data files shipped by the
reference (notebooks/assets/obj_04: scene_0/camera_0.scene and templates/template_{0..11}.tmpl).

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from openfdcm_amd import lineio, synthetic  # noqa: E402
from oracle import pyoracle as P  # noqa: E402


def matches_to_arrays(m):
    idx = np.array([x[0] for x in m], dtype=np.int32)
    score = np.array([x[1] for x in m], dtype=np.float32)
    tr = np.array([x[2].reshape(6) for x in m], dtype=np.float32).reshape(-1, 6)
    return idx, score, tr


def case(name, scene, templates, depth, coeff, padding, dist, maxT, maxS, kind, B):
    fm = P.build(scene, depth=depth, coeff=coeff, padding=padding, dist=dist)
    m = P.search(fm, templates, scene, maxT, maxS, kind=kind, B=B)
    idx, score, tr = matches_to_arrays(m)
    flat = np.concatenate([t.T for t in templates], axis=0).astype(np.float32)
    offs = np.cumsum([0] + [t.shape[1] for t in templates]).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), scene=scene, tmpl_lines=flat, tmpl_offsets=offs,
                        params=np.array([depth, coeff, padding, dist, maxT, maxS, kind, B], dtype=np.float64),
                        keys=fm["keys"], volume=fm["vol"], translation=fm["t"], size=np.array([fm["W"], fm["H"]]),
                        m_idx=idx, m_score=score, m_transform=tr)
    print(name, "volume", fm["vol"].shape, "matches", len(m))


def main():
    S = 48
    sc = synthetic.scene(S, 14, 21)
    tm = synthetic.templates(4, 6, S, 22)
    case("synth48_l2_batch", sc, tm, 8, 5.0, 1.0, P.L2, 3, 3, 1, 10)
    case("synth48_l2sq_default", sc, tm, 8, 5.0, 1.0, P.L2_SQUARED, 3, 3, 0, 1)
    case("synth48_l1_batch3", sc, tm, 8, 5.0, 1.25, P.L1, 2, 4, 1, 3)
    # config 1 plumbing case on the reference's own data, down-scaled so the numpy build stays short:
    # scene lines x 0.15 (a 74 x 74 feature map), templates 0..3 scaled alike, depth 30, coeff 5, L2.
    scene = (lineio.read(os.path.join(HERE, "obj_04", "scene_0.scene")) * np.float32(0.15)).astype(np.float32)
    tmpls = [(lineio.read(os.path.join(HERE, "obj_04", f"template_{i}.tmpl")) * np.float32(0.15)).astype(np.float32)
             for i in range(4)]
    case("obj04_scaled_l2_batch", scene, tmpls, 30, 5.0, 1.0, P.L2, 4, 4, 1, 10)


if __name__ == "__main__":
    main()
