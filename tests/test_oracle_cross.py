"""The two independent CPU restatements (C++ oracle, numpy pyoracle) must agree bit for bit.

This is what stands in for running the reference binary (which cannot be built here): the
restatements were written separately from the cited reference lines.  Sizes are small because the
numpy one is slow.
"""
import numpy as np
import pytest

from helpers import create_lines
from openfdcm_amd import synthetic
from oracle import oracle as O
from oracle import pyoracle as P


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("dist", [O.L2, O.L2_SQUARED, O.L1])
@pytest.mark.parametrize("S,n,depth,seed,pad", [(40, 9, 6, 1, 1.0), (57, 14, 9, 2, 1.3), (64, 20, 30, 3, 1.0)])
def test_build_bit_exact(dist, S, n, depth, seed, pad):
    scene = synthetic.scene(S, n, seed)
    for stage in (1, 2, 3):
        a = O.build(scene, depth=depth, coeff=5.0, padding=pad, distance=dist, stop_after=stage)
        b = P.build(scene, depth=depth, coeff=5.0, padding=pad, dist=dist, stop_after=stage)
        assert (a.W, a.H) == (b["W"], b["H"]) and np.array_equal(_bits(a.translation), _bits(b["t"]))
        assert np.array_equal(_bits(a.keys), _bits(b["keys"]))
        va, vb = a.volume(), b["vol"]
        assert va.shape == vb.shape
        assert np.array_equal(_bits(va), _bits(vb)), f"stage {stage}: {np.sum(_bits(va) != _bits(vb))} voxels differ"


def test_inplace_pass_on_random_columns():
    rng = np.random.default_rng(0)
    for _ in range(20):
        n = int(rng.integers(2, 60))
        f = rng.integers(0, 400, size=n).astype(np.float32)
        f[rng.random(n) < 0.3] = np.finfo(np.float32).max
        a = O.column_pass_l2(f.reshape(n, 1))[:, 0]
        b = f.reshape(n, 1).copy()
        P.column_pass_l2(b)
        assert np.array_equal(_bits(a), _bits(b[:, 0]))


@pytest.mark.parametrize("kind,B", [(O.BATCH_OPTIMIZE, 10), (O.BATCH_OPTIMIZE, 3), (O.DEFAULT_OPTIMIZE, 1)])
def test_search_bit_exact(kind, B):
    S = 72
    scene = synthetic.scene(S, 18, 7)
    tmpls = synthetic.templates(5, 7, S, 8) + synthetic.templates(2, 12, S, 9)
    a = O.build(scene, depth=10, coeff=5.0, padding=1.0)
    b = P.build(scene, depth=10, coeff=5.0, padding=1.0)
    ma = O.search(a, tmpls, scene, 3, 3, kind=kind, batch=B)
    mb = P.search(b, tmpls, scene, 3, 3, kind=kind, B=B)
    assert len(ma) == len(mb) > 0
    for x, y in zip(ma, mb):
        assert x["tmpl_idx"] == y[0]
        assert _bits(x["score"]) == _bits(y[1])
        assert np.array_equal(_bits(x["transform"]), _bits(y[2].reshape(6)))


def test_eigen_sum_orders():
    rng = np.random.default_rng(1)
    for n in list(range(0, 20)) + [31, 32, 33, 47]:
        v = (rng.random(n) * 1000).astype(np.float32)
        assert _bits(O.eigen_sum(v)) == _bits(P.eigen_sum(v)), n
