"""The statement the balanced L2 sweep (openfdcm_amd/csrc/fdcm_sweep.hip) rests on, pinned on the CPU against the oracle's
literal restatement of the reference's pass (imgproc.h:91-130, oracle/fdcm_oracle.cpp: columnPassL2):

    for W^2 + H^2 <= 2^24, run on the output of the first pass (squares of integer distances, FLT_MAX for a column without
    a seed), the second pass gives   out[q] = base + (q - o)^2,   o = the EXACT owner of pixel q -- the seeded column that
    minimises f[u] + (q - u)^2 over the integers, the smallest such u on a tie --, base = f[o] for q <= o and out[o] for q > o.

DESIGN.md section 4 has the argument (the only rounded quantity is the quotient, and an integer pixel compares with a
rounded quotient as with the exact one); tools/sim/exact_owner_sim.cpp runs the same comparison over every row of the
BASELINE scenes.  Here: random columns, staircases (rows along and far from a scene line), near-collinear points with long
gaps (where two quotients round to the same float), few seeds, values at the 2^24 bound."""
import numpy as np
import pytest

from oracle import oracle as O

FMAX = np.float32(np.finfo(np.float32).max)


def exact_pass(f):
    """f: float32 vector (squares of integers or FLT_MAX) -> the statement above, in integers."""
    n = len(f)
    cols = np.flatnonzero(f != FMAX)
    if len(cols) == 0:
        return f.copy()
    fi = f[cols].astype(np.int64)
    q = np.arange(n, dtype=np.int64)
    cost = fi[None, :] + (q[:, None] - cols[None, :]) ** 2       # [pixel][seeded column]
    owner = cols[np.argmin(cost, axis=1)]                          # argmin takes the first (smallest) column on a tie
    out = np.zeros(n, dtype=np.int64)
    fint = np.zeros(n, dtype=np.int64)
    fint[cols] = fi
    for p in range(n):
        o = owner[p]
        out[p] = (out[o] if o < p else fint[o]) + (p - o) ** 2
    assert out.max() < 2 ** 24 + 2 ** 23
    return out.astype(np.float32)


def literal_pass(f):
    # oracle.column_pass_l2 takes an (H, W) image and runs the pass down every column: one column of length n
    return O.column_pass_l2(f.reshape(-1, 1)).reshape(-1)


def _cases(rng, n):
    H = int(rng.integers(1, 2897))
    hmax = int(np.floor(np.sqrt(2 ** 24 - n * n))) if n * n < 2 ** 24 else 0
    kinds = []
    # random squares
    d = rng.integers(0, min(H, hmax) + 1, size=n)
    kinds.append(("random", d, rng.uniform(size=n) < rng.uniform(0.05, 1.0)))
    # a scene line crossing: |a (u - c) + b|
    a, c, b = rng.uniform(-2, 2), rng.integers(0, n), rng.integers(0, 200)
    d = np.abs(np.rint(a * (np.arange(n) - c) + b)).astype(np.int64) % (min(H, hmax) + 1)
    kinds.append(("staircase", d, rng.uniform(size=n) < rng.uniform(0.3, 1.0)))
    # near-collinear P = f + u^2 with long gaps between seeds: quotients that round together
    u = np.arange(n, dtype=np.float64)
    P = 2.0 * n * n + 2.0 * n * (u - c) * 0.999 + rng.uniform(0, 0.01) * (u - c) ** 2
    fv = np.clip(P - u * u, 0, None)
    d = np.minimum(np.floor(np.sqrt(fv)).astype(np.int64) + rng.integers(0, 2, size=n), hmax)
    kinds.append(("near-collinear", d, rng.uniform(size=n) < 0.03))
    # few seeds, far apart; values at the bound
    kinds.append(("few", np.full(n, hmax, dtype=np.int64) - rng.integers(0, 3, size=n), rng.uniform(size=n) < 0.01))
    return kinds


@pytest.mark.parametrize("n", [7, 64, 333, 1024, 2048, 2896])
def test_literal_pass_equals_exact_owners(n):
    rng = np.random.default_rng(n)
    reps = 40 if n <= 333 else (12 if n <= 1024 else 4)
    checked = 0
    for _ in range(reps):
        for name, d, seeded in _cases(rng, n):
            f = np.where(seeded, (d.astype(np.int64) ** 2).astype(np.float32), FMAX).astype(np.float32)
            got, want = literal_pass(f), exact_pass(f)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (n, name, int(np.flatnonzero(got != want)[0]))
            checked += 1
    assert checked == 4 * reps


def test_all_seedless_and_single_seed():
    n = 50
    f = np.full(n, FMAX, dtype=np.float32)
    assert np.array_equal(literal_pass(f), exact_pass(f))          # FLT_MAX + d^2 == FLT_MAX everywhere
    for pos in (0, 17, n - 1):
        g = f.copy()
        g[pos] = 9.0
        assert np.array_equal(literal_pass(g), exact_pass(g))
