"""The reference's penalty known answers (tests/matching/src/penaltystrategy.test.cpp) against the
product's host tail: fdcm_penalize / fdcm_sort_matches through the Python API mirror.  Host-only
entry points of libfdcm_hip.so: no GPU needed."""
import numpy as np
import pytest

from helpers import create_lines
from oracle import oracle as O


@pytest.fixture(scope="module")
def api():
    import __graft_entry__ as g
    import os
    from openfdcm_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        g.build()
    import openfdcm_amd
    return openfdcm_amd


def lengths_of(templates):
    """getTemplateLengths (math.h:319-324) with Eigen's sum order, from the oracle (the product's version needs
    a device handle)."""
    out = []
    for t in templates:
        t = np.asarray(t, dtype=np.float32)
        out.append(O.eigen_sum(np.array([O.line_props(t[:, i])[1] for i in range(t.shape[1])], dtype=np.float32))
                   if t.shape[1] else 0.0)
    return np.array(out, dtype=np.float32)


def two_matches(api):
    return [api.Match(0, 1.0, np.array([[1, 2, 3], [4, 5, 6]], dtype=np.float32)),
            api.Match(1, 1.0, np.array([[4, 5, 6], [1, 2, 3]], dtype=np.float32))]


@pytest.mark.parametrize("which", ["default", "exponential"])
def test_penalize_null_length_and_inconsistent_lengths(api, which):
    # penaltystrategy.test.cpp:35-73
    penalty = api.DefaultPenalty() if which == "default" else api.ExponentialPenalty(2.0)
    lens = lengths_of([np.zeros((4, 1), dtype=np.float32)])
    out = api.penalize(penalty, [api.Match(0, 1.0, np.array([[1, 2, 3], [4, 5, 6]], dtype=np.float32))], lens)
    assert not np.isnan(out[0].score)
    with pytest.raises(IndexError):  # std::out_of_range
        api.penalize(penalty, two_matches(api), [])


def test_validate_default_and_exponential_penalty(api):
    # penaltystrategy.test.cpp:75-141: rotating / translating a template does not change its length
    t1 = -np.asarray(create_lines(4, 4), dtype=np.float32)
    t2 = np.asarray(create_lines(3, 3), dtype=np.float32) + np.array([[1], [2], [1], [2]], dtype=np.float32)
    lens = lengths_of([t1, t2])
    for penalty, denom in ((api.DefaultPenalty(), lambda l: l), (api.ExponentialPenalty(1.45), lambda l: np.power(np.float32(l), np.float32(1.45)))):
        original = two_matches(api)
        out = api.penalize(penalty, original, lens)
        assert len(out) == len(original)
        for i, (a, b) in enumerate(zip(out, original)):
            assert a.tmpl_idx == b.tmpl_idx and np.allclose(a.transform, b.transform, atol=1e-5)
            want = np.float32(b.score) / np.float32(denom(lens[i]))
            assert abs(a.score - want) <= np.finfo(np.float32).eps + 1e-10 * max(abs(a.score), abs(want))


def test_sort_matches_ascending(api):
    # python/src/matching.cpp:302-307
    ms = [api.Match(i, s, np.eye(2, 3, dtype=np.float32)) for i, s in enumerate([3.0, 0.5, 2.0, 0.25])]
    assert [m.tmpl_idx for m in api.sort_matches(ms)] == [3, 1, 2, 0]
