"""Pins the CPU oracle (oracle/fdcm_oracle.cpp) to the reference's own known-answer tests.

Each test names the reference test it transcribes (inputs and expected values are data from
/root/reference/tests/**).  CPU only.
"""
import math

import numpy as np
import pytest

from helpers import apply_transform, create_lines, make_rotation, rotate_about
from oracle import oracle as O

f32 = np.float32
PI = math.pi
PI_2F = float(f32(PI / 2))
PI_4F = float(f32(PI / 4))


def L(*cols):
    """Lines from (x1,y1,x2,y2) tuples -> (4, N)."""
    return np.array(cols, dtype=np.float32).T.reshape(4, -1)


# ---------------------------------------------------------------- imgproc.test.cpp
@pytest.mark.parametrize("angle,expected", [
    (-PI_2F, [[8, 8, 8, 8], [8, 7, 6, 5]]),
    (float(f32(-PI / 4)), [[8, 9, 10], [8, 7, 6]]),
    (0.0, [[8, 9, 10, 11], [8, 8, 8, 8]]),
    (float(f32(PI / 4)), [[8, 9, 10], [8, 9, 10]]),
    (PI_2F, [[8, 8, 8, 8], [8, 9, 10, 11]]),
])
def test_rasterize_line_specific_angles(angle, expected):
    # imgproc.test.cpp:35-84
    line = np.array([8, 8, 11, 8], dtype=np.float32)
    rl = rotate_about(line, make_rotation(angle), (8, 8))[:, 0] if angle != 0.0 else line
    assert np.array_equal(O.rasterize_line(rl), np.array(expected))


def test_rasterize_line_shorter_than_half():
    # imgproc.test.cpp:86-95
    r = O.rasterize_line([0, 0, 0.4, 0])
    assert r.shape == (2, 1) and np.array_equal(r[:, 0], [0, 0])


def test_draw_lines_cases():
    # imgproc.test.cpp:97-144
    z = np.zeros((2, 2), dtype=np.float32)
    assert not np.array_equal(O.draw_lines(z, L((-1, -1, 3, 3)), 1.0), z)           # clipped line
    assert np.array_equal(O.draw_lines(z, L((1, -1, -1, 0)), 1.0), z)               # out of bound
    assert np.array_equal(O.draw_lines(z, np.zeros((4, 0)), 1.0), z)                # empty
    assert np.array_equal(O.draw_lines(np.zeros((1, 7)), L((2, 0, 5, 0)), 1.0), [[0, 0, 1, 1, 1, 1, 0]])
    assert np.array_equal(O.draw_lines(np.zeros((7, 1)), L((0, 2, 0, 5)), 1.0)[:, 0], [0, 0, 1, 1, 1, 1, 0])
    exp = np.eye(5, dtype=np.float32)
    exp[0, 0] = exp[4, 4] = 0
    assert np.array_equal(O.draw_lines(np.zeros((5, 5)), L((1, 1, 3, 3)), 1.0), exp)


@pytest.mark.parametrize("angle", [-PI_2F, -PI_4F, 0.0, PI_4F, float(f32(f32(PI_2F) - f32(1e-4)))])
def test_line_integral_orientations(angle):
    # imgproc.test.cpp:146-164
    line = np.array([8, 8, 11, 8], dtype=np.float32)
    lr = rotate_about(line, make_rotation(angle), (8, 8))
    img = O.draw_lines(np.zeros((20, 20)), lr, 1.0)
    out = O.line_integral(img, angle)
    assert out.max() in (3.0, 4.0)


@pytest.mark.parametrize("dist,single,line", [
    (O.L2, [2, 1, 0, 1], [2, 1, 0, 0, 0, 0, 1, 2]),
    (O.L1, [2, 1, 0, 1], [2, 1, 0, 0, 0, 0, 1, 2]),
    (O.L2_SQUARED, [4, 1, 0, 1], [4, 1, 0, 0, 0, 0, 1, 4]),
])
def test_distance_transform(dist, single, line):
    # imgproc.test.cpp:166-214
    dt = O.distance_transform(L((0, 0, 0, 9)), 5, 10, dist)
    assert dt.shape == (10, 5) and dt[:, 0].sum() == 0
    for i in range(5):
        assert np.allclose(dt[:, i], float(i) ** (2 if dist == O.L2_SQUARED else 1), rtol=1e-5)
    assert abs(dt[:, 1].sum() - dt.shape[0]) <= 1e-5
    assert np.allclose(O.distance_transform(L((2, 0, 5, 0)), 8, 2, dist)[0], line, atol=1e-5, rtol=0)
    assert np.allclose(O.distance_transform(L((2, 0, 2, 0)), 4, 1, dist)[0], single, atol=1e-5, rtol=0)


def test_column_pass_inplace_quirk():
    # f = [0,4,100,100] -> [0,1,4,5] (exact lower envelope would give [0,1,4,8]);
    # follows imgproc.h:122-128 reading already overwritten cells.
    out = O.column_pass_l2(np.array([[0], [4], [100], [100]], dtype=np.float32))
    assert np.array_equal(out[:, 0], [0, 1, 4, 5])


# ---------------------------------------------------------------- drawing.test.cpp
CLIP_CASES = [
    ((2, 3, 7, 8), (2, 3, 7, 8)),
    ((-2, 1, 7, 1), (0, 1, 7, 1)),
    ((-2, 1, 12, 1), (0, 1, 10, 1)),
    ((1, -12, 1, 9), (1, 0, 1, 9)),
    ((-2, -2, 12, 12), (0, 0, 10, 10)),
    ((-2, 12, 12, -2), (0, 10, 10, 0)),
    ((12, 12, -2, -2), (10, 10, 0, 0)),
    ((-2, 5, 12, 5), (0, 5, 10, 5)),
    ((12, 5, -2, 5), (10, 5, 0, 5)),
    ((5, 12, 5, -2), (5, 10, 5, 0)),
    ((5, -2, 5, 12), (5, 0, 5, 10)),
    ((-2000001.0, -2000001.0, 12000001.0, 12000001.0), (0, 0, 10, 10)),
]


@pytest.mark.parametrize("line,expected", CLIP_CASES)
def test_clip_lines(line, expected):
    # drawing.test.cpp:31-127 (exact equality)
    out = O.clip_lines(L(line), 0.0, 10.0, 0.0, 10.0)
    assert out.shape == (4, 1) and np.array_equal(out[:, 0], np.array(expected, dtype=np.float32))


def test_clip_lines_outside():
    assert O.clip_lines(L((-2, -3, -7, -8)), 0.0, 10.0, 0.0, 10.0).shape[1] == 0


# ---------------------------------------------------------------- math.test.cpp
def test_rasterize_vector():
    # math.test.cpp:251-299
    tan60 = 1.0 / math.sqrt(3.0)
    eps = float(f32(PI / 12))
    cases = [(-PI_4F - eps, (tan60, -1)), (-PI_4F + eps, (1, -tan60)), (PI / 4 - eps, (1, tan60)),
             (PI_4F + eps, (tan60, 1)), (3 * PI_4F - eps, (-tan60, 1)), (3 * PI_4F + eps, (-1, tan60)),
             (-3 * PI_4F - eps, (-1, -tan60)), (-3 * PI_4F + eps, (-tan60, -1))]
    for ang, exp in cases:
        v = make_rotation(ang) @ np.array([2.0, 0.0], dtype=np.float32)
        assert np.allclose(O.rasterize_vector(v[0], v[1]), exp, atol=1e-5, rtol=0), ang
    assert np.isnan(O.rasterize_vector(0.0, 0.0)).any()


def test_transform_and_align():
    # math.test.cpp:133-248
    la = L((0, 0, 0, 1), (0, 0, 1, 0), (1, 1, 2, 2), (-1, -2, -3, 4))
    T = np.array([[-1, 0, 1], [0, -1, 2]], dtype=np.float32)
    exp = L((1, 2, 1, 1), (1, 2, 0, 2), (0, 1, -1, 0), (2, 4, 4, -2))
    assert np.allclose(O.transform(la, T), exp, atol=1e-5)
    la2 = L((0, -4, 0, 0), (0, 0, 2, 0), (0, 0, 8, 8), (0, 0, 0, 16))
    aline = np.array([-1, -1, 1, 1], dtype=np.float32)
    for T in O.align(la2[:, 0], aline):
        al = O.transform(la2, T)
        c = (al[2:, 0] + al[:2, 0]) / 2
        assert np.allclose(c, [0, 0], atol=1e-5)  # centre of the alignment line
        # all lines turned by the same angle (mod pi)
        d0 = np.arctan((la2[3] - la2[1]) / (la2[2] - la2[0]))
        d1 = np.arctan((al[3] - al[1]) / (al[2] - al[0]))
        diff = np.arctan(1.0) - d0[0]
        want = (d0 + diff + PI / 2) % PI - PI / 2
        assert np.allclose(d1, want, atol=1e-5)


# ---------------------------------------------------------------- dt3cpu.test.cpp
def test_scene_centered_translation():
    # dt3cpu.test.cpp:37-73
    t, size = O.scene_centered_translation(L((0, 0, 9, 0), (0, 0, 0, 9)), 1.0)
    assert tuple(size) == (10, 10) and np.allclose(t, [0, 0])
    t, size = O.scene_centered_translation(L((-6, 1, 4, 1), (0, -10, 0, 10)), 2.0)
    assert tuple(size) == (41, 41) and np.allclose(t, [21, 20])


MINMAX_CASES = [
    (L((4, 0, 5, 0), (5, 0, 6, 0)), (1, 0), (10, 1), (-4, 3)),
    (L((0, 4, 0, 5), (0, 5, 0, 6)), (0, 1), (1, 10), (-4, 3)),
    (L((3, 4, 4, 5), (4, 5, 4, 6)), (0.5, 0.5), (10, 10), (-6, 6)),
    (L((0, 0, 10, 10)), (1, 0), (20, 20), (0, 9)),
    (L((19, 0, 19, 19)), (1, 0), (20, 20), (-19, 0)),
    (L((0, 0, 19, 19)), (1, 0), (20, 20), (0, 0)),
    (L((10, 0, 10, 10)), (-1, 0), (20, 20), (-9, 10)),
    (L((0, 10, 10, 10)), (0, -1), (20, 20), (-9, 10)),
]


@pytest.mark.parametrize("tmpl,av,size,expected", MINMAX_CASES)
def test_minmax_translation(tmpl, av, size, expected):
    # dt3cpu.test.cpp:78-224
    r = O.minmax_translation(tmpl, av, size)
    assert r[0] == expected[0] and r[1] == expected[1]


def test_minmax_translation_degenerate():
    r = O.minmax_translation(np.zeros((4, 0)), (0, 0), (0, 0))
    assert np.isposinf(r).all()
    for tm in [L((3, 4, 4, 5), (4, 5, 10, 6)), L((-1, 4, 4, 5), (4, 5, 9, 6)),
               L((3, 4, 4, 5), (4, 10, 9, 6)), L((1, 4, 4, 5), (4, -1, 9, 6))]:
        assert np.isnan(O.minmax_translation(tm, (1, 1), (10, 10))).all()


def test_closest_orientation_and_classify():
    # dt3cpu.test.cpp:230-266
    keys = np.sort(np.array([-PI_2F + PI / 100, -PI / 4.0, 0.0, PI / 4.0, PI_2F - PI / 100, PI], dtype=np.float32))
    for ang in keys:
        rl = rotate_about(np.array([0, 0, 1, 0], dtype=np.float32), make_rotation(ang), (0, 0))[:, 0]
        k = O.closest_orientation(keys, rl)
        want = (float(ang) + PI / 2) % PI - PI / 2
        assert abs(float(keys[k]) - want) < 1e-6
    la = L((0, 0, 0, 10), (0, 0, 20, 20), (0, 0, 10, 0), (0, 10, 10, 0), (10, 10, 10, 0))
    ks = np.array([-PI_4F, 0.0, PI_4F, PI_2F], dtype=np.float32)
    bins = [O.closest_orientation(ks, la[:, i]) for i in range(5)]
    assert bins == [3, 2, 1, 0, 3]


def test_propagate_orientation():
    # dt3cpu.test.cpp:268-295
    coeff, W, H = 0.5, 30, 40
    keys = np.array([-PI_2F, -PI_4F, 0.0, PI_4F], dtype=np.float32)
    vol = np.full((4, W, H), np.inf, dtype=np.float32)
    vol[0] = O.distance_transform(L((0, 0, 0, 39)), W, H, O.L2).T
    out = O.propagate(keys, vol, coeff)
    d1 = out[0][29, 0]
    assert d1 == 29.0
    for k in range(4):
        da = abs((float(keys[0]) - float(keys[k]) + PI / 2) % PI - PI / 2)
        assert abs((d1 + da * coeff) - out[k][29, 0]) <= 1e-5


def test_build_featuremap_precision():
    # dt3cpu.test.cpp:318-345: end-to-end known answers of the whole build
    for scale, exp in [(1.0, [2, 3, 3, 3, 3, 3, 4]), (2.0, [3, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7, 9, 12])]:
        scene = L((2, 0, 5, 0)) * f32(scale)
        fm = O.build(scene, depth=4, coeff=1.0, padding=2.0, distance=O.L2)
        k = O.closest_orientation(fm.keys, scene[:, 0])
        feat = fm.slice(k)
        assert np.allclose(feat[feat.shape[0] // 2], exp, atol=1e-5, rtol=0)


def test_build_featuremap_lines_are_flat():
    # dt3cpu.test.cpp:297-317
    scene = L((0, 0, 0, 1), (0, 0, 1, 1), (0, 0, 1, 0), (0, 1, 1, 0), (1, 1, 1, 0))
    fm = O.build(scene, depth=4, coeff=50.0, padding=1.0, distance=O.L2)
    for i in range(5):
        l = scene[:, i]
        feat = fm.slice(O.closest_orientation(fm.keys, l))
        p1 = np.round(l[:2]).astype(int)
        p2 = np.round(l[2:]).astype(int)
        assert abs(feat[p2[1], p2[0]] - feat[p1[1], p1[0]]) <= 1.0


# ---------------------------------------------------------------- batchoptimize / defaultoptimize / indulgentoptimize tests
@pytest.mark.parametrize("kind", [O.BATCH_OPTIMIZE, O.DEFAULT_OPTIMIZE, O.INDULGENT_OPTIMIZE])
def test_optimize_known_answers(kind):
    # batchoptimize.test.cpp:39-116, defaultoptimize.test.cpp, indulgentoptimize.test.cpp:34-118 (same three cases)
    tmpl = L((10, 0, 10, 10), (0, 0, 10, 0))
    scene = L((15, 0, 15, 10), (5, 0, 15, 0))
    fm = O.build(scene, depth=4, coeff=1.0, padding=1.0)
    r = O.optimize(fm, O.transform(tmpl, [[1, 0, 5], [0, 1, 0]]), (1, 0), kind, 10)
    assert r is not None and np.allclose(r[1], [0, 0], atol=1e-5) and r[0] == 0
    scene = L((3, 0, 6, 0), (0, 10, 7, 10))
    fm = O.build(scene, depth=4, coeff=1.0, padding=1.0)
    r = O.optimize(fm, L((0, 0, 5, 0)), (1, 0), kind, 10)
    assert r is not None and np.allclose(r[1], [2, 0], atol=1e-5) and abs(r[0] - 1.0) < 1e-6
    fm = O.build(L((0, 0, 1, 0)), depth=4, coeff=1.0, padding=1.0)
    assert O.optimize(fm, L((0, 0, 10, 10)), (1, 0), kind, 10) is None


# ---------------------------------------------------------------- searchstrategy.test.cpp
def test_default_search_and_centered_range():
    # searchstrategy.test.cpp:42-90
    scene = L((0, 0, 1, 0), (0, 0, 2, 0), (0, 0, 3, 0), (0, 0, 6, 0), (0, 0, 5, 0))
    tmpl = L((0, 0, 2, 0), (0, 0, 3, 0), (0, 0, 1, 0), (0, 0, 8, 0))
    combos = {tuple(c) for c in O.default_search(tmpl, scene, 2, 2)}
    assert combos == {(3, 3), (3, 4), (1, 2), (1, 4)}
    assert O.centered_range(30, 60, 60) == (0, 60)
    assert O.centered_range(3, 6, 10) == (0, 6)
    assert O.centered_range(0, 6, 2) == (0, 2)
    assert O.centered_range(5, 6, 2) == (4, 6)


# ---------------------------------------------------------------- matchstrategy.test.cpp / test_matching.py
@pytest.mark.parametrize("n_lines,length,maxT,maxS,dist,kind", [
    (10, 10, 3, 3, O.L2, O.DEFAULT_OPTIMIZE),          # matchstrategy.test.cpp:36-111
    (10, 100, 4, 10, O.L2, O.DEFAULT_OPTIMIZE),        # test_matching.py:45-104
    (10, 100, 4, 10, O.L1, O.DEFAULT_OPTIMIZE),
    (10, 100, 4, 10, O.L2_SQUARED, O.DEFAULT_OPTIMIZE),
    (10, 100, 4, 10, O.L2, O.BATCH_OPTIMIZE),
])
def test_end_to_end_matching(n_lines, length, maxT, maxS, dist, kind):
    tmpl = create_lines(n_lines, length)
    # test_matching.py:62-104 reassigns scene_transform inside its distance loop, so the reference
    # exercises the 180-degree case with L2 only; L1 / L2_SQUARED see the identity case.
    cases = [[[1, 0, 0], [0, 1, 0]]]
    if dist == O.L2:
        cases.insert(0, [[-1, 0, length], [0, -1, length]])
    for T in cases:
        T = np.array(T, dtype=np.float32)
        scene = apply_transform(tmpl, T)
        fm = O.build(scene, depth=30, coeff=5.0, padding=2.2, distance=dist, nthreads=2)
        m = O.search(fm, [tmpl], scene, maxT, maxS, kind=kind, batch=10, nthreads=2)
        assert len(m) == min(maxT, n_lines) * min(n_lines, maxS) * 2
        best = m[np.argmin(m["score"])]
        tr = best["transform"].reshape(2, 3)
        assert best["tmpl_idx"] == 0
        assert np.allclose(tr[:, :2], T[:, :2], atol=1e-5)
        assert np.allclose(tr[:, 2], T[:, 2], atol=1.0)
    fm = O.build(tmpl, depth=30, coeff=5.0, padding=2.2, distance=dist)
    assert len(O.search(fm, [], tmpl, maxT, maxS, kind=kind)) == 0
    assert len(O.search(fm, [np.zeros((4, 0))], tmpl, maxT, maxS, kind=kind)) == 0
    fm0 = O.build(np.zeros((4, 0)), depth=30, coeff=5.0, padding=2.2, distance=dist)
    assert (fm0.W, fm0.H, fm0.depth) == (0, 0, 0)
    assert len(O.search(fm0, [tmpl], np.zeros((4, 0)), maxT, maxS, kind=kind)) == 0


# ---------------------------------------------------------------- searchstrategy.test.cpp (ConcentricRange)
def test_concentric_range_known_answers():
    # searchstrategy.test.cpp:91-194
    tm = L((0, 0, 5, 5), (2, 2, 4, 4), (0, 0, 5, 0), (0, 0, 0, 5), (0, 0, 2, 2), (3, 3, 4, 4), (4, 0, 5, 5))
    assert list(O.filter_in_range(tm, (2.5, 2.5), 0.0, 2.0)) == [0, 1, 5]
    tmpl = L((0, 0, 2, 0), (0, 0, 3, 0), (0, 0, 1, 0), (0, 0, 8, 0))
    assert len(O.concentric_search(tmpl, np.zeros((4, 0)), 2, 2, (0, 0), 5, 15)) == 0
    scene = L((0, 0, 1, 0), (0, 0, 13, 0), (0, 0, 30, 0), (0, 0, 20, 0), (0, 0, 5, 0))
    combos = {tuple(c) for c in O.concentric_search(tmpl, scene, 2, 2, (0, 0), 5, 15)}
    assert combos and combos <= {(3, 1), (3, 3), (1, 1), (1, 3)}
    scene = L((0, 0, 2, 0), (2, 0, 4, 0), (4, 0, 7, 0), (7, 0, 15, 0))
    one = L((0, 0, 2, 0))
    inf = float("inf")
    for lo, hi, want in [(0, 2, (0, 1)), (3, 15, (0, 3)), (3, inf, (0, 3)), (2, 4, (0, 0))]:
        assert tuple(O.concentric_search(one, scene, 1, 1, (4, 0), lo, hi)[0]) == want


# ---------------------------------------------------------------- math.test.cpp (vectors, lines)
def test_argsort_binary_search_minmax_point():
    # math.test.cpp:33-63
    assert O.argsort_greater([-4, 3, -1, 2]) == [1, 3, 2, 0]
    # binarySearch returns the index of the closest element; the reference's case is ascending with the
    # default comparator, the path uses the descending variant (defaultsearch.cpp:41): negate both
    vec = [0, 2, 3, 6, 7, 10, 14, 30, 40, 123]
    desc = [-x for x in vec]
    for value, idx in [(0, 0), (123, 9), (2, 1), (40, 8), (5, 3), (4, 2)]:
        assert O.binary_search_greater(desc, -value) == idx, (value, idx)
    mn, mx = O.minmax_point(L((0, -4, 0, 0), (0, 0, 2, 0), (0, 0, 8, 8), (0, 0, 0, 16)))
    assert np.allclose(mn, [0, -4], atol=1e-5) and np.allclose(mx, [8, 16], atol=1e-5)


def test_line_angle_length_normalize_translate():
    # math.test.cpp:82-129,195-211 with the fixture at :84-89
    lines = [(0, 0, 2, 2), (0, 0, 1, 0), (-1, 1, 0, 0)]
    exp_angle = [np.pi / 4, 0.0, -np.pi / 4]
    exp_len = [np.sqrt(8.0), 1.0, np.sqrt(2.0)]
    exp_dir = [(np.sqrt(2) / 2, np.sqrt(2) / 2), (1.0, 0.0), (np.sqrt(2) / 2, -np.sqrt(2) / 2)]
    for l, a, ln, d in zip(lines, exp_angle, exp_len, exp_dir):
        angle, length, direction = O.line_props(l)
        assert abs(angle - a) < 1e-5 and abs(length - ln) < 1e-5 and np.allclose(direction, d, atol=1e-5)
    # getTemplateLengths == getLength(tmpl).sum()
    assert abs(O.eigen_sum(np.array(exp_len, dtype=np.float32)) - float(np.float32(sum(np.float32(x) for x in exp_len)))) < 1e-5
    moved = O.translate(L((0, 0, 1, 1), (1, 1, 2, 2)), (1, 2))
    assert np.allclose(moved, L((1, 2, 2, 3), (2, 3, 3, 4)), atol=1e-6)
    # combine(translation, transform), the overload defaultmatch.cpp:83 uses (math.h:427-432)
    assert np.allclose(O.combine((3, 4), [[-1, 0, 1], [0, -1, 2]]), [[-1, 0, 4], [0, -1, 6]], atol=1e-6)
