"""GPU tests of the feature-map plug-in seam (featuremap.h:27-52): fdcm_featuremap_minmax_translation and
fdcm_featuremap_evaluate through the C ABI against the oracle, bit for bit; and the type-erasure shells of the
mirrored Python API (FeatureMap, MatchStrategy, SearchStrategy, OptimizeStrategy, PenaltyStrategy)."""
import numpy as np
import pytest

from helpers import create_lines
from oracle import oracle as O
from test_oracle_kat import MINMAX_CASES, L

pytestmark = pytest.mark.gpu


def _blank_map(size):
    """A feature map of the given (W, H) with one all-zero slice and no scene translation (what the reference's
    minmaxTranslation tests construct: Dt3Cpu({}, {0, 0}, size), dt3cpu.test.cpp:78-224)."""
    from openfdcm_amd.engine import DeviceFeatureMap
    W, H = size
    return DeviceFeatureMap.from_volume(np.zeros(1, dtype=np.float32), np.zeros((1, W, H), dtype=np.float32), (0.0, 0.0))


def _same_bits(a, b):
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    return a.shape == b.shape and bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize("tmpl,av,size,expected", MINMAX_CASES)
def test_minmax_translation_reference_kats(tmpl, av, size, expected):
    """dt3cpu.test.cpp:78-224, the reference's own table, on the device function through the ABI."""
    fm = _blank_map(size)
    r = fm.minmax_translation(tmpl, av)
    assert r[0] == expected[0] and r[1] == expected[1], (r, expected)
    assert _same_bits(r, O.minmax_translation(tmpl, av, size))


def test_minmax_translation_degenerate_branches():
    """zero align vector -> {inf, inf}; a bounding box that starts outside the image -> {NaN, NaN} (dt3cpu.cpp:34-45);
    an axis-parallel vector leaves one row of infinities (the col(0)/col(1) branches, dt3cpu.cpp:69-74)."""
    fm = _blank_map((10, 10))
    assert np.isposinf(fm.minmax_translation(L((3, 4, 4, 5)), (0, 0))).all()
    assert np.isposinf(fm.minmax_translation(np.zeros((4, 0)), (0, 0))).all()
    for tm in [L((3, 4, 4, 5), (4, 5, 10, 6)), L((-1, 4, 4, 5), (4, 5, 9, 6)),
               L((3, 4, 4, 5), (4, 10, 9, 6)), L((1, 4, 4, 5), (4, -1, 9, 6))]:
        assert np.isnan(fm.minmax_translation(tm, (1, 1))).all()
    for av in [(1, 0), (0, 1), (-1, 0), (0, -1), (1e-3, 1), (-0.0, 1), (1, -0.0), (0.3, -0.7)]:
        tm = L((3, 4, 4, 5), (4, 5, 6, 2))
        assert _same_bits(fm.minmax_translation(tm, av), O.minmax_translation(tm, av, (10, 10))), av


def test_minmax_translation_batch_random_against_oracle():
    """One launch for 300 random templates on a built feature map (non-zero scene translation, padding 1.4)."""
    from openfdcm_amd import synthetic
    from openfdcm_amd.engine import DeviceFeatureMap
    scene = synthetic.scene(200, 30, 3)
    fm = DeviceFeatureMap.build(scene, depth=6, coeff=5.0, padding=1.4, distance=0)
    rng = np.random.default_rng(5)
    tmpls, avs = [], []
    for i in range(300):
        n = int(rng.integers(1, 70))
        c = rng.uniform(-20, 220, size=2)
        t = (c[:, None].repeat(2 * n, 1) + rng.uniform(-40, 40, size=(2, 2 * n))).astype(np.float32).reshape(4, n, order="F")
        tmpls.append(t)
        a = rng.uniform(0, 2 * np.pi)
        avs.append(O.rasterize_vector(np.cos(a), np.sin(a)) if i % 3 else (np.float32(np.cos(a)), np.float32(np.sin(a))))
    got = fm.minmax_translation_batch(tmpls, avs)
    size = (fm.width, fm.height)
    inside = 0
    for i, (t, a) in enumerate(zip(tmpls, avs)):
        want = O.minmax_translation(t, a, size, fm.scene_translation)
        assert _same_bits(got[i], want), (i, got[i], want)
        inside += int(np.isfinite(want).all())
    assert 20 < inside < 280  # both the NaN exits and the finite branch are exercised


@pytest.fixture(scope="module")
def built_pair():
    from openfdcm_amd import synthetic
    from openfdcm_amd.engine import DeviceFeatureMap
    scene = synthetic.scene(256, 48, 9)
    dev = DeviceFeatureMap.build(scene, depth=12, coeff=5.0, padding=1.2, distance=0)
    orc = O.build(scene, depth=12, coeff=5.0, padding=1.2, distance=O.L2, nthreads=8)
    return scene, dev, orc


def _random_templates(rng, S, count, max_lines=40):
    out = []
    for _ in range(count):
        n = int(rng.integers(0, max_lines + 1))  # ragged, some empty: score_per_line.sum() of nothing is 0
        c = rng.uniform(0.3 * S, 0.7 * S, size=2)
        pts = c[:, None] + rng.uniform(-0.2 * S, 0.2 * S, size=(2, 2 * n))
        out.append(pts.astype(np.float32).reshape(4, n, order="F"))
    return out


def test_evaluate_random_batches_bit_exact(built_pair):
    """evaluate<Dt3Cpu> (dt3cpu.cpp:126-179): 120 ragged templates x 0..90 translations each in ONE call, every score
    equal to the oracle's bits (Eigen's sum() order, cast<int>() truncation, bins from the host libm)."""
    scene, dev, orc = built_pair
    rng = np.random.default_rng(17)
    S = dev.width / 1.2
    tmpls = _random_templates(rng, S, 120)
    trans = [rng.uniform(-0.12 * S, 0.12 * S, size=(int(rng.integers(0, 91)), 2)).astype(np.float32) for _ in tmpls]
    trans[3] = np.zeros((1, 2), dtype=np.float32)
    got = dev.evaluate(tmpls, trans)
    total = 0
    for i, (t, tr) in enumerate(zip(tmpls, trans)):
        want = O.evaluate(orc, t, tr) if len(tr) else np.zeros(0, dtype=np.float32)
        assert _same_bits(got[i], want), (i, t.shape, got[i][:4], want[:4])
        total += len(tr)
    assert total > 3000
    assert len(dev.evaluate([], [])) == 0


def test_evaluate_and_minmax_from_concurrent_threads(built_pair):
    """The reference's optimisers call minmaxTranslation / evaluate on ONE feature map from the tasks of a thread pool
    (batchoptimize.cpp:102-110, `submit_task(func(tmpl_idx))`).  Eight host threads hammer one handle with different
    templates and translation sets (different sizes, so the shared scratch is re-laid-out between calls); every score
    and every interval must be the oracle's bits."""
    import threading
    scene, dev, orc = built_pair
    S = dev.width / 1.2
    size = (dev.width, dev.height)
    errors, done = [], [0] * 8

    def worker(k):
        try:
            rng = np.random.default_rng(100 + k)
            for it in range(6):
                tmpls = _random_templates(rng, S, 5 + 7 * ((k + it) % 4), max_lines=10 + 6 * k)
                trans = [rng.uniform(-0.1 * S, 0.1 * S, size=(int(rng.integers(1, 40 + 30 * (it % 3))), 2)).astype(np.float32) for _ in tmpls]
                got = dev.evaluate(tmpls, trans)
                for i, (t, tr) in enumerate(zip(tmpls, trans)):
                    want = O.evaluate(orc, t, tr)
                    if not _same_bits(got[i], want):
                        errors.append(("evaluate", k, it, i))
                full = [t for t in tmpls if t.shape[1] > 0]
                avs = [O.rasterize_vector(np.cos(a), np.sin(a)) for a in rng.uniform(0, 2 * np.pi, size=len(full))]
                mm = dev.minmax_translation_batch(full, avs)
                for i, (t, a) in enumerate(zip(full, avs)):
                    if not _same_bits(mm[i], O.minmax_translation(t, a, size, dev.scene_translation)):
                        errors.append(("minmax", k, it, i))
                done[k] += 1
        except Exception as e:  # noqa: BLE001 -- reported below, on the main thread
            errors.append(("exception", k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]
    assert done == [6] * 8


def test_evaluate_outside_the_image_scores_nan(built_pair):
    scene, dev, orc = built_pair
    t = L((10, 10, 40, 30), (20, 15, 60, 18))
    far = np.array([[0, 0], [1e4, 0], [0, -1e4], [np.nan, 0], [3, 2]], dtype=np.float32)
    got = dev.evaluate([t], [far])[0]
    want = O.evaluate(orc, t, far[[0, 4]])
    assert _same_bits(got[[0, 4]], want) and np.isnan(got[1:4]).all()


def test_a_host_optimiser_runs_on_the_seam(built_pair):
    """What the seam is for: an optimiser that lives OUTSIDE the library -- here optimize<DefaultOptimize>
    (defaultoptimize.cpp:6-93) restated in a few lines on top of minmax_translation + evaluate -- reaches the same
    optimum as the oracle's, candidate by candidate."""
    scene, dev, orc = built_pair
    rng = np.random.default_rng(23)
    S = dev.width / 1.2
    n_checked = 0
    for t in _random_templates(rng, S, 25, max_lines=24):
        if t.shape[1] == 0:
            continue
        a = rng.uniform(0, 2 * np.pi)
        av = np.array([np.cos(a), np.sin(a)], dtype=np.float32)
        want = O.optimize(orc, t, av, kind=O.DEFAULT_OPTIMIZE, batch=1)
        sav = np.asarray(O.rasterize_vector(av[0], av[1]), dtype=np.float32)
        lo, hi = dev.minmax_translation(t, sav)
        if not (np.isfinite(lo) and np.isfinite(hi)):
            assert want is None
            continue
        f32 = np.float32
        best_s = dev.evaluate([t], [np.zeros((1, 2), dtype=np.float32)])[0][0]
        best_t = np.zeros(2, dtype=np.float32)
        back = best_s
        for sign, lim in ((1, int(hi)), (-1, int(lo))):
            k = sign
            while (k <= lim) if sign > 0 else (k >= lim):
                tr = (f32(k) * sav).astype(np.float32)
                s = dev.evaluate([t], [tr.reshape(1, 2)])[0][0]
                if s > back:
                    break
                back = s
                if s < best_s:
                    best_s, best_t = s, tr
                k += sign
        assert want is not None
        assert np.float32(want[0]).view(np.uint32) == np.float32(best_s).view(np.uint32)
        assert np.array_equal(np.asarray(want[1], dtype=np.float32), best_t)
        n_checked += 1
    assert n_checked >= 10


def test_type_erasure_shells_are_usable(built_pair):
    """a15: FeatureMap(Dt3Cpu), MatchStrategy(DefaultMatch()), SearchStrategy(...), OptimizeStrategy(...),
    PenaltyStrategy(...) (featuremap.h:98-124, matchstrategy.h:84-140, searchstrategy.h:73-124,
    optimizestrategy.h:66-118, penaltystrategy.h) are constructed and searched through; same result as the bare
    strategies and as the oracle."""
    import openfdcm_amd as openfdcm
    from openfdcm_amd import synthetic
    scene, dev, orc = built_pair
    tmpls = synthetic.templates(12, 10, 256, 4)
    fm = openfdcm.build_cpu_featuremap(scene, openfdcm.Dt3CpuParameters(depth=12, dt3Coeff=5.0, padding=1.2))
    erased = openfdcm.FeatureMap(fm)
    assert tuple(erased.get_feature_size()) == (dev.width, dev.height)
    assert repr(erased) == "<FeatureMap>" and isinstance(openfdcm.FeatureMap(erased), openfdcm.FeatureMap)
    for opt, kind, batch in [(openfdcm.BatchOptimize(4, openfdcm.ThreadPool(2)), O.BATCH_OPTIMIZE, 4),
                             (openfdcm.DefaultOptimize(num_threads=2), O.DEFAULT_OPTIMIZE, 1),
                             (openfdcm.IndulgentOptimize(2, openfdcm.ThreadPool(2)), O.INDULGENT_OPTIMIZE, 2)]:
        bare = openfdcm.search(openfdcm.DefaultMatch(), openfdcm.DefaultSearch(3, 5), opt, fm, tmpls, scene)
        wrapped = openfdcm.search(openfdcm.MatchStrategy(openfdcm.DefaultMatch()),
                                  openfdcm.SearchStrategy(openfdcm.DefaultSearch(3, 5)),
                                  openfdcm.OptimizeStrategy(opt), erased, tmpls, scene)
        want = O.search(orc, tmpls, scene, 3, 5, kind=kind, batch=batch, nthreads=4)
        assert len(bare) == len(wrapped) == len(want) > 0
        for a, b, w in zip(bare, wrapped, want):
            assert a.tmpl_idx == b.tmpl_idx == int(w["tmpl_idx"])
            assert np.float32(a.score) == np.float32(b.score) == w["score"]
            assert np.array_equal(a.transform, b.transform) and np.array_equal(a.transform.reshape(6), w["transform"])
    lens = openfdcm.get_template_lengths(tmpls)
    for pen in (openfdcm.ExponentialPenalty(1.5), openfdcm.DefaultPenalty()):
        p1 = openfdcm.penalize(pen, bare, lens)
        p2 = openfdcm.penalize(openfdcm.PenaltyStrategy(pen), wrapped, lens)
        assert [m.score for m in p1] == [m.score for m in p2]
        assert [m.score for m in openfdcm.sort_matches(p1)] == sorted(m.score for m in p1)
    # the erased map serves the C++ class's two other members too (featuremap.h:113-120)
    t0 = np.asarray(tmpls[0], dtype=np.float32) + np.float32(60)
    lo_hi = erased.minmax_translation(t0, (1.0, 0.25))
    assert _same_bits(lo_hi, O.minmax_translation(t0, (1.0, 0.25), (dev.width, dev.height), dev.scene_translation))
    tr = np.array([[0, 0], [2, 0.5], [-3, -0.75]], dtype=np.float32)
    assert _same_bits(erased.evaluate([t0], [tr])[0], O.evaluate(orc, t0, tr))
    # a concentric-range searcher goes through its shell as well
    conc = openfdcm.ConcentricRangeStrategy(3, 5, (128.0, 128.0), 10.0, 90.0)
    a = openfdcm.search(openfdcm.DefaultMatch(), conc, openfdcm.BatchOptimize(4), fm, tmpls, scene)
    b = openfdcm.search(openfdcm.MatchStrategy(openfdcm.DefaultMatch()), openfdcm.SearchStrategy(conc),
                        openfdcm.OptimizeStrategy(openfdcm.BatchOptimize(4)), erased, tmpls, scene)
    assert len(a) == len(b) and all(x.score == y.score for x, y in zip(a, b))


def test_pin_against_reference_script_runs_with_a_stand_in():
    """tools/pin_against_reference.py compares the real reference (where its wheel is installed) with the oracle and the HIP path;
    here openfdcm_amd plays the reference's part so that every line of the script runs: config 1 from the shipped assets and
    config 2 (three distances, two optimisers, the penalised and sorted list)."""
    import os
    import subprocess
    import sys
    from helpers import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pin_against_reference.py"), "--stand-in", "--templates", "60"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "reference pin: all identical" in out.stdout and "DIFF" not in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    assert out.stdout.count("ok   ") >= 20
