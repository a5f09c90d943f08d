"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): DT3 volume bit-exact; raw match list positionally equal in
tmpl_idx, score/transform within 1e-4 relative (+1e-6).  In practice scores are expected to be
bit-identical too; the tests report when they are not.
"""
import numpy as np
import pytest

from helpers import apply_transform, create_lines
from oracle import oracle as O

pytestmark = pytest.mark.gpu

REL, ABS = 1e-4, 1e-6  # north_star tolerance for score / transform


@pytest.fixture(scope="module")
def amd():
    import openfdcm_amd
    from openfdcm_amd import _capi
    import ctypes as C
    n = C.c_int()
    _capi.check(_capi.lib().fdcm_device_count(C.byref(n)))
    assert n.value >= 1, "no HIP device visible"
    return openfdcm_amd


def small_scene(S, n, seed):
    from openfdcm_amd import synthetic
    return synthetic.scene(S, n, seed)


def assert_volume_equal(dev, orc, what):
    a, b = dev.volume(), orc.volume()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    same = a.view(np.uint32) == b.view(np.uint32)
    if not same.all():
        bad = np.argwhere(~same)
        k, x, y = bad[0]
        raise AssertionError(f"{what}: {len(bad)} of {a.size} voxels differ; first at slice {k} x {x} y {y}: "
                             f"hip {a[k, x, y]!r} oracle {b[k, x, y]!r}")


def assert_matches_close(got, want, what):
    assert len(got) == len(want), (what, len(got), len(want))
    assert np.array_equal(got["tmpl_idx"], want["tmpl_idx"]), f"{what}: tmpl_idx differs positionally"
    def close(a, b):
        return np.abs(a - b) <= REL * np.maximum(np.abs(a), np.abs(b)) + ABS
    assert close(got["score"], want["score"]).all(), f"{what}: score outside 1e-4 relative"
    assert close(got["transform"], want["transform"]).all(), f"{what}: transform outside 1e-4 relative"
    return bool(np.array_equal(got["score"].view(np.uint32), want["score"].view(np.uint32))
                and np.array_equal(got["transform"].view(np.uint32), want["transform"].view(np.uint32)))


@pytest.mark.parametrize("dist", [O.L2_SQUARED, O.L2, O.L1])
@pytest.mark.parametrize("stage", [1, 2, 3])
@pytest.mark.parametrize("S,n,depth,seed", [(64, 12, 4, 3), (97, 25, 7, 4), (200, 40, 30, 5)])
def test_staged_build_bit_exact(amd, dist, stage, S, n, depth, seed):
    from openfdcm_amd.engine import DeviceFeatureMap
    scene = small_scene(S, n, seed)
    # non-trivial padding on the odd size: exercises scene translation and out-of-box clipping
    padding = 1.0 if S != 97 else 1.37
    dev = DeviceFeatureMap.build(scene, depth=depth, coeff=5.0, padding=padding, distance=dist, stop_after=stage)
    orc = O.build(scene, depth=depth, coeff=5.0, padding=padding, distance=dist, nthreads=4, stop_after=stage)
    assert (dev.width, dev.height, dev.depth) == (orc.W, orc.H, orc.depth)
    assert np.array_equal(dev.scene_translation, orc.translation)
    assert np.array_equal(dev.keys, orc.keys)
    assert_volume_equal(dev, orc, f"dist {dist} stage {stage} S {S}")


@pytest.mark.parametrize("dist", [O.L2, O.L2_SQUARED, O.L1])
def test_reference_kat_scenes_bit_exact(amd, dist):
    """The scenes of the reference's own build tests (dt3cpu.test.cpp:297-345, batchoptimize.test.cpp)."""
    from openfdcm_amd.engine import DeviceFeatureMap
    L = lambda *c: np.array(c, dtype=np.float32).T.reshape(4, -1)
    cases = [(L((2, 0, 5, 0)), 4, 1.0, 2.0), (L((4, 0, 10, 0)), 4, 1.0, 2.0),
             (L((0, 0, 0, 1), (0, 0, 1, 1), (0, 0, 1, 0), (0, 1, 1, 0), (1, 1, 1, 0)), 4, 50.0, 1.0),
             (L((15, 0, 15, 10), (5, 0, 15, 0)), 4, 1.0, 1.0), (L((3, 0, 6, 0), (0, 10, 7, 10)), 4, 1.0, 1.0),
             (L((0, 0, 1, 0)), 4, 1.0, 1.0)]
    for scene, depth, coeff, pad in cases:
        dev = DeviceFeatureMap.build(scene, depth=depth, coeff=coeff, padding=pad, distance=dist)
        orc = O.build(scene, depth=depth, coeff=coeff, padding=pad, distance=dist)
        assert_volume_equal(dev, orc, f"kat scene {scene.T.tolist()}")


def test_build_precision_known_answer(amd):
    """dt3cpu.test.cpp:318-345 through the GPU path itself."""
    from openfdcm_amd.engine import DeviceFeatureMap
    for scale, exp in [(1.0, [2, 3, 3, 3, 3, 3, 4]), (2.0, [3, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7, 9, 12])]:
        scene = np.array([[2], [0], [5], [0]], dtype=np.float32) * np.float32(scale)
        dev = DeviceFeatureMap.build(scene, depth=4, coeff=1.0, padding=2.0, distance=O.L2)
        k = O.closest_orientation(dev.keys, scene[:, 0])
        feat = dev.slice(k)
        assert np.allclose(feat[feat.shape[0] // 2], exp, atol=1e-5, rtol=0)


@pytest.mark.parametrize("name,dist", [("2", O.L2), ("2", O.L2_SQUARED), ("2", O.L1)])
def test_config2_build_bit_exact(amd, name, dist):
    """BASELINE config 2 geometry: 1024^2, depth 30, 200 lines -- whole volume, bit for bit."""
    from openfdcm_amd import synthetic
    from openfdcm_amd.engine import DeviceFeatureMap
    c = synthetic.CONFIGS[name]
    scene = synthetic.scene(c["S"], c["scene_lines"], 1)
    dev = DeviceFeatureMap.build(scene, depth=c["depth"], coeff=5.0, padding=1.0, distance=dist)
    orc = O.build(scene, depth=c["depth"], coeff=5.0, padding=1.0, distance=dist, nthreads=8)
    assert (dev.width, dev.height) == (c["S"], c["S"])
    assert_volume_equal(dev, orc, f"config {name} dist {dist}")


@pytest.mark.parametrize("kind,batch", [(O.BATCH_OPTIMIZE, 10), (O.BATCH_OPTIMIZE, 3), (O.DEFAULT_OPTIMIZE, 1),
                                        (O.BATCH_OPTIMIZE, 100), (O.INDULGENT_OPTIMIZE, 3)])
@pytest.mark.parametrize("dist", [O.L2, O.L2_SQUARED, O.L1])
def test_search_parity_small(amd, kind, batch, dist):
    from openfdcm_amd import synthetic
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, search_raw
    S = 256
    scene = synthetic.scene(S, 60, 11)
    tmpls = synthetic.templates(40, 13, S, 12) + synthetic.templates(10, 32, S, 13) + synthetic.templates(5, 3, S, 14)
    dev = DeviceFeatureMap.build(scene, depth=30, coeff=5.0, padding=1.0, distance=dist)
    orc = O.build(scene, depth=30, coeff=5.0, padding=1.0, distance=dist, nthreads=4)
    assert_volume_equal(dev, orc, "search fixture volume")
    tset = DeviceTemplates(tmpls)
    got = search_raw(dev, tset, scene, 4, 4, kind, batch)
    want, stats = O.search(orc, tmpls, scene, 4, 4, kind=kind, batch=batch, nthreads=4, return_stats=True)
    exact = assert_matches_close(got, want, f"kind {kind} batch {batch} dist {dist}")
    assert exact, "scores/transforms within tolerance but not bit-identical"
    st = dev.search_timing()
    assert st["candidates"] == stats[1]
    assert st["evaluations"] * 0 == 0  # populated


def test_search_parity_config2(amd):
    """BASELINE config 2: 1024^2, depth 30, 100 templates x 32 lines, DefaultSearch(4,4), BatchOptimize(10)."""
    from openfdcm_amd import synthetic
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, search_raw
    c, scene, tmpls = synthetic.make_config("2")
    dev = DeviceFeatureMap.build(scene, depth=c["depth"], coeff=5.0, padding=1.0, distance=c["distance"])
    orc = O.build(scene, depth=c["depth"], coeff=5.0, padding=1.0, distance=c["distance"], nthreads=8)
    got = search_raw(dev, DeviceTemplates(tmpls), scene, 4, 4, O.BATCH_OPTIMIZE, 10)
    want = O.search(orc, tmpls, scene, 4, 4, kind=O.BATCH_OPTIMIZE, batch=10, nthreads=8)
    assert len(want) > 1000
    assert assert_matches_close(got, want, "config 2"), "not bit-identical"


def test_search_edge_cases(amd):
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, search_raw
    tmpl = create_lines(10, 100)
    scene = tmpl
    dev = DeviceFeatureMap.build(scene, depth=30, coeff=5.0, padding=2.2, distance=O.L2)
    assert len(search_raw(dev, DeviceTemplates([]), scene, 4, 10)) == 0
    assert len(search_raw(dev, DeviceTemplates([np.zeros((4, 0))]), scene, 4, 10)) == 0
    empty = DeviceFeatureMap.build(np.zeros((4, 0)), depth=30, coeff=5.0, padding=2.2, distance=O.L2)
    assert (empty.width, empty.height, empty.depth) == (0, 0, 0)
    assert len(search_raw(empty, DeviceTemplates([tmpl]), np.zeros((4, 0)), 4, 10)) == 0
    # ragged: empty template between real ones keeps positional tmpl_idx
    orc = O.build(scene, depth=30, coeff=5.0, padding=2.2, distance=O.L2)
    tl = [tmpl, np.zeros((4, 0)), tmpl[:, :3], tmpl[:, :1]]
    got = search_raw(dev, DeviceTemplates(tl), scene, 4, 10, O.BATCH_OPTIMIZE, 10)
    want = O.search(orc, tl, scene, 4, 10, kind=O.BATCH_OPTIMIZE, batch=10)
    assert assert_matches_close(got, want, "ragged templates")
    assert set(np.unique(got["tmpl_idx"])) <= {0, 2, 3}
    # "all lines" limits: max_tmpl_lines is a size_t under min() (defaultsearch.cpp:38), max_scene_lines goes through
    # int casts (defaultsearch.h:42-46) and is well defined up to 2^31 - 1; both mean "every line" here
    big_t, big_s = 2 ** 40, 2 ** 31 - 1
    got = search_raw(dev, DeviceTemplates(tl), scene, big_t, big_s, O.BATCH_OPTIMIZE, 10)
    want = O.search(orc, tl, scene, big_t, big_s, kind=O.BATCH_OPTIMIZE, batch=10)
    assert len(want) > 200 and assert_matches_close(got, want, "unbounded search window")
    assert DeviceTemplates(tl).capacity(10, big_t, big_s) == 2 * (10 + 0 + 3 + 1) * 10
    # degenerate lines: zero-length scene line and zero-length template line
    sc2 = np.concatenate([scene, np.array([[5], [5], [5], [5]], dtype=np.float32)], axis=1)
    t2 = np.concatenate([tmpl, np.array([[1], [1], [1], [1]], dtype=np.float32)], axis=1)
    dev2 = DeviceFeatureMap.build(sc2, depth=30, coeff=5.0, padding=2.2, distance=O.L2)
    orc2 = O.build(sc2, depth=30, coeff=5.0, padding=2.2, distance=O.L2)
    assert_volume_equal(dev2, orc2, "degenerate scene line")
    got = search_raw(dev2, DeviceTemplates([t2]), sc2, 11, 11, O.BATCH_OPTIMIZE, 10)
    want = O.search(orc2, [t2], sc2, 11, 11, kind=O.BATCH_OPTIMIZE, batch=10)
    assert len(got) == len(want) and np.array_equal(got["tmpl_idx"], want["tmpl_idx"])
    ok = ~np.isnan(want["score"])
    assert np.array_equal(got["score"][ok].view(np.uint32), want["score"][ok].view(np.uint32))


def test_refused_search_leaves_nothing_in_flight(amd):
    """a search the library refuses (batch above 4096, a partial build) throws before anything is queued on the
    preparation stream; the next search on the same handle -- whose staging buffer it rewrites -- is the oracle's."""
    from openfdcm_amd import synthetic, _capi
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, search_raw
    scene, scene2 = synthetic.scene(256, 50, 8), synthetic.scene(256, 70, 9)
    tmpls = synthetic.templates(30, 12, 256, 10)
    tset = DeviceTemplates(tmpls)
    dev = DeviceFeatureMap.build(scene, depth=16, coeff=5.0, padding=1.0, distance=O.L2)
    orc = O.build(scene, depth=16, coeff=5.0, padding=1.0, distance=O.L2, nthreads=4)
    for _ in range(3):
        with pytest.raises(_capi.FdcmError, match="4096"):
            search_raw(dev, tset, scene2, 4, 4, O.BATCH_OPTIMIZE, 5000)
        got = search_raw(dev, tset, scene, 4, 4, O.BATCH_OPTIMIZE, 10)
        want = O.search(orc, tmpls, scene, 4, 4, kind=O.BATCH_OPTIMIZE, batch=10, nthreads=4)
        assert assert_matches_close(got, want, "after a refused search"), "not bit-identical"
    part = DeviceFeatureMap.build(scene, depth=16, coeff=5.0, padding=1.0, distance=O.L2, stop_after=2)
    with pytest.raises(_capi.FdcmError, match="partial build"):
        search_raw(part, tset, scene, 4, 4, O.BATCH_OPTIMIZE, 10)
    part.rebuild(scene)
    got = search_raw(part, tset, scene, 4, 4, O.BATCH_OPTIMIZE, 10)
    assert assert_matches_close(got, want, "the partial handle, rebuilt"), "not bit-identical"


def test_api_end_to_end_like_reference(amd):
    """tests/python/test_matching.py:45-104 of the reference, through the mirrored API."""
    openfdcm = amd
    threadpool = openfdcm.ThreadPool(4)
    search_strategy = openfdcm.DefaultSearch(4, 10)
    matcher = openfdcm.DefaultMatch()
    penalizer = openfdcm.ExponentialPenalty(1.5)
    tmpl = create_lines(10, 100)
    # The reference's own end-to-end test uses DefaultOptimize only (test_matching.py:52); with
    # BatchOptimize + L2_SQUARED the algorithm itself (oracle included) does not recover the pose.
    for optimizer_strategy in (openfdcm.DefaultOptimize(threadpool),):
        scene_transform = np.array([[-1, 0, 100], [0, -1, 100]], dtype=np.float32)
        scene = apply_transform(tmpl, scene_transform)
        for distance in [openfdcm.distance.L2, openfdcm.distance.L1, openfdcm.distance.L2_SQUARED]:
            params = openfdcm.Dt3CpuParameters(depth=30, dt3Coeff=5.0, padding=2.2, distance=distance)
            fm = openfdcm.build_cpu_featuremap(scene, params, threadpool)
            raw = openfdcm.search(matcher, search_strategy, optimizer_strategy, fm, [tmpl], scene)
            best = openfdcm.sort_matches(raw)[0].transform
            assert len(raw) == 80
            assert np.allclose(scene_transform[:, :2], best[:, :2], atol=1e-5)
            assert np.allclose(scene_transform[:, 2], best[:, 2], atol=1.0)
            scene_transform = np.array([[1, 0, 0], [0, 1, 0]], dtype=np.float32)
            scene = apply_transform(tmpl, scene_transform)
            fm = openfdcm.build_cpu_featuremap(scene, params, threadpool)
            raw = openfdcm.search(matcher, search_strategy, optimizer_strategy, fm, [tmpl], scene)
            pen = openfdcm.penalize(penalizer, raw, openfdcm.get_template_lengths([tmpl]))
            best = openfdcm.sort_matches(pen)[0].transform
            assert len(raw) == 80
            assert np.allclose(scene_transform[:, :2], best[:, :2], atol=1e-5)
            assert np.allclose(scene_transform[:, 2], best[:, 2], atol=1.0)
            fm0 = openfdcm.build_cpu_featuremap(np.zeros((4, 0)), params, threadpool)
            assert len(openfdcm.search(matcher, search_strategy, optimizer_strategy, fm0, [tmpl], np.zeros((4, 0)))) == 0
            scene = tmpl
            fm = openfdcm.build_cpu_featuremap(scene, params, threadpool)
            assert len(openfdcm.search(matcher, search_strategy, optimizer_strategy, fm, [], scene)) == 0
            assert len(openfdcm.search(matcher, search_strategy, optimizer_strategy, fm, [np.zeros((4, 0))], scene)) == 0


def test_adopted_volume_and_dt3_map_roundtrip(amd):
    """Dt3Cpu(dt3map, translation, size) constructor + get_dt3_map (matching.cpp:72-84)."""
    openfdcm = amd
    scene = small_scene(128, 20, 21)
    fm = openfdcm.build_cpu_featuremap(scene, openfdcm.Dt3CpuParameters(depth=8, dt3Coeff=5.0, padding=1.0))
    d = fm.get_dt3_map()
    orc = O.build(scene, depth=8, coeff=5.0, padding=1.0)
    for i, k in enumerate(sorted(d)):
        assert d[k].shape == (orc.H, orc.W)
        assert np.array_equal(d[k], orc.slice(i))
    fm2 = openfdcm.Dt3Cpu(d, fm.get_scene_translation(), fm.get_feature_size())
    tm = [create_lines(6, 20)]
    a = openfdcm.search(openfdcm.DefaultMatch(), openfdcm.DefaultSearch(3, 3), openfdcm.BatchOptimize(5), fm, tm, scene)
    b = openfdcm.search(openfdcm.DefaultMatch(), openfdcm.DefaultSearch(3, 3), openfdcm.BatchOptimize(5), fm2, tm, scene)
    assert len(a) == len(b) and all(x.score == y.score and x.tmpl_idx == y.tmpl_idx for x, y in zip(a, b))


def test_device_volume_layout_is_the_documented_one():
    """fdcm_featuremap_device_volume + _stride (include/fdcm.h): pixel (k, x, y) at k * stride + ((x/4) * H + y) * 4 + x%4,
    for a width that is not a multiple of 4 and for an adopted volume."""
    import ctypes
    from openfdcm_amd.engine import DeviceFeatureMap
    hip = ctypes.CDLL("libamdhip64.so")
    scene = small_scene(150, 24, 5)
    dev = DeviceFeatureMap.build(scene, depth=6, coeff=5.0, padding=1.3, distance=0)
    for fm in (dev, DeviceFeatureMap.from_volume(dev.keys, dev.volume(), dev.scene_translation)):
        W, H, m, stride = fm.width, fm.height, fm.depth, fm.device_slice_stride()
        assert stride >= ((W + 3) // 4) * H * 4
        raw = np.zeros(m * stride, dtype=np.float32)
        assert hip.hipMemcpy(ctypes.c_void_p(raw.ctypes.data), ctypes.c_void_p(fm.device_pointer()), ctypes.c_size_t(raw.nbytes), 2) == 0
        vol = fm.volume()  # [k][x][y] through fdcm_featuremap_slice
        x, y = np.meshgrid(np.arange(W), np.arange(H), indexing="ij")
        for k in range(m):
            assert np.array_equal(raw[k * stride + ((x // 4) * H + y) * 4 + x % 4], vol[k])


def test_concentric_range_strategy(amd):
    """ConcentricRangeStrategy through the mirrored API vs the oracle's restatement (concentricrange.cpp:29-60)."""
    openfdcm = amd
    from openfdcm_amd import synthetic
    S = 256
    scene = synthetic.scene(S, 60, 31)
    tmpls = synthetic.templates(12, 9, S, 32)
    fm = openfdcm.build_cpu_featuremap(scene, openfdcm.Dt3CpuParameters(depth=30, dt3Coeff=5.0, padding=1.0))
    orc = O.build(scene, depth=30, coeff=5.0, padding=1.0)
    for center, lo, hi in [((128.0, 128.0), 20.0, 90.0), ((40.0, 200.0), 0.0, 60.0), ((0.0, 0.0), 1000.0, 2000.0)]:
        strat = openfdcm.ConcentricRangeStrategy(3, 5, center, lo, hi)
        got = openfdcm.search(openfdcm.DefaultMatch(), strat, openfdcm.BatchOptimize(10), fm, tmpls, scene)
        want = O.search_concentric(orc, tmpls, scene, 3, 5, center, lo, hi, kind=O.BATCH_OPTIMIZE, batch=10)
        assert len(got) == len(want)
        assert [m.tmpl_idx for m in got] == list(want["tmpl_idx"])
        assert np.array_equal(np.array([m.score for m in got], dtype=np.float32).view(np.uint32), want["score"].view(np.uint32))


# ---------------------------------------------------------------- frame pipeline (include/fdcm.h)
@pytest.mark.parametrize("slots", [1, 3])
def test_frame_pipeline_matches_oracle_in_order(amd, slots):
    """Different frames in flight at once: every ticket returns the oracle's list for ITS scene."""
    from openfdcm_amd import synthetic, _capi
    from openfdcm_amd.engine import DeviceTemplates, FramePipeline
    S = 160
    scenes = [synthetic.scene(S, 18 + 4 * i, 20 + i) for i in range(7)]
    tmpls = synthetic.templates(12, 10, S, 9)
    tset = DeviceTemplates(tmpls)
    pipe = FramePipeline(tset, depth=15, coeff=5.0, padding=1.0, distance=O.L2, max_tmpl_lines=3, max_scene_lines=3,
                         optimizer=_capi.BATCH_OPTIMIZE, batch_size=10, tmpl_index_base=0, slots=slots)
    got, tickets = [], []
    for sc in scenes:
        if len(tickets) == slots:
            got.append(pipe.wait(tickets.pop(0)))
        tickets.append(pipe.submit(sc))
    while tickets:
        got.append(pipe.wait(tickets.pop(0)))
    for i, (sc, res) in enumerate(zip(scenes, got)):
        fm = O.build(sc, depth=15, coeff=5.0, padding=1.0, distance=O.L2)
        want = O.search(fm, tmpls, sc, 3, 3, kind=O.BATCH_OPTIMIZE, batch=10)
        assert assert_matches_close(res, want, f"pipeline frame {i}"), f"frame {i}: not bit-identical"
    assert pipe.last_build_timing["pass2_ms"] > 0 and pipe.last_search_timing["candidates"] > 0
    pipe.close()


def test_frame_pipeline_device_buffers_and_errors(amd):
    import ctypes as C
    import torch
    from openfdcm_amd import synthetic, _capi
    from openfdcm_amd.engine import DeviceTemplates, FramePipeline, DeviceFeatureMap, search_raw
    S = 128
    scene = synthetic.scene(S, 25, 31)
    tmpls = synthetic.templates(9, 8, S, 32)
    tset = DeviceTemplates(tmpls)
    pipe = FramePipeline(tset, depth=10, coeff=5.0, padding=1.0, distance=O.L2_SQUARED, max_tmpl_lines=4,
                         max_scene_lines=4, optimizer=_capi.DEFAULT_OPTIMIZE, batch_size=1, tmpl_index_base=100, slots=2)
    cap = tset.capacity(25, 4, 4)
    bufs = [torch.empty(cap * 32, dtype=torch.uint8, device="cuda") for _ in range(2)]
    t0 = pipe.submit(scene, bufs[0].data_ptr())
    t1 = pipe.submit(scene, bufs[1].data_ptr())
    with pytest.raises(_capi.FdcmError):  # both slots hold uncollected frames
        pipe.submit(scene)
    n0, n1 = pipe.wait(t0), pipe.wait(t1)
    with pytest.raises(_capi.FdcmError):  # a ticket is collected once
        pipe.wait(t0)
    fm = DeviceFeatureMap.build(scene, depth=10, coeff=5.0, padding=1.0, distance=O.L2_SQUARED)
    want = search_raw(fm, tset, scene, 4, 4, _capi.DEFAULT_OPTIMIZE, 1, 100)
    assert n0 == n1 == len(want) and want["tmpl_idx"].min() >= 100
    for b in bufs:
        got = b[: n0 * 32].cpu().numpy().view(_capi.MATCH_DTYPE)
        assert got.tobytes() == want.tobytes()
    # an empty scene is a valid frame (no matches), as in the blocking calls
    t2 = pipe.submit(np.zeros((4, 0), dtype=np.float32))
    assert len(pipe.wait(t2)) == 0
    pipe.close()


# ---------------------------------------------------------------- BASELINE configs 1, 2', 3, 4 (per-GPU shape), 5
def _mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def test_config1_full_size_assets(amd):
    """Config 1 at full size (notebooks/general_template_matching_example.ipynb:275-306 on the line files shipped
    beside it): obj_04/scene_0 (646 lines, 487 x 487 feature map), all 122 templates, depth 30,
    coeff 5, L2, padding 1.0.  Whole volume bit for bit; match lists for DefaultSearch(4,4) + BatchOptimize(10) and
    for the notebook's own DefaultSearch(3,10) + BatchOptimize(5); the notebook's tail (ExponentialPenalty(1.5) +
    sort + first 30) on the device against the product's host tail."""
    import os
    import ctypes as C
    from helpers import ROOT
    from openfdcm_amd import lineio, _capi
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, search_raw, topk
    d = os.path.join(ROOT, "tests", "golden", "obj_04")
    scene = lineio.read(os.path.join(d, "scene_0.scene"))
    tmpls = [lineio.read(os.path.join(d, f"template_{i}.tmpl")) for i in range(122)]
    assert scene.shape == (4, 646) and min(t.shape[1] for t in tmpls) >= 12 and max(t.shape[1] for t in tmpls) <= 31
    dev = DeviceFeatureMap.build(scene, depth=30, coeff=5.0, padding=1.0, distance=O.L2)
    orc = O.build(scene, depth=30, coeff=5.0, padding=1.0, distance=O.L2, nthreads=8)
    assert (dev.width, dev.height, dev.depth) == (487, 487, 30) == (orc.W, orc.H, orc.depth)
    assert np.array_equal(dev.scene_translation, orc.translation)
    assert_volume_equal(dev, orc, "config 1")
    tset = DeviceTemplates(tmpls)
    for maxT, maxS, B in [(4, 4, 10), (3, 10, 5)]:
        got = search_raw(dev, tset, scene, maxT, maxS, O.BATCH_OPTIMIZE, B)
        want = O.search(orc, tmpls, scene, maxT, maxS, kind=O.BATCH_OPTIMIZE, batch=B, nthreads=8)
        assert len(want) > 3000
        assert assert_matches_close(got, want, f"config 1 DefaultSearch({maxT},{maxS})"), "not bit-identical"
    # tail of the notebook: penalize(ExponentialPenalty(1.5)) + sort_matches + [:30]
    rec = np.array(got, copy=True)
    lens = tset.lengths()
    _capi.check(_capi.lib().fdcm_penalize(1, 1.5, C.c_void_p(rec.ctypes.data), len(rec), _capi.fptr(lens), len(lens)))
    want_tail = rec[np.argsort(rec["score"], kind="stable")[:30]]
    assert topk(dev, tset, 30, 1, 1.5).tobytes() == want_tail.tobytes()


def test_config2p_match_list_1000_templates(amd):
    """Config 2' (the headline of BASELINE.json:metric): 1024^2, depth 30, L2, 1000 templates x 32 lines -- the whole
    match list against the oracle."""
    import os
    from openfdcm_amd import synthetic
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, search_raw
    c, scene, tmpls = synthetic.make_config("2p")
    nt = min(32, os.cpu_count() or 1)
    dev = DeviceFeatureMap.build(scene, depth=c["depth"], coeff=5.0, padding=1.0, distance=c["distance"])
    orc = O.build(scene, depth=c["depth"], coeff=5.0, padding=1.0, distance=c["distance"], nthreads=nt)
    got = search_raw(dev, DeviceTemplates(tmpls), scene, 4, 4, O.BATCH_OPTIMIZE, 10)
    want = O.search(orc, tmpls, scene, 4, 4, kind=O.BATCH_OPTIMIZE, batch=10, nthreads=nt)
    assert len(tmpls) == 1000 and len(want) > 20000
    assert assert_matches_close(got, want, "config 2'"), "not bit-identical"
    dev.close()
    # The same frame through the reference's Python names only (matching.cpp:116-130,279-307): the lazy MatchList holds
    # the same records, the tail equals the oracle's penalize + std::sort (ties included), and the calls cost
    # milliseconds, not the 0.58 s that a Python object per match cost in round 5.
    import time
    openfdcm = amd
    params = openfdcm.Dt3CpuParameters(depth=c["depth"], dt3Coeff=5.0, padding=1.0, distance=openfdcm.distance(c["distance"]))
    strategy, optimizer, penalty = openfdcm.DefaultSearch(4, 4), openfdcm.BatchOptimize(10), openfdcm.ExponentialPenalty(1.5)
    walls = []
    for i in range(6):
        t0 = time.perf_counter()
        featuremap = openfdcm.build_cpu_featuremap(scene, params)
        matches = openfdcm.search(openfdcm.DefaultMatch(), strategy, optimizer, featuremap, tmpls, scene)
        lens = openfdcm.get_template_lengths(tmpls)
        best = openfdcm.sort_matches(openfdcm.penalize(penalty, matches, lens))
        walls.append(time.perf_counter() - t0)
    assert isinstance(matches, openfdcm.MatchList) and matches.records().tobytes() == np.asarray(want).tobytes()
    assert best.records().tobytes() == O.sort_matches(O.penalize(want, np.array(lens, dtype=np.float32), 1.5)).tobytes()
    assert best[0].score == min(m.score for m in best[:50]) and best[0].transform.shape == (2, 3)
    t0 = time.perf_counter()
    n = sum(1 for m in matches if m.tmpl_idx >= 0)
    t_iter = time.perf_counter() - t0
    assert n == len(want)
    assert min(walls[2:]) < 0.010, f"API frame {min(walls[2:]) * 1e3:.2f} ms"
    assert t_iter < 0.030, f"iterating {n} matches {t_iter * 1e3:.1f} ms"
    openfdcm.clear_featuremap_pool()
    openfdcm.clear_template_cache()


def test_api_template_cache_and_featuremap_pool(amd):
    """search() / get_template_lengths() keep the device copy of a template list they have seen (same list object, same
    bytes) and notice an in-place edit; build_cpu_featuremap() reuses the device handle of a dropped Dt3Cpu."""
    import openfdcm_amd as api
    from openfdcm_amd import synthetic
    S = 200
    scene, scene2 = synthetic.scene(S, 40, 3), synthetic.scene(S, 50, 4)
    tmpls = synthetic.templates(20, 10, S, 5)
    params = api.Dt3CpuParameters(depth=12, dt3Coeff=5.0, padding=1.1)
    args = (api.DefaultMatch(), api.DefaultSearch(3, 4), api.BatchOptimize(5))
    api.clear_template_cache()
    api.clear_featuremap_pool()
    fm = api.build_cpu_featuremap(scene, params)
    a = api.search(*args, fm, tmpls, scene)
    tset = api._template_cache._entries[0][3]
    b = api.search(*args, fm, tmpls, scene)
    assert api._template_cache._entries[0][3] is tset and len(api._template_cache._entries) == 1 and a == b
    assert api.get_template_lengths(tmpls) == api.get_template_lengths([t.copy() for t in tmpls])
    orc = O.build(scene, depth=12, coeff=5.0, padding=1.1)
    assert a.records().tobytes() == np.asarray(O.search(orc, tmpls, scene, 3, 4, kind=O.BATCH_OPTIMIZE, batch=5)).tobytes()
    tmpls[7][:, 2] += 11.0                                     # edited in place: same list, same arrays, other bytes
    c = api.search(*args, fm, tmpls, scene)
    assert api._template_cache._entries[0][3] is not tset
    assert c.records().tobytes() == np.asarray(O.search(orc, tmpls, scene, 3, 4, kind=O.BATCH_OPTIMIZE, batch=5)).tobytes()
    tmpls.append(tmpls[0].copy())                               # grown
    d = api.search(*args, fm, tmpls, scene)
    assert d.records().tobytes() == np.asarray(O.search(orc, tmpls, scene, 3, 4, kind=O.BATCH_OPTIMIZE, batch=5)).tobytes()
    # the pool: a dropped feature map's handle serves the next build with the same parameters (and only those)
    h = fm._fm._h.value
    del fm
    assert sum(len(v) for v in api._featuremap_pool._idle.values()) == 1
    other = api.build_cpu_featuremap(scene2, api.Dt3CpuParameters(depth=8, dt3Coeff=5.0, padding=1.1))
    assert other._fm._h.value != h
    fm2 = api.build_cpu_featuremap(scene2, params)
    assert fm2._fm._h.value == h and not api._featuremap_pool._idle[fm2._pool_key]
    orc2 = O.build(scene2, depth=12, coeff=5.0, padding=1.1)
    assert_volume_equal(fm2._fm, orc2, "pooled handle, new scene")
    e = api.search(*args, fm2, tmpls, scene2)
    assert e.records().tobytes() == np.asarray(O.search(orc2, tmpls, scene2, 3, 4, kind=O.BATCH_OPTIMIZE, batch=5)).tobytes()
    assert tuple(fm2.get_feature_size()) == (orc2.W, orc2.H) and np.array_equal(fm2.get_scene_translation(), orc2.translation)
    del fm2, other
    api.clear_featuremap_pool()
    api.clear_template_cache()
    assert not api._featuremap_pool._idle and not api._template_cache._entries


@pytest.fixture(scope="module")
def config3_maps(amd):
    """Configs 3 and 4 share the scene: 2048^2, depth 60, L2_SQUARED (1 GB volume), built once on both sides."""
    import os
    from openfdcm_amd import synthetic
    from openfdcm_amd.engine import DeviceFeatureMap
    c = synthetic.CONFIGS["3"]
    scene = synthetic.scene(c["S"], c["scene_lines"], 1)
    nt = min(32, os.cpu_count() or 1)
    dev = DeviceFeatureMap.build(scene, depth=c["depth"], coeff=5.0, padding=1.0, distance=c["distance"])
    orc = O.build(scene, depth=c["depth"], coeff=5.0, padding=1.0, distance=c["distance"], nthreads=nt)
    yield c, scene, dev, orc, nt
    dev.close()


def test_config3_build_and_search(amd, config3_maps):
    """Config 3: 2048^2, depth 60, L2_SQUARED -- the whole 1 GB volume bit for bit (compared slice by slice)
    and the match list of all 1000 templates."""
    from openfdcm_amd import synthetic
    from openfdcm_amd.engine import DeviceTemplates, search_raw
    c, scene, dev, orc, nt = config3_maps
    tmpls = synthetic.templates(c["T"], c["n"], c["S"], 2)
    assert len(tmpls) == 1000
    assert (dev.width, dev.height, dev.depth) == (2048, 2048, 60) == (orc.W, orc.H, orc.depth)
    for k in range(dev.depth):
        a, b = dev.slice(k), orc.slice(k)
        bad = int(np.sum(a.view(np.uint32) != b.view(np.uint32)))
        assert bad == 0, f"config 3 slice {k}: {bad} pixels differ"
    got = search_raw(dev, DeviceTemplates(tmpls), scene, 4, 4, O.BATCH_OPTIMIZE, 10)
    want = O.search(orc, tmpls, scene, 4, 4, kind=O.BATCH_OPTIMIZE, batch=10, nthreads=nt)
    assert len(want) > 10000
    assert assert_matches_close(got, want, "config 3"), "not bit-identical"


def test_config4_per_gpu_shard(amd, config3_maps):
    """Config 4's per-GPU shape on one GPU: 2048^2, depth 60, L2_SQUARED, the shard of rank 1 of 8 -- templates
    1000..1999 of the 8000, tmpl_index_base 1000 -- against the oracle's list for the same templates."""
    from openfdcm_amd import synthetic
    from openfdcm_amd.dist import shard_range
    from openfdcm_amd.engine import DeviceTemplates, search_raw
    c, scene, dev, orc, nt = config3_maps
    all_t = synthetic.templates(2000, c["n"], c["S"], 2)  # the generator is a stream: the first 2000 of the 8000
    lo, hi = shard_range(8000, 1, 8)
    assert (lo, hi) == (1000, 2000)
    shard = all_t[lo:hi]
    got = search_raw(dev, DeviceTemplates(shard), scene, 4, 4, O.BATCH_OPTIMIZE, 10, lo)
    want = O.search(orc, shard, scene, 4, 4, kind=O.BATCH_OPTIMIZE, batch=10, nthreads=nt)
    want = np.array(want, copy=True)
    want["tmpl_idx"] += lo
    assert len(want) > 20000 and got["tmpl_idx"].min() >= lo and got["tmpl_idx"].max() < hi
    assert assert_matches_close(got, want, "config 4 shard 1/8"), "not bit-identical"


def test_config5_sampled_slices(amd):
    """Config 5 (stress): 4096^2, depth 180, L1 -- a 12 GB volume; seven slices compared bit for bit, and the match
    list of a whole per-GPU shard (2000 templates x 32 lines: rank 3 of 8, tmpl_index_base 6000).  The oracle's
    volume needs ~30 GB of host memory: runs when MemAvailable >= 48 GB (always on the GPU box)."""
    import os
    if _mem_available_gb() < 48 and os.environ.get("FDCM_TEST_STRESS") != "1":
        pytest.skip("needs 48 GB of free host memory for the oracle's 12 GB volume")
    from openfdcm_amd import synthetic
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, search_raw
    c = synthetic.CONFIGS["5"]
    scene = synthetic.scene(c["S"], c["scene_lines"], 1)
    nt = os.cpu_count() or 1
    dev = DeviceFeatureMap.build(scene, depth=c["depth"], coeff=5.0, padding=1.0, distance=c["distance"])
    orc = O.build(scene, depth=c["depth"], coeff=5.0, padding=1.0, distance=c["distance"], nthreads=nt)
    for k in sorted(set(np.linspace(0, dev.depth - 1, 7).astype(int))):
        a, b = dev.slice(int(k)), orc.slice(int(k))
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"config 5 slice {k}"
    tmpls = synthetic.templates(8000, c["n"], c["S"], 2)[6000:8000]  # the generator is a stream: templates 6000..7999 of the 16000
    got = search_raw(dev, DeviceTemplates(tmpls), scene, 4, 4, O.BATCH_OPTIMIZE, 10, 6000)
    want = np.array(O.search(orc, tmpls, scene, 4, 4, kind=O.BATCH_OPTIMIZE, batch=10, nthreads=nt), copy=True)
    want["tmpl_idx"] += 6000
    assert len(want) > 20000
    assert assert_matches_close(got, want, "config 5 shard"), "not bit-identical"


# ---------------------------------------------------------------- device tail (include/fdcm.h, fdcm_topk)
@pytest.mark.parametrize("penalty,tau", [(None, 1.0), (0, 1.0), (1, 1.5), (1, 0.7)])
def test_device_topk_equals_penalize_sort_slice(amd, penalty, tau):
    """fdcm_topk == penalize() + stable sort by score + [:k] on the host, bit for bit."""
    import ctypes as C
    import torch
    from openfdcm_amd import synthetic, _capi
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, search_raw, search_into, topk
    S = 256
    scene = synthetic.scene(S, 60, 11)
    tmpls = synthetic.templates(40, 13, S, 12) + synthetic.templates(10, 32, S, 13) + synthetic.templates(5, 3, S, 14)
    fm = DeviceFeatureMap.build(scene, depth=30, coeff=5.0, padding=1.0, distance=O.L2)
    tset = DeviceTemplates(tmpls)
    raw = search_raw(fm, tset, scene, 4, 4, _capi.BATCH_OPTIMIZE, 10, 7)  # tmpl_index_base 7
    assert len(raw) > 500

    def host_tail(k):
        rec = np.array(raw, copy=True)
        if penalty is not None:
            rec["tmpl_idx"] -= 7
            lens = tset.lengths()
            _capi.check(_capi.lib().fdcm_penalize(penalty, tau, C.c_void_p(rec.ctypes.data), len(rec), _capi.fptr(lens),
                                                  len(lens)))
            rec["tmpl_idx"] += 7
        return rec[np.argsort(rec["score"], kind="stable")[:k]]

    for k in (1, 10, 100, len(raw), len(raw) + 50):
        got = topk(fm, tset, k, penalty, tau, tmpl_index_base=7)
        want = host_tail(k)
        assert got.tobytes() == want.tobytes(), (penalty, tau, k)
    # the same from a caller-owned device buffer (the sharded path)
    cap = tset.capacity(60, 4, 4)
    buf = torch.empty(cap * 32, dtype=torch.uint8, device="cuda")
    n = search_into(fm, tset, scene, 4, 4, _capi.BATCH_OPTIMIZE, 10, 7, buf.data_ptr())
    assert n == len(raw)
    got = topk(fm, tset, 25, penalty, tau, tmpl_index_base=7, device_ptr=buf.data_ptr(), n=n)
    assert got.tobytes() == host_tail(25).tobytes()
    assert len(topk(fm, tset, 0, penalty, tau, tmpl_index_base=7)) == 0


def test_search_many_templates_generic_work_list(amd):
    """More than 16384 pair slots: the work list is built by the generic (memory-resident) path."""
    from openfdcm_amd import synthetic, _capi
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, search_raw
    S = 200
    scene = synthetic.scene(S, 45, 41)
    tmpls = synthetic.templates(1500, 6, S, 42) + [np.zeros((4, 0), np.float32)] + synthetic.templates(30, 2, S, 43)
    dev = DeviceFeatureMap.build(scene, depth=20, coeff=5.0, padding=1.0, distance=O.L2)
    orc = O.build(scene, depth=20, coeff=5.0, padding=1.0, distance=O.L2, nthreads=8)
    got = search_raw(dev, DeviceTemplates(tmpls), scene, 4, 3, _capi.BATCH_OPTIMIZE, 10)
    want = O.search(orc, tmpls, scene, 4, 3, kind=O.BATCH_OPTIMIZE, batch=10, nthreads=8)
    assert len(tmpls) * 4 * 3 > 16384 and len(want) > 20000
    assert assert_matches_close(got, want, "1531 templates"), "not bit-identical"


@pytest.mark.parametrize("env,cases,seed", [({}, 80, 11),
                                            ({"FDCM_SWEEP_MINCOLS": "2"}, 60, 12),
                                            ({"FDCM_SWEEP_MINCOLS": "1", "FDCM_SWEEP_ORDER": "1"}, 40, 13),
                                            ({"FDCM_L2_SWEEP": "literal"}, 40, 14),
                                            ({"FDCM_FORCE_HOST_BINS": "1"}, 40, 15),
                                            ({"FDCM_INT_XC": "256"}, 30, 16),
                                            ({"FDCM_INT_XC": "128"}, 30, 17),
                                            ({"FDCM_SEARCH_FLAT": "1"}, 30, 18),
                                            ({"FDCM_SEARCH_COMPACT2": "1"}, 30, 19),
                                            ({"FDCM_SWEEP_STEAL": "1"}, 80, 20),
                                            ({"FDCM_SWEEP_STEAL": "1", "FDCM_SWEEP_MINCOLS": "2"}, 60, 21),
                                            ({"FDCM_SWEEP_STEAL": "1", "FDCM_SWEEP_MINCOLS": "1", "FDCM_SWEEP_ORDER": "1"}, 40, 22),
                                            ({"FDCM_SWEEP_STEAL": "0"}, 40, 23)],
                         ids=["default", "8-ranges-on-small-slices", "1-column-ranges+launch-order", "literal-one-wave-per-chunk", "host-libm-bins",
                              "integral-252-chain-blocks", "integral-124-chain-blocks", "search-with-64-bit-addresses",
                              "two-kernel-compaction", "ranges-taken-over-at-every-chance", "taken-over-on-small-slices",
                              "taken-over-1-column-blocks+launch-order", "equal-count-ranges-only"])
def test_randomised_cases(amd, env, cases, seed):
    """Random (scene, depth, distance, padding, coefficient, optimiser, template set) cases: volume and match list
    bit for bit (tools/fuzz_parity.py).  The variants force the paths of the L2 sweep that the default sizes do not
    take: all eight column ranges per row on slices with few columns (ranges of two columns and of one: junctions that
    pop whole ranges, rows whose stack is one entry), the launch order from a cost table, and the literal
    one-wave-per-chunk kernel that sizes above the exact-integer bound take; the orientation bins of the aligned template lines from the host
    libm (the path a host whose atanf differs from the device restatement takes); the wide-block forms of the steep
    line integral that only large volumes select; the search with 64-bit flat addresses (what volumes of 4 GB and more take);
    the two-kernel form of the positional compaction (what searches of more than 65 536 candidates take); the sweep's dynamic cuts
    forced at every chance (also on slices of few columns and with blocks of one column) and forbidden."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), str(cases), str(seed)],
                         capture_output=True, text=True, timeout=600, env={**os.environ, **env})
    assert out.returncode == 0 and f"{cases} random cases identical" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    # the dynamic cuts of the L2 sweep (a wave out of columns begins a new range in what nobody has started): the variants that
    # force them at every chance must have taken ranges over, the one that forbids them none
    import re
    taken = int(re.search(r"(\d+) ranges taken over", out.stdout).group(1))
    if env.get("FDCM_SWEEP_STEAL") == "1" and env.get("FDCM_SWEEP_MINCOLS") != "1":  # (ranges of one column leave nothing to take over)
        assert taken > cases, out.stdout[-500:]
    if env.get("FDCM_SWEEP_STEAL") == "0" or env.get("FDCM_L2_SWEEP") == "literal":
        assert taken == 0, out.stdout[-500:]


@pytest.mark.parametrize("env", [{}, {"FDCM_SWEEP_STEAL": "1"}], ids=["default-cuts", "cuts-at-every-chance"])
def test_dynamic_cuts_at_full_size(amd, env):
    """The L2 sweep's waves cut a row's columns among themselves as they go (whoever runs dry takes over the far half of what
    nobody has started): the cuts depend on timing, so every build of a scene is cut differently -- and must be the oracle's
    volume bit for bit every time (tools/soak_volumes.py: config 2 scenes of seeds the other tests do not use, L2 and L2^2, each
    built twice on one handle, and one config 3 scene)."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_volumes.py"), "41", "3", "2", "1"], capture_output=True, text=True,
                         timeout=900, env={**os.environ, **env})
    assert out.returncode == 0 and "14 full-size builds identical" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert int(re.search(r"(\d+) ranges taken over", out.stdout).group(1)) > 0, out.stdout[-300:]


@pytest.mark.parametrize("env", [{}, {"FDCM_SWEEP_ORDER": "1"}], ids=["default", "launch-order-from-history"])
def test_rebuilds_of_one_handle(amd, env):
    """A handle rebuilt over scenes of changing content and size (tools/rebuild_parity.py): every volume bit for bit.
    The second variant forces the L2 sweep's launch order by the previous build's chunk times at these small sizes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "rebuild_parity.py"), "8", "3"],
                         capture_output=True, text=True, timeout=600, env={**os.environ, **env})
    assert out.returncode == 0 and "24 rebuilds identical" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_stage_timing_switch_changes_events_not_results(amd):
    """fdcm_featuremap_stage_timing(fm, 0): builds and searches record no events (the device times of the timings are 0,
    the counters are not) -- the volume and the match list are what they were."""
    from openfdcm_amd import synthetic, _capi
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, search_raw
    scene = synthetic.scene(256, 60, 5)
    tmpls = synthetic.templates(40, 9, 256, 6)
    ts = DeviceTemplates(tmpls)
    dev = DeviceFeatureMap.build(scene, depth=12, coeff=5.0, padding=1.0, distance=O.L2)
    want_vol = [dev.slice(k).copy() for k in range(12)]
    want = np.array(search_raw(dev, ts, scene, 4, 4, _capi.BATCH_OPTIMIZE, 10), copy=True)
    t_on = dev.build_timing()
    assert t_on["pass2_ms"] > 0 and t_on["integral_ms"] > 0 and t_on["total_ms"] > 0
    dev.stage_timing(False)
    for _ in range(3):
        dev.rebuild(scene)
        got = np.array(search_raw(dev, ts, scene, 4, 4, _capi.BATCH_OPTIMIZE, 10), copy=True)
        t_off, s_off = dev.build_timing(), dev.search_timing()
        assert t_off["pass2_ms"] == 0 and t_off["integral_ms"] == 0 and s_off["kernel_ms"] == 0 and s_off["evaluations"] > 0
        assert got.tobytes() == want.tobytes()
    assert all(np.array_equal(dev.slice(k).view(np.uint32), want_vol[k].view(np.uint32)) for k in range(12))
    dev.stage_timing(2)  # the build's and the search's spans only
    dev.rebuild(scene)
    got = np.array(search_raw(dev, ts, scene, 4, 4, _capi.BATCH_OPTIMIZE, 10), copy=True)
    t2, s2 = dev.build_timing(), dev.search_timing()
    assert t2["pass2_ms"] == 0 and t2["span_ms"] > 0 and t2["total_ms"] >= t2["span_ms"] and s2["kernel_ms"] > 0
    assert got.tobytes() == want.tobytes()
    dev.stage_timing(True)
    dev.rebuild(scene)
    t1 = dev.build_timing()
    assert t1["pass2_ms"] > 0 and t1["span_ms"] >= t1["pass2_ms"]
    dev.close()


def test_frame_pipeline_reports_a_failed_frame_and_keeps_going(amd):
    """A frame whose build fails (feature size above the supported maximum) is reported by its wait();
    the slot stays usable."""
    from openfdcm_amd import synthetic, _capi
    from openfdcm_amd.engine import DeviceTemplates, FramePipeline
    S = 128
    good = synthetic.scene(S, 20, 77)
    huge = np.array([[0.0, 0.0], [0.0, 0.0], [40000.0, 0.0], [0.0, 40000.0]], dtype=np.float32)  # 40001 x 40001
    tset = DeviceTemplates(synthetic.templates(5, 8, S, 78))
    pipe = FramePipeline(tset, depth=8, coeff=5.0, padding=1.0, distance=O.L2, slots=2)
    t0, t1, = pipe.submit(good), pipe.submit(huge)
    first = np.array(pipe.wait(t0), copy=True)
    with pytest.raises(_capi.FdcmError, match="feature size"):
        pipe.wait(t1)
    t2, t3 = pipe.submit(good), pipe.submit(good)  # t3 runs on the slot whose last frame failed
    assert pipe.wait(t2).tobytes() == first.tobytes() and pipe.wait(t3).tobytes() == first.tobytes()
    pipe.close()


# ---------------------------------------------------------------- template shards behind the C ABI (fdcm_sharded_*)
@pytest.mark.parametrize("always_collective", [False, True], ids=["direct", "through-rccl"])
def test_sharded_engine_one_device(amd, always_collective):
    """fdcm_sharded_* with one device (the test box has one GPU).  With FDCM_SHARDED_ALWAYS_COLLECTIVE the shard's
    records go to the first device through the grouped RCCL send/recv that multi-GPU runs use (one per frame), so the
    driver's suite executes the RCCL path.  Full list and top-k against the oracle and the single-device calls."""
    import ctypes as C
    from openfdcm_amd import synthetic, _capi
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, ShardedEngine, search_raw, topk
    S = 256
    tmpls = synthetic.templates(30, 13, S, 12) + [np.zeros((4, 0), np.float32)] + synthetic.templates(9, 32, S, 13)
    eng = ShardedEngine(tmpls, n_devices=1, depth=30, coeff=5.0, padding=1.0, distance=O.L2, always_collective=always_collective)
    assert eng.info()["shard_begin"] == [0, len(tmpls)]
    frames = 0
    for seed, n_lines in [(11, 60), (12, 45), (13, 0)]:
        scene = synthetic.scene(S, n_lines, seed) if n_lines else np.zeros((4, 0), np.float32)
        got = eng.search(scene, 4, 4, _capi.BATCH_OPTIMIZE, 10)
        frames += 1
        if n_lines == 0:
            assert len(got) == 0
            continue
        orc = O.build(scene, depth=30, coeff=5.0, padding=1.0, distance=O.L2, nthreads=4)
        want = O.search(orc, tmpls, scene, 4, 4, kind=O.BATCH_OPTIMIZE, batch=10, nthreads=4)
        assert assert_matches_close(got, want, f"sharded search, scene {seed}"), "not bit-identical"
        # top-k: the shard's k best through the same exchange == fdcm_topk on one device
        fm = DeviceFeatureMap.build(scene, depth=30, coeff=5.0, padding=1.0, distance=O.L2)
        tset = DeviceTemplates(tmpls)
        search_raw(fm, tset, scene, 4, 4, _capi.BATCH_OPTIMIZE, 10)
        for k in (1, 25, len(want) + 10):
            a = eng.search_topk(scene, 4, 4, k, penalty=1, tau=1.5)
            frames += 1
            b = topk(fm, tset, k, 1, 1.5)
            assert a.tobytes() == b.tobytes(), (seed, k)
    info = eng.info()
    if always_collective:
        assert info["collectives"] >= frames - 1 and info["bytes_moved"] > 0  # the empty frame moves nothing
    else:
        assert info["collectives"] == 0
    bt, st = eng.timing(0)
    assert bt["pass2_ms"] >= 0 and st["candidates"] >= 0
    eng.close()


def test_sharded_engine_frames_in_flight(amd):
    """fdcm_sharded_submit / _wait with three frame slots per device and the RCCL exchange: six different scenes are in
    flight three at a time, every collected frame equals the blocking single-device search of its own scene (a slot or
    frame mix-up in the exchange would show), top-k frames interleaved, tickets checked."""
    from openfdcm_amd import synthetic, _capi
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, ShardedEngine, search_raw, topk
    S = 256
    tmpls = synthetic.templates(40, 11, S, 21)
    scenes = [synthetic.scene(S, 30 + 7 * i, 40 + i) for i in range(6)] + [np.zeros((4, 0), np.float32)]
    eng = ShardedEngine(tmpls, n_devices=1, depth=16, coeff=5.0, padding=1.0, distance=O.L2_SQUARED, always_collective=True)
    eng.set_frames_in_flight(3)
    fm = DeviceFeatureMap.build(scenes[0], depth=16, coeff=5.0, padding=1.0, distance=O.L2_SQUARED)
    tset = DeviceTemplates(tmpls)
    want, want_top = [], []
    for sc in scenes:
        fm.rebuild(sc)
        want.append(np.array(search_raw(fm, tset, sc, 4, 4, _capi.BATCH_OPTIMIZE, 10), copy=True))
        want_top.append(np.array(topk(fm, tset, 17, 1, 1.5), copy=True) if sc.shape[1] else want[-1][:0])
    order = list(range(len(scenes))) * 2
    pending = []
    for step, i in enumerate(order):
        if len(pending) == 3:
            t, j, top = pending.pop(0)
            got = eng.wait(t)
            assert got.tobytes() == (want_top[j] if top else want[j]).tobytes(), (t, j, top)
        top = step % 3 == 1
        t = eng.submit(scenes[i], 4, 4, _capi.BATCH_OPTIMIZE, 10, k=17 if top else None, penalty=1, tau=1.5)
        assert t == step
        pending.append((t, i, top))
    with pytest.raises(_capi.FdcmError, match="not been waited for"):
        eng.submit(scenes[0], 4, 4)
    with pytest.raises(_capi.FdcmError, match="in flight"):
        eng.set_frames_in_flight(2)
    for t, j, top in pending:
        assert eng.wait(t).tobytes() == (want_top[j] if top else want[j]).tobytes(), (t, j, top)
    with pytest.raises(_capi.FdcmError, match="ticket"):
        eng.wait(0)
    eng.set_frames_in_flight(1)
    assert eng.search(scenes[2], 4, 4).tobytes() == want[2].tobytes()
    eng.close()


def test_sharded_engine_slot_keeps_sweep_history_and_survives_a_failed_first_frame(amd):
    """fdcm_sharded_submit reserves the slot's buffers before every frame; the reservation must not wipe the L2
    sweep's per-chunk cost history (one launch order from the host proxy, then the previous build's), and a slot whose
    first frame cannot be built serves the next (tools/sharded_history.py, FDCM_SWEEP_ORDER=1 in a process of its own)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "sharded_history.py")], capture_output=True, text=True,
                         timeout=600, env={**os.environ, "FDCM_SWEEP_ORDER": "1"})
    assert out.returncode == 0 and "sharded history ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_sharded_engine_leaves_the_callers_device_alone(amd):
    """fdcm_sharded_* switch devices on the caller's thread for allocations and the exchange; the library's
    thread-local device and HIP's current device are what they were when a call returns."""
    import ctypes as C
    from openfdcm_amd import synthetic, _capi
    from openfdcm_amd.engine import ShardedEngine
    hip = C.CDLL("libamdhip64.so")
    def current():
        a, b = C.c_int(-1), C.c_int(-1)
        _capi.check(_capi.lib().fdcm_get_device(C.byref(a)))
        assert hip.hipGetDevice(C.byref(b)) == 0
        return a.value, b.value
    n = C.c_int()
    _capi.check(_capi.lib().fdcm_device_count(C.byref(n)))
    last = n.value - 1
    _capi.check(_capi.lib().fdcm_set_device(0))
    before = current()
    tmpls = synthetic.templates(6, 7, 128, 3)
    eng = ShardedEngine(tmpls, devices=[last], depth=8, coeff=5.0, padding=1.0, always_collective=True)
    assert current() == before
    scene = synthetic.scene(128, 20, 5)
    eng.search(scene, 3, 3)
    assert current() == before
    eng.search_topk(scene, 3, 3, 5, penalty=0)
    assert current() == before
    # the asynchronous entry points too (submit sizes the slot's buffers on the shard's device on the caller's thread)
    t = eng.submit(scene, 3, 3)
    assert current() == before
    eng.wait(t)
    assert current() == before
    t = eng.submit(scene, 3, 3, k=4, penalty=0)
    assert current() == before
    eng.wait(t)
    assert current() == before
    eng.close()
    assert current() == before


def test_sharded_engine_several_shards_on_one_device(amd):
    """The multi-shard logic of fdcm_sharded_* on a one-GPU box (FDCM_SHARDED_ALLOW_SAME_DEVICE: devices = [0, 0, 0]):
    contiguous ranges, a worker per shard and frame slot, per-shard offsets into the gathered array, uneven shards and
    shards without templates (fewer templates than shards), frames in flight, the top-k merge across shards -- everything
    of the multi-device path except RCCL's transport between devices, which the one-device RCCL test covers."""
    from openfdcm_amd import synthetic, _capi
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, ShardedEngine, search_raw, topk
    S = 256
    scenes = [synthetic.scene(S, 50, 61), synthetic.scene(S, 35, 62), synthetic.scene(S, 44, 63)]
    for T, shards in ((17, 3), (2, 3), (1, 4), (9, 2)):
        tmpls = synthetic.templates(T, 12, S, 70 + T)
        eng = ShardedEngine(tmpls, devices=[0] * shards, depth=16, coeff=5.0, padding=1.0, distance=O.L2, allow_same_device=True)
        assert eng.info()["shard_begin"] == [T * i // shards for i in range(shards + 1)]
        eng.set_frames_in_flight(2)
        tset = DeviceTemplates(tmpls)
        fm = DeviceFeatureMap.build(scenes[0], depth=16, coeff=5.0, padding=1.0, distance=O.L2)
        want, want_top = [], []
        for sc in scenes:
            fm.rebuild(sc)
            want.append(np.array(search_raw(fm, tset, sc, 4, 4, _capi.BATCH_OPTIMIZE, 10), copy=True))
            want_top.append({k: np.array(topk(fm, tset, k, 1, 1.5), copy=True) for k in (1, 7, len(want[-1]) + 3)})
        pend = []
        for i, sc in enumerate(scenes * 2):
            if len(pend) == 2:
                t, j = pend.pop(0)
                assert eng.wait(t).tobytes() == want[j].tobytes(), (T, shards, t)
            pend.append((eng.submit(sc, 4, 4, _capi.BATCH_OPTIMIZE, 10), i % len(scenes)))
        for t, j in pend:
            assert eng.wait(t).tobytes() == want[j].tobytes(), (T, shards, t)
        for j, sc in enumerate(scenes):
            for k, w in want_top[j].items():
                assert eng.search_topk(sc, 4, 4, k, penalty=1, tau=1.5).tobytes() == w.tobytes(), (T, shards, j, k)
        assert eng.info()["collectives"] == 0
        eng.close()
    with pytest.raises(_capi.FdcmError, match="twice"):
        ShardedEngine(synthetic.templates(4, 5, 128, 3), devices=[0, 0])


def test_sharded_engine_frame_shards(amd):
    """fdcm_sharded_set_mode(FDCM_SHARD_FRAMES): ticket t runs whole on shard t % N (here two and three shards on the one
    device, through the ALLOW_SAME_DEVICE hook) over the WHOLE template list; every frame -- full list and top-k -- equals the
    single-device call on its own scene, N x frames-in-flight tickets may be outstanding, nothing is exchanged, and the
    engine switches back to template shards."""
    from openfdcm_amd import synthetic, _capi
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, ShardedEngine, search_raw, topk
    S = 256
    tmpls = synthetic.templates(23, 12, S, 91)
    scenes = [synthetic.scene(S, 30 + 6 * i, 80 + i) for i in range(5)] + [np.zeros((4, 0), np.float32)]
    tset = DeviceTemplates(tmpls)
    fm = DeviceFeatureMap.build(scenes[0], depth=16, coeff=5.0, padding=1.0, distance=O.L2)
    want, want_top = [], []
    for sc in scenes:
        fm.rebuild(sc)
        want.append(np.array(search_raw(fm, tset, sc, 4, 4, _capi.BATCH_OPTIMIZE, 10), copy=True))
        want_top.append(np.array(topk(fm, tset, 9, 1, 1.5), copy=True) if sc.shape[1] else want[-1][:0])
    orc = O.build(scenes[1], depth=16, coeff=5.0, padding=1.0, distance=O.L2, nthreads=4)
    assert want[1].tobytes() == np.asarray(O.search(orc, tmpls, scenes[1], 4, 4, kind=O.BATCH_OPTIMIZE, batch=10, nthreads=4)).tobytes()
    for shards, in_flight in ((2, 1), (3, 2)):
        eng = ShardedEngine(tmpls, devices=[0] * shards, depth=16, coeff=5.0, padding=1.0, distance=O.L2, allow_same_device=True)
        assert eng.search(scenes[0], 4, 4).tobytes() == want[0].tobytes()      # template shards first
        eng.set_frames_in_flight(in_flight)
        eng.set_mode(_capi.SHARD_FRAMES)
        cap = shards * in_flight
        pend = []
        for step, i in enumerate(list(range(len(scenes))) * 3):
            if len(pend) == cap:
                t, j, top = pend.pop(0)
                assert eng.wait(t).tobytes() == (want_top[j] if top else want[j]).tobytes(), (shards, t, j, top)
            top = step % 4 == 3
            t = eng.submit(scenes[i], 4, 4, _capi.BATCH_OPTIMIZE, 10, k=9 if top else None, penalty=1, tau=1.5)
            assert t == step
            pend.append((t, i, top))
        with pytest.raises(_capi.FdcmError, match="not been waited for"):
            eng.submit(scenes[0], 4, 4)
        with pytest.raises(_capi.FdcmError, match="in flight"):
            eng.set_mode(_capi.SHARD_TEMPLATES)
        for t, j, top in reversed(pend):                                          # any order
            assert eng.wait(t).tobytes() == (want_top[j] if top else want[j]).tobytes(), (shards, t, j, top)
        with pytest.raises(_capi.FdcmError, match="ticket"):
            eng.wait(0)
        assert eng.search(scenes[2], 4, 4).tobytes() == want[2].tobytes()
        assert eng.search_topk(scenes[3], 4, 4, 9, penalty=1, tau=1.5).tobytes() == want_top[3].tobytes()
        info = eng.info()
        assert info["collectives"] == 0 and info["bytes_moved"] == 0
        eng.set_mode(_capi.SHARD_TEMPLATES)
        assert eng.search(scenes[4], 4, 4).tobytes() == want[4].tobytes()
        with pytest.raises(_capi.FdcmError, match="mode"):
            eng.set_mode(7)
        eng.close()


def test_sharded_engine_several_devices(amd):
    """fdcm_sharded_* with more than one device -- ncclCommInitAll over several devices, workers per device,
    cross-device send/recv into the gathered array at per-shard offsets, the top-k merge across shards, uneven and empty
    shards (fewer templates than devices).  Needs >= 2 GPUs: skipped on the one-GPU test box (the feature stays marked
    UNVERIFIED on multi-GPU hardware in README/INTEGRATION until this has run on such a node)."""
    import ctypes as C
    from openfdcm_amd import synthetic, _capi
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, ShardedEngine, search_raw, topk
    n = C.c_int()
    _capi.check(_capi.lib().fdcm_device_count(C.byref(n)))
    if n.value < 2:
        pytest.skip("needs at least two GPUs")
    nd = min(n.value, 8)
    S = 256
    scenes = [synthetic.scene(S, 50, 61), synthetic.scene(S, 35, 62)]
    for T in (nd * 5 + 3, nd - 1, 1):  # uneven shards; fewer templates than devices (empty shards)
        tmpls = synthetic.templates(T, 12, S, 70 + T)
        eng = ShardedEngine(tmpls, n_devices=nd, depth=16, coeff=5.0, padding=1.0, distance=O.L2)
        eng.set_frames_in_flight(2)
        _capi.check(_capi.lib().fdcm_set_device(0))
        tset = DeviceTemplates(tmpls)
        fm = DeviceFeatureMap.build(scenes[0], depth=16, coeff=5.0, padding=1.0, distance=O.L2)
        tickets = [(eng.submit(sc, 4, 4, _capi.BATCH_OPTIMIZE, 10), sc, None) for sc in scenes]
        for t, sc, _ in tickets:
            fm.rebuild(sc)
            want = search_raw(fm, tset, sc, 4, 4, _capi.BATCH_OPTIMIZE, 10)
            assert eng.wait(t).tobytes() == want.tobytes(), (T, t)
            for k in (1, 7, len(want) + 3):
                assert eng.search_topk(sc, 4, 4, k, penalty=1, tau=1.5).tobytes() == topk(fm, tset, k, 1, 1.5).tobytes(), (T, k)
        info = eng.info()
        assert info["shard_begin"][0] == 0 and info["shard_begin"][-1] == T and info["collectives"] > 0
        # frame shards over the real devices: ticket t whole on device t % nd, nothing exchanged
        eng.set_mode(_capi.SHARD_FRAMES)
        moved = eng.info()["bytes_moved"]
        tickets = [(eng.submit(scenes[i % 2], 4, 4, _capi.BATCH_OPTIMIZE, 10), scenes[i % 2]) for i in range(2 * nd)]
        for t, sc in tickets:
            fm.rebuild(sc)
            assert eng.wait(t).tobytes() == search_raw(fm, tset, sc, 4, 4, _capi.BATCH_OPTIMIZE, 10).tobytes(), (T, t)
        assert eng.info()["bytes_moved"] == moved
        eng.close()


def test_sharded_engine_rejects_bad_devices(amd):
    from openfdcm_amd import synthetic, _capi
    from openfdcm_amd.engine import ShardedEngine
    tmpls = synthetic.templates(4, 5, 128, 3)
    with pytest.raises(_capi.FdcmError, match="does not exist"):
        ShardedEngine(tmpls, devices=[0, 63])
    with pytest.raises(_capi.FdcmError, match="twice"):
        ShardedEngine(tmpls, devices=[0, 0])


def test_bench_force_dist_runs_the_rccl_gather_and_passes_its_parity_gate(amd):
    """bench.py --force-dist: the multi-process path (torch.distributed backend nccl = RCCL, per-frame gather with the
    count in a trailing record, slot buffers guarded by events) with world size 1, frames in flight, checked by the
    bench's own parity gate against the oracle."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--steps", "8", "--warmup", "2",
                          "--templates", "120", "--cpu-sample", "40", "--cpu-reps", "1", "--single-frames", "3"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    doc = json.loads(out.stdout.strip().splitlines()[-1])
    assert doc["parity_gate"] == "ok" and doc["n_gpus"] == 1 and doc["config"]["matches_per_step"] > 1000
    assert doc["roofline"]["frac"] > 0 and doc["roofline_search"]["achieved"] > 0 and doc["cpu_baseline"]["cores"] >= 1
    assert doc["config"]["distinct_scenes"] == 4 and "all 8 timed frames" in doc["parity_gate_detail"]
    assert doc["roofline"]["same_scene_ms"] > 0 and doc["dt3_build_ms"] >= doc["dt3_build_kernels_ms"] > 0
    assert doc["api"]["parity_gate"] == "ok" and 0 < doc["api_frame_ms"] < 50
    # --scaling frames with one rank through the process group: the frame-sharded schedule, the summed match count and the
    # digest gate (with N ranks: every rank's own frames)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--scaling", "frames", "--steps", "8", "--warmup", "2",
                          "--templates", "120", "--cpu-sample", "40", "--cpu-reps", "1", "--single-frames", "3", "--api-frames", "0"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    doc = json.loads(out.stdout.strip().splitlines()[-1])
    assert doc["parity_gate"] == "ok" and doc["config"]["sharding"] == "frames" and doc["scaling"] == "weak"
    assert "every rank's own frames" in doc["parity_gate_detail"] and doc["scaling_bounds"]["frames"]["8"] == 8.0


def test_bench_two_gpus_gates_the_gathered_list(amd):
    """bench.py --gpus 2 (what the driver's scaling run launches): two ranks over RCCL, the gathered list of both shards
    against the oracle on rank 0 (parity_gate) and the CPU baseline on the N > 1 line.  Runs by itself wherever two GPUs
    are visible; the one-GPU test box skips it (the multi-GPU curve stays UNMEASURED until such a node runs this)."""
    import ctypes as C
    import json
    import os
    import subprocess
    import sys
    from openfdcm_amd import _capi
    n = C.c_int()
    _capi.check(_capi.lib().fdcm_device_count(C.byref(n)))
    if n.value < 2:
        pytest.skip("needs at least two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for scaling in ("weak", "strong"):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--scaling", scaling,
                              "--templates", "120", "--cpu-reps", "1", "--single-frames", "3"], capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
        doc = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
        assert doc["parity_gate"] == "ok" and doc["n_gpus"] == 2 and doc["scaling"] == scaling
        assert "the gathered list of all ranks" in doc["parity_gate_detail"] and doc["cpu_baseline"]["cores"] >= 1
        assert doc["config"]["templates_total"] == (240 if scaling == "weak" else 120)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--scaling", "frames",
                          "--templates", "120", "--cpu-reps", "1", "--single-frames", "3", "--api-frames", "0"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    doc = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert doc["parity_gate"] == "ok" and doc["n_gpus"] == 2 and doc["config"]["sharding"] == "frames"
    assert "all 16 timed frames" in doc["parity_gate_detail"] and doc["config"]["templates_total"] == 120


def test_envelope_quotient_is_the_division_for_every_operand_pair(amd, tmp_path):
    """The L2 sweep's envelope test divides with 4 instructions instead of the compiler's 11 (csrc/fdcm_quotient.h);
    tools/div_check.hip compares the two bit for bit on every operand pair the sweep can produce (2.75e12 pairs on this
    GPU's own v_rcp_f32), and its self-test makes sure the comparison can fail."""
    import json
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine")
    exe = str(tmp_path / "div_check")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                           os.path.join(root, "tools", "div_check.hip"), "-o", exe], timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    doc = json.loads(out.stdout.strip().splitlines()[-1])
    assert out.returncode == 0 and doc["mismatches"] == 0 and doc["pairs_checked"] > 2.7e12, out.stdout[-500:]
    out = subprocess.run([exe, "selftest"], capture_output=True, text=True, timeout=600)
    doc = json.loads(out.stdout.strip().splitlines()[-1])
    assert out.returncode == 0 and doc["mismatches"] > 0, out.stdout[-500:]


@pytest.mark.parametrize("size", [2060, 2081, 2112, 3000, 4096])
def test_sizes_around_the_descriptor_tiles(amd, size):
    """Feature sizes above 2048 rows take the 32-column descriptor tiles, whose workgroups each write half a word of the
    slice's seeded-column mask (the L2 sweep skips the other columns): widths that end in the lower half of a word, in the
    upper half, on a word, and the largest size of that kernel, whole volume against the oracle."""
    from openfdcm_amd.engine import DeviceFeatureMap
    rng = np.random.default_rng(size)
    p = rng.uniform(0, 1, size=(24, 4)).astype(np.float32)
    lines = (p * np.float32(size - 1)).astype(np.float32)
    lines[0] = [0, 0, 5, 3]
    lines[1] = [size - 1, size - 1, size - 7, size - 4]
    scene = np.ascontiguousarray(lines.T)
    for dist in (0, 1, 2):  # (L1 too: its word-by-word pass has a partial last word at the same sizes)
        dev = DeviceFeatureMap.build(scene, depth=2, coeff=5.0, padding=1.0, distance=dist)
        orc = O.build(scene, depth=2, coeff=5.0, padding=1.0, distance=dist, nthreads=8)
        assert dev.volume().shape[1] == size
        assert_volume_equal(dev, orc, f"size {size} distance {dist}")


@pytest.mark.parametrize("size,kind", [(2896, "far corners"), (2897, "far corners"), (2896, "dense"), (2895, "one line"), (2890, "empty slices")])
def test_sizes_at_the_exact_integer_bound(amd, size, kind):
    """The balanced L2 sweep runs where every value of the reference's pass is an exact integer in float (W^2 + H^2 <= 2^24,
    i.e. up to 2896 px); 2897 takes the literal kernel.  Scenes that push the values to the bound: seeds only in two
    opposite corners (squared distances up to 2895^2 in pass 1 and numerators close to 2^24 in pass 2), a dense scene,
    a single line (every other orientation slice has no seed at all: FLT_MAX everywhere), few lines at depth 9 (mostly
    seedless slices, slices with a handful of columns).  Whole volume, L2 and L2^2, against the oracle."""
    from openfdcm_amd.engine import DeviceFeatureMap
    rng = np.random.default_rng(size + len(kind))
    a, b = 0.0, float(size - 1)
    if kind == "far corners":
        lines, depth = [[a, a, 9, 4], [b, b, b - 6, b - 11], [a + 3, a + 20, a + 30, a + 2], [b - 40, b - 2, b - 3, b - 25]], 3
    elif kind == "dense":
        pts = rng.uniform(0, size - 1, size=(400, 4)).astype(np.float32)
        pts[0], pts[1] = [a, a, 5, 3], [b, b, b - 7, b - 4]
        lines, depth = pts.tolist(), 2
    elif kind == "one line":
        lines, depth = [[a, a, 700, 260], [b, b, b, b]], 4  # (the zero-length line only pins the size)
    else:
        lines, depth = [[a, a, 300, 5], [b, b, b - 100, b - 330], [1500, 20, 1510, 2800]], 9
    scene = np.ascontiguousarray(np.array(lines, dtype=np.float32).T)
    for dist in (0, 1):
        dev = DeviceFeatureMap.build(scene, depth=depth, coeff=5.0, padding=1.0, distance=dist)
        orc = O.build(scene, depth=depth, coeff=5.0, padding=1.0, distance=dist, nthreads=8)
        assert dev.volume().shape[1] == size
        assert_volume_equal(dev, orc, f"size {size} ({kind}) distance {dist}")


@pytest.mark.parametrize("size", [63, 65, 130, 5000])
def test_l1_single_pass_sizes(amd, size):
    """The L1 transform as one pass over the volume (per-word minima, carries over the words of a row, word by word): one
    word and a bit, less than a word, and a height above 4096 rows (the other descriptor kernel), whole volume."""
    from openfdcm_amd.engine import DeviceFeatureMap
    rng = np.random.default_rng(size)
    p = rng.uniform(0, 1, size=(12, 4)).astype(np.float32)
    lines = (p * np.float32(size - 1)).astype(np.float32)
    lines[0] = [0, 0, 5, 3]
    lines[1] = [size - 1, size - 1, size - 7, size - 4]
    scene = np.ascontiguousarray(lines.T)
    dev = DeviceFeatureMap.build(scene, depth=2, coeff=5.0, padding=1.0, distance=2)
    orc = O.build(scene, depth=2, coeff=5.0, padding=1.0, distance=2, nthreads=8)
    assert_volume_equal(dev, orc, f"L1 size {size}")
