"""openfdcm_amd -- MI355X-native engine for OpenFDCM's DT3 build + DefaultMatch search.

Mirrors the Python surface of the reference's pybind11 module for this path
(modules/python/src/{openfdcm,core,matching}.cpp): the same class,
function and keyword names, so `import openfdcm_amd as openfdcm` is a drop-in for
build_cpu_featuremap / search / penalize / sort_matches.  All compute runs in libfdcm_hip.so
(HIP kernels for gfx950) through the C ABI in include/fdcm.h; there is no CPU fallback.
"""
import enum as _enum

import numpy as _np

from . import _capi
from .engine import DeviceFeatureMap, DeviceTemplates, FramePipeline, ShardedEngine, search_raw, topk  # noqa: F401  (extensions)

__version__ = "0.10.0"  # API level of the reference this mirrors (openfdcm.cpp:43)


class distance(_enum.IntEnum):
    """core::Distance (imgproc.h:148; core.cpp:45-49)."""
    L2 = 0
    L2_SQUARED = 1
    L1 = 2


L2, L2_SQUARED, L1 = distance.L2, distance.L2_SQUARED, distance.L1


class ThreadPool:
    """Placeholder for BS::thread_pool (matching.cpp:86-101).  The GPU engine does not use host
    threads; the object only carries the count so reference call sites run unchanged."""

    def __init__(self, num_threads=None):
        import os
        self._n = int(num_threads) if num_threads else (os.cpu_count() or 1)

    def get_tasks_queued(self): return 0
    def get_tasks_running(self): return 0
    def get_tasks_total(self): return 0
    def get_thread_count(self): return self._n
    def get_thread_ids(self): return []
    def purge(self): return None
    def __repr__(self): return f"<ThreadPool: threads={self._n}, tasks queued=0, tasks running=0>"


class Dt3CpuParameters:
    """matching.cpp:51-60,103-114.  Keyword `dt3Coeff`, attribute `dt3_coeff`, as in the reference."""

    def __init__(self, depth=30, dt3Coeff=5.0, padding=2.2, distance=distance.L2):
        self.depth = int(depth)
        self.dt3_coeff = float(dt3Coeff)
        self.padding = float(padding)
        self.distance = globals()["distance"](int(distance))

    def __repr__(self):
        return f"<PyDt3CpuParameters: depth={self.depth}, dt3_coeff={self.dt3_coeff:f}, padding={self.padding:f}>"


class Dt3Cpu:
    """The DT3 feature map (dt3cpu.h:46-63), resident in HBM.

    Dt3Cpu(dt3map, scene_translation, feature_size) adopts caller slices like the reference's
    constructor (matching.cpp:73): dt3map is {angle: (H, W) array}.
    """

    def __init__(self, dt3map, scene_translation=(0.0, 0.0), feature_size=(0, 0), _device=None):
        if _device is not None:
            self._fm = _device
            return
        keys = sorted(float(_np.float32(k)) for k in dt3map)
        W, H = int(feature_size[0]), int(feature_size[1])
        vol = _np.zeros((len(keys), W, H), dtype=_np.float32)
        by_key = {float(_np.float32(k)): v for k, v in dt3map.items()}
        for i, k in enumerate(keys):
            img = _np.asarray(by_key[k], dtype=_np.float32)
            if img.shape != (H, W):
                raise ValueError(f"slice shape {img.shape} != (H, W) = {(H, W)}")
            vol[i] = img.T
        self._fm = DeviceFeatureMap.from_volume(_np.array(keys, dtype=_np.float32), vol, scene_translation)

    def get_scene_translation(self):
        return self._fm.scene_translation.copy()

    def get_feature_size(self):
        return _np.array([self._fm.width, self._fm.height], dtype=_np.uint64)

    def get_dt3_map(self):
        return {float(k): self._fm.slice(i) for i, k in enumerate(self._fm.keys)}

    def __repr__(self):
        t = self._fm.scene_translation
        return (f"<Dt3Cpu: scene translation=({t[0]:f}, {t[1]:f}), "
                f"feature size=({self._fm.width}, {self._fm.height})>")


class FeatureMap:
    """Type-erased feature map (featuremap.h:98-124).  Wraps without copying the volume.

    The reference's Python module binds only the constructor and __repr__ (matching.cpp:66-70); the three
    methods below are the C++ class's (featuremap.h:109-121), served by the HIP kernels of the seam
    (fdcm_featuremap_minmax_translation / fdcm_featuremap_evaluate)."""

    def __init__(self, dt3):
        if isinstance(dt3, FeatureMap):
            dt3 = dt3._dt3
        if not isinstance(dt3, Dt3Cpu):
            raise TypeError("FeatureMap expects a Dt3Cpu")
        self._dt3 = dt3

    def get_feature_size(self):
        return self._dt3.get_feature_size()

    def minmax_translation(self, tmpl, align_vec):
        return self._dt3._fm.minmax_translation(tmpl, align_vec)

    def evaluate(self, templates, translations):
        return self._dt3._fm.evaluate(templates, translations)

    def __repr__(self): return "<FeatureMap>"


def build_cpu_featuremap(scene, params=None, pool=None):
    """matching.cpp:116-130.  The name is the reference's; the build runs on the GPU."""
    params = params if params is not None else Dt3CpuParameters()
    fm = DeviceFeatureMap.build(scene, depth=params.depth, coeff=params.dt3_coeff, padding=params.padding,
                                distance=int(params.distance))
    return Dt3Cpu(None, _device=fm)


# ---------------------------------------------------------------- optimise strategies
def _pool_arg(pool, num_threads):
    if isinstance(pool, int) and num_threads is None:
        return ThreadPool(pool)
    if num_threads is not None:
        return ThreadPool(num_threads)
    return pool if pool is not None else ThreadPool()


class DefaultOptimize:
    def __init__(self, pool=None, num_threads=None):
        self._pool = _pool_arg(pool, num_threads)

    def get_pool(self): return self._pool
    def __repr__(self): return "<DefaultOptimize>"


class BatchOptimize:
    def __init__(self, batch_size, pool=None, num_threads=None):
        self._batch_size = int(batch_size)
        self._pool = _pool_arg(pool, num_threads)

    def get_batch_size(self): return self._batch_size
    def get_pool(self): return self._pool
    def __repr__(self): return "<BatchOptimize>"


class IndulgentOptimize:
    """optimizestrategies/indulgentoptimize.h:30-47; runs on the device like the other two optimisers."""

    def __init__(self, indulgent_number_of_passthroughs, pool=None, num_threads=None):
        self._n = int(indulgent_number_of_passthroughs)
        self._pool = _pool_arg(pool, num_threads)

    def get_number_of_passthroughs(self): return self._n
    def get_pool(self): return self._pool
    def __repr__(self): return f"<IndulgentOptimize: number_of_passthroughs={self._n}>"


class OptimizeStrategy:
    def __init__(self, impl):
        self._impl = impl._impl if isinstance(impl, OptimizeStrategy) else impl

    def __repr__(self): return "<OptimizeStrategy>"


# ---------------------------------------------------------------- penalties
class DefaultPenalty:
    def __repr__(self): return "<DefaultPenalty>"


class ExponentialPenalty:
    def __init__(self, tau):
        self._tau = float(_np.float32(tau))

    def get_tau(self): return self._tau
    def __repr__(self): return f"<ExponentialPenalty: tau={self._tau:f}>"


class PenaltyStrategy:
    def __init__(self, impl):
        self._impl = impl._impl if isinstance(impl, PenaltyStrategy) else impl

    def __repr__(self): return "<PenaltyStrategy>"


# ---------------------------------------------------------------- search strategies
class DefaultSearch:
    def __init__(self, max_tmpl_lines, max_scene_lines):
        self._t, self._s = int(max_tmpl_lines), int(max_scene_lines)

    def get_max_tmpl_lines(self): return self._t
    def get_max_scene_lines(self): return self._s

    def __repr__(self):
        return f"<DefaultSearch: max tmpl lines={self._t}, max scene lines={self._s}>"


class ConcentricRangeStrategy:
    """DefaultSearch restricted to the scene lines whose centre lies in an annulus
    (searchstrategies/concentricrange.h:36-84, concentricrange.cpp:29-60)."""

    def __init__(self, max_tmpl_lines, max_scene_lines, center_position, low_boundary, high_boundary):
        self._t, self._s = int(max_tmpl_lines), int(max_scene_lines)
        self._c = _np.asarray(center_position, dtype=_np.float32)
        self._lo, self._hi = float(low_boundary), float(high_boundary)

    def get_max_tmpl_lines(self): return self._t
    def get_max_scene_lines(self): return self._s
    def get_center_position(self): return self._c
    def get_low_radius_boundary(self): return self._lo
    def get_high_radius_boundary(self): return self._hi


class SearchStrategy:
    def __init__(self, impl):
        self._impl = impl._impl if isinstance(impl, SearchStrategy) else impl

    def __repr__(self): return "<SearchStrategy>"


class DefaultMatch:
    def __repr__(self): return "<DefaultMatch>"


class MatchStrategy:
    def __init__(self, impl):
        self._impl = impl._impl if isinstance(impl, MatchStrategy) else impl

    def __repr__(self): return "<MatchStrategy>"


class Match:
    """matchstrategy.h:35-44; transform is a (2, 3) float32 array."""

    def __init__(self, tmpl_idx, score, transform):
        self.tmpl_idx = int(tmpl_idx)
        self.score = float(score)
        self.transform = _np.asarray(transform, dtype=_np.float32).reshape(2, 3)

    def __repr__(self):
        return f"<Match tmplIdx={self.tmpl_idx}, score={self.score:g}, transform=\n{self.transform}>"


def _unwrap(x, wrapper):
    return x._impl if isinstance(x, wrapper) else x


def _matches_to_records(matches):
    rec = _np.zeros(len(matches), dtype=_capi.MATCH_DTYPE)
    for i, m in enumerate(matches):
        rec[i] = (m.tmpl_idx, m.score, _np.asarray(m.transform, dtype=_np.float32).reshape(6))
    return rec


def _records_to_matches(rec):
    return [Match(int(r["tmpl_idx"]), float(r["score"]), r["transform"].reshape(2, 3).copy()) for r in rec]


def search(matcher, searcher, optimizer, featuremap, templates, scene):
    """matching.cpp:279-289 -> search<DefaultMatch> (defaultmatch.cpp:32-89).  Returns the raw,
    unsorted list[Match] in the reference's positional order."""
    matcher = _unwrap(matcher, MatchStrategy)
    searcher = _unwrap(searcher, SearchStrategy)
    optimizer = _unwrap(optimizer, OptimizeStrategy)
    if not isinstance(matcher, DefaultMatch):
        raise TypeError("matcher must be a DefaultMatch")
    scene_for_search = scene
    if isinstance(searcher, ConcentricRangeStrategy):
        # The strategy is DefaultSearch over the filtered scene lines, and search<DefaultMatch> only
        # uses the geometry of the scene line of each combination (defaultmatch.cpp:57-61).
        import ctypes as C
        rec = _capi.as_records(scene)
        if rec.shape[0]:
            idx = _np.zeros(rec.shape[0], dtype=_np.int64)
            n = C.c_int64()
            ctr = _np.ascontiguousarray(searcher.get_center_position(), dtype=_np.float32)
            _capi.check(_capi.lib().fdcm_filter_in_range(_capi.fptr(rec), rec.shape[0], _capi.fptr(ctr),
                                                         searcher.get_low_radius_boundary(),
                                                         searcher.get_high_radius_boundary(),
                                                         idx.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(n)))
            scene_for_search = _np.ascontiguousarray(rec[idx[:n.value]].T)
    elif not isinstance(searcher, DefaultSearch):
        raise NotImplementedError("only DefaultSearch and ConcentricRangeStrategy run on the GPU path")
    if isinstance(optimizer, BatchOptimize):
        kind, batch = _capi.BATCH_OPTIMIZE, optimizer.get_batch_size()
    elif isinstance(optimizer, DefaultOptimize):
        kind, batch = _capi.DEFAULT_OPTIMIZE, 1
    elif isinstance(optimizer, IndulgentOptimize):
        kind, batch = _capi.INDULGENT_OPTIMIZE, max(1, optimizer.get_number_of_passthroughs())
    else:
        raise TypeError("optimizer must be DefaultOptimize, BatchOptimize or IndulgentOptimize")
    dt3 = featuremap._dt3 if isinstance(featuremap, FeatureMap) else featuremap
    if not isinstance(dt3, Dt3Cpu):
        raise TypeError("featuremap must be a Dt3Cpu or FeatureMap")
    tset = templates if isinstance(templates, DeviceTemplates) else DeviceTemplates(list(templates))
    rec = search_raw(dt3._fm, tset, scene_for_search, searcher.get_max_tmpl_lines(), searcher.get_max_scene_lines(), kind,
                     batch)
    return _records_to_matches(rec)


def get_template_lengths(templates):
    """core::getTemplateLengths (math.h:319-324)."""
    tset = templates if isinstance(templates, DeviceTemplates) else DeviceTemplates(list(templates))
    return [float(v) for v in tset.lengths()]


def penalize(penalty, matches, templatelengths):
    """matching.cpp:291-297; returns a new list."""
    import ctypes as C
    penalty = _unwrap(penalty, PenaltyStrategy)
    if isinstance(penalty, ExponentialPenalty):
        kind, tau = _capi.EXPONENTIAL_PENALTY, penalty.get_tau()
    elif isinstance(penalty, DefaultPenalty):
        kind, tau = _capi.DEFAULT_PENALTY, 1.0
    else:
        raise TypeError("penalty must be DefaultPenalty or ExponentialPenalty")
    rec = _matches_to_records(matches)
    lens = _np.ascontiguousarray(templatelengths, dtype=_np.float32)
    rc = _capi.lib().fdcm_penalize(kind, tau, C.c_void_p(rec.ctypes.data), len(rec), _capi.fptr(lens), len(lens))
    if rc == -1 and "templatelengths" in _capi.lib().fdcm_last_error().decode():
        raise IndexError(_capi.lib().fdcm_last_error().decode())  # std::out_of_range -> IndexError
    _capi.check(rc)
    return _records_to_matches(rec)


def sort_matches(matches):
    """matching.cpp:302-307: ascending score (std::sort, unstable on ties)."""
    import ctypes as C
    rec = _matches_to_records(matches)
    _capi.check(_capi.lib().fdcm_sort_matches(C.c_void_p(rec.ctypes.data), len(rec)))
    return _records_to_matches(rec)


from .lineio import read, write  # noqa: E402  (.lines/.scene/.tmpl files, serialization.h)
