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
from .matchlist import Match, MatchList, records_of  # noqa: F401
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

    _pool_key = None  # set by build_cpu_featuremap: where the device handle goes when this object is dropped

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

    def __del__(self):
        try:
            if self._pool_key is not None and self._fm._h:
                _featuremap_pool.give(self._pool_key, self._fm)
        except Exception:  # interpreter shutdown: the handle's own __del__ frees it
            pass

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


class _FeatureMapPool:
    """Device feature maps whose `Dt3Cpu` was dropped, kept for the next `build_cpu_featuremap` with the same
    parameters.  The reference's callers build a new feature map every frame (matching.cpp:116-130: a fresh Dt3Cpu of
    O(V) host memory each call); here a fresh handle is two volumes of HBM, workspaces and a stream -- allocating and
    freeing them costs more than the build -- so `fm = build_cpu_featuremap(scene, params)` in a loop alternates
    between two handles (the old `fm` is dropped after the new one exists) and each call is a rebuild.
    At most PER_KEY idle handles per parameter set and TOTAL in all; `clear_featuremap_pool()` frees them."""
    PER_KEY, TOTAL = 2, 4

    def __init__(self):
        self._idle = {}  # (depth, coeff, padding, distance, device) -> [DeviceFeatureMap]

    def take(self, key):
        lst = self._idle.get(key)
        return lst.pop() if lst else None

    def give(self, key, fm):
        lst = self._idle.setdefault(key, [])
        if len(lst) >= self.PER_KEY or sum(len(v) for v in self._idle.values()) >= self.TOTAL:
            fm.close()
        else:
            lst.append(fm)

    def clear(self):
        idle, self._idle = self._idle, {}
        for lst in idle.values():
            for fm in lst:
                fm.close()


_featuremap_pool = _FeatureMapPool()


def clear_featuremap_pool():
    """Free the idle device feature maps kept for build_cpu_featuremap (extension)."""
    _featuremap_pool.clear()


def build_cpu_featuremap(scene, params=None, pool=None):
    """matching.cpp:116-130.  The name is the reference's; the build runs on the GPU (queued: the call returns once the
    kernels are launched, whatever reads the feature map next is ordered behind them)."""
    import ctypes as C
    params = params if params is not None else Dt3CpuParameters()
    dev = C.c_int()
    _capi.check(_capi.lib().fdcm_get_device(C.byref(dev)))
    key = (int(params.depth), float(params.dt3_coeff), float(params.padding), int(params.distance), dev.value)
    fm = _featuremap_pool.take(key)
    if fm is None:
        fm = DeviceFeatureMap.build(scene, depth=key[0], coeff=key[1], padding=key[2], distance=key[3])
    else:
        try:
            fm.rebuild(scene)
        except Exception:
            fm.close()
            raise
    out = Dt3Cpu(None, _device=fm)
    out._pool_key = key
    return out


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


def _unwrap(x, wrapper):
    return x._impl if isinstance(x, wrapper) else x


class _TemplateCache:
    """The DeviceTemplates of the template lists `search` / `get_template_lengths` were last called with.  The
    reference's callers pass the same Python list frame after frame (matching.cpp:279-300 copies it into a std::vector
    on every call); here that would be a device upload per call.  A hit needs the same list object with the same
    line counts and the same bytes: the list is packed (one concatenate, ~0.3 ms for 1000 x 32 lines) and compared with
    what was uploaded, so a template edited in place is seen.  Four lists are remembered; `clear_template_cache()`
    drops them (each holds its list and a device copy alive)."""
    SLOTS = 4

    def __init__(self):
        self._entries = []  # most recent first: (list object, line counts, packed lines, DeviceTemplates)

    def get(self, templates):
        if isinstance(templates, DeviceTemplates):
            return templates
        if not isinstance(templates, (list, tuple)):
            templates = list(templates)
        if templates and all(type(t) is _np.ndarray and t.ndim == 2 and t.shape[0] == 4 for t in templates):
            counts = [t.shape[1] for t in templates]
            data = _np.concatenate(templates, axis=1)  # (4, sum N_i), the elements' own dtype
            packed = None
        else:
            packed = _capi.pack_templates(templates)
            counts, data = packed[1].tolist(), packed[0]
        for k, (obj, ecounts, edata, tset) in enumerate(self._entries):
            if (obj is templates and tset._h and ecounts == counts and edata.dtype == data.dtype
                    and edata.shape == data.shape and _np.array_equal(edata, data)):
                if k:
                    self._entries.insert(0, self._entries.pop(k))
                return tset
        self._entries = [e for e in self._entries if e[0] is not templates]
        if packed is None:
            offsets = _np.zeros(len(counts) + 1, dtype=_np.int64)
            _np.cumsum(counts, out=offsets[1:])
            packed = (_np.ascontiguousarray(data.T, dtype=_np.float32).reshape(-1, 4), offsets)
        tset = DeviceTemplates(templates, _packed=packed)
        self._entries.insert(0, (templates, counts, data, tset))
        del self._entries[self.SLOTS:]
        return tset

    def clear(self):
        self._entries.clear()


_template_cache = _TemplateCache()


def clear_template_cache():
    """Forget the template lists uploaded on behalf of search() / get_template_lengths() (extension)."""
    _template_cache.clear()


def search(matcher, searcher, optimizer, featuremap, templates, scene):
    """matching.cpp:279-289 -> search<DefaultMatch> (defaultmatch.cpp:32-89).  Returns the raw, unsorted matches in
    the reference's positional order as a MatchList (matchlist.py: a list[Match] whose elements are made on access)."""
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
    tset = _template_cache.get(templates)
    rec = search_raw(dt3._fm, tset, scene_for_search, searcher.get_max_tmpl_lines(), searcher.get_max_scene_lines(), kind,
                     batch)
    return MatchList(rec)


def get_template_lengths(templates):
    """core::getTemplateLengths (math.h:319-324)."""
    return _template_cache.get(templates).lengths().tolist()


def penalize(penalty, matches, templatelengths):
    """matching.cpp:291-297; returns a new list (a MatchList; `matches` may be one or any list of Match)."""
    import ctypes as C
    penalty = _unwrap(penalty, PenaltyStrategy)
    if isinstance(penalty, ExponentialPenalty):
        kind, tau = _capi.EXPONENTIAL_PENALTY, penalty.get_tau()
    elif isinstance(penalty, DefaultPenalty):
        kind, tau = _capi.DEFAULT_PENALTY, 1.0
    else:
        raise TypeError("penalty must be DefaultPenalty or ExponentialPenalty")
    rec = records_of(matches)
    lens = _np.ascontiguousarray(templatelengths, dtype=_np.float32)
    rc = _capi.lib().fdcm_penalize(kind, tau, C.c_void_p(rec.ctypes.data), len(rec), _capi.fptr(lens), len(lens))
    if rc == -1 and "templatelengths" in _capi.lib().fdcm_last_error().decode():
        raise IndexError(_capi.lib().fdcm_last_error().decode())  # std::out_of_range -> IndexError
    _capi.check(rc)
    return MatchList(rec)


def sort_matches(matches, max_num_candidates=None):
    """matching.cpp:302-307: ascending score (std::sort, unstable on ties; libstdc++'s order between equal scores).
    max_num_candidates (extension: the C++ overload sortMatches(matches, maxNumCandidates), matchstrategy.h:52-55, which the
    reference's Python module does not bind): std::partial_sort -- only that many best matches are put in order in front."""
    import ctypes as C
    rec = records_of(matches)
    if max_num_candidates is None:
        _capi.check(_capi.lib().fdcm_sort_matches(C.c_void_p(rec.ctypes.data), len(rec)))
    else:
        _capi.check(_capi.lib().fdcm_partial_sort_matches(C.c_void_p(rec.ctypes.data), len(rec), int(max_num_candidates)))
    return MatchList(rec)


from .lineio import read, write  # noqa: E402  (.lines/.scene/.tmpl files, serialization.h)
