"""ctypes loader for the CPU oracle (oracle/fdcm_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (openfdcm_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfdcm_oracle.so")

L2, L2_SQUARED, L1 = 0, 1, 2
DEFAULT_OPTIMIZE, BATCH_OPTIMIZE, INDULGENT_OPTIMIZE = 0, 1, 2  # for INDULGENT `batch` is the number of passthroughs


def build_library(force=False):
    src = os.path.join(_HERE, "fdcm_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libfdcm_oracle.so"])
    return _LIB_PATH


class _Match(C.Structure):
    _fields_ = [("tmpl_idx", C.c_int), ("score", C.c_float), ("transform", C.c_float * 6)]


MATCH_DTYPE = np.dtype([("tmpl_idx", "<i4"), ("score", "<f4"), ("transform", "<f4", (6,))])

_lib = None


def lib():
    global _lib
    if _lib is None:
        build_library()
        _lib = C.CDLL(_LIB_PATH)
        fp, lp, vp = C.POINTER(C.c_float), C.POINTER(C.c_long), C.c_void_p
        _lib.fdcmo_build.restype = vp
        _lib.fdcmo_build.argtypes = [fp, C.c_long, C.c_long, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]
        _lib.fdcmo_free.argtypes = [vp]
        _lib.fdcmo_info.argtypes = [vp, lp, lp, fp, fp, lp]
        _lib.fdcmo_keys.argtypes = [vp, fp]
        _lib.fdcmo_slice.argtypes = [vp, C.c_long, fp]
        _lib.fdcmo_from_slices.restype = vp
        _lib.fdcmo_from_slices.argtypes = [fp, C.c_long, fp, C.c_long, C.c_long, C.c_float, C.c_float]
        _lib.fdcmo_search.restype = C.c_long
        _lib.fdcmo_search.argtypes = [vp, fp, lp, C.c_long, fp, C.c_long, C.c_long, C.c_long, C.c_int, C.c_long,
                                      C.c_int, C.POINTER(C.POINTER(_Match)), lp]
        _lib.fdcmo_free_matches.argtypes = [C.POINTER(_Match)]
        _lib.fdcmo_search_concentric.restype = C.c_long
        _lib.fdcmo_search_concentric.argtypes = [vp, fp, lp, C.c_long, fp, C.c_long, C.c_long, C.c_long, C.c_float, C.c_float,
                                                 C.c_float, C.c_float, C.c_int, C.c_long, C.c_int,
                                                 C.POINTER(C.POINTER(_Match))]
        _lib.fdcmo_filter_in_range.restype = C.c_long
        _lib.fdcmo_filter_in_range.argtypes = [fp, C.c_long, C.c_float, C.c_float, C.c_float, C.c_float, lp]
        _lib.fdcmo_concentric_search.restype = C.c_long
        _lib.fdcmo_concentric_search.argtypes = [fp, C.c_long, fp, C.c_long, C.c_long, C.c_long, C.c_float, C.c_float,
                                                 C.c_float, C.c_float, lp]
        _lib.fdcmo_rasterize_vector.argtypes = [C.c_float, C.c_float, fp]
        _lib.fdcmo_rasterize_line.restype = C.c_long
        _lib.fdcmo_rasterize_line.argtypes = [fp, lp, lp, C.c_long]
        _lib.fdcmo_clip_lines.restype = C.c_long
        _lib.fdcmo_clip_lines.argtypes = [fp, C.c_long, C.c_float, C.c_float, C.c_float, C.c_float, fp]
        _lib.fdcmo_draw_lines.argtypes = [fp, C.c_long, C.c_long, fp, C.c_long, C.c_float]
        _lib.fdcmo_distance_transform.argtypes = [fp, C.c_long, C.c_long, C.c_long, C.c_int, fp]
        _lib.fdcmo_column_pass_l2.argtypes = [fp, C.c_long, C.c_long]
        _lib.fdcmo_line_integral.argtypes = [fp, C.c_long, C.c_long, C.c_float]
        _lib.fdcmo_scene_centered_translation.argtypes = [fp, C.c_long, C.c_float, fp, lp]
        _lib.fdcmo_minmax_translation.argtypes = [fp, C.c_long, C.c_float, C.c_float, C.c_long, C.c_long,
                                                  C.c_float, C.c_float, fp]
        _lib.fdcmo_closest_orientation.restype = C.c_long
        _lib.fdcmo_closest_orientation.argtypes = [fp, C.c_long, fp]
        _lib.fdcmo_propagate.argtypes = [fp, C.c_long, fp, C.c_long, C.c_long, C.c_float]
        _lib.fdcmo_default_search.restype = C.c_long
        _lib.fdcmo_default_search.argtypes = [fp, C.c_long, fp, C.c_long, C.c_long, C.c_long, lp]
        _lib.fdcmo_centered_range.argtypes = [C.c_long, C.c_long, C.c_long, lp]
        _lib.fdcmo_align.argtypes = [fp, fp, fp]
        _lib.fdcmo_transform.argtypes = [fp, C.c_long, fp, fp]
        _lib.fdcmo_optimize.restype = C.c_int
        _lib.fdcmo_optimize.argtypes = [vp, fp, C.c_long, C.c_float, C.c_float, C.c_int, C.c_long, fp]
        _lib.fdcmo_evaluate.argtypes = [vp, fp, C.c_long, fp, C.c_long, fp]
        _lib.fdcmo_eigen_sum.restype = C.c_float
        _lib.fdcmo_eigen_sum.argtypes = [fp, C.c_long]
        _lib.fdcmo_atanf.restype = C.c_float
        _lib.fdcmo_atanf.argtypes = [C.c_float]
        _lib.fdcmo_sort_matches.argtypes = [vp, C.c_long]
        _lib.fdcmo_partial_sort_matches.argtypes = [vp, C.c_long, C.c_long]
        _lib.fdcmo_penalize.restype = C.c_int
        _lib.fdcmo_penalize.argtypes = [C.c_int, C.c_float, vp, C.c_long, fp, C.c_long]
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_long))


def as_lines(a):
    """(4, N) array-like -> contiguous float32 array of N records x1,y1,x2,y2 (= column-major 4xN)."""
    a = np.asarray(a, dtype=np.float32)
    if a.ndim == 1:
        a = a.reshape(4, -1)
    assert a.shape[0] == 4, a.shape
    return np.ascontiguousarray(a.T)


def from_lines(rec):
    return np.ascontiguousarray(rec.T)


class FeatureMap:
    def __init__(self, handle):
        self._h = handle
        W, H, d = C.c_long(), C.c_long(), C.c_long()
        tx, ty = C.c_float(), C.c_float()
        lib().fdcmo_info(handle, C.byref(W), C.byref(H), C.byref(tx), C.byref(ty), C.byref(d))
        self.W, self.H, self.depth = W.value, H.value, d.value
        self.translation = np.array([tx.value, ty.value], dtype=np.float32)
        self.keys = np.zeros(self.depth, dtype=np.float32)
        if self.depth:
            lib().fdcmo_keys(handle, _fp(self.keys))

    def slice(self, k):
        """Slice k as an (H, W) float32 array (Fortran order in memory, like the reference)."""
        out = np.zeros((self.W, self.H), dtype=np.float32)
        lib().fdcmo_slice(self._h, k, _fp(out))
        return out.T

    def volume(self):
        """(depth, W, H) float32: [k][x][y], y fastest."""
        out = np.zeros((self.depth, self.W, self.H), dtype=np.float32)
        for k in range(self.depth):
            lib().fdcmo_slice(self._h, k, _fp(out[k]))
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fdcmo_free(self._h)
            self._h = None


def build(scene, depth=30, coeff=5.0, padding=2.2, distance=L2, nthreads=1, stop_after=3):
    s = as_lines(scene)
    h = lib().fdcmo_build(_fp(s), s.shape[0], depth, coeff, padding, distance, nthreads, stop_after)
    return FeatureMap(h)


def from_volume(keys, vol, translation):
    """vol: (depth, W, H) float32 [k][x][y]."""
    keys = np.ascontiguousarray(keys, dtype=np.float32)
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    m, W, H = vol.shape
    h = lib().fdcmo_from_slices(_fp(keys), m, _fp(vol), W, H, float(translation[0]), float(translation[1]))
    return FeatureMap(h)


def pack_templates(templates):
    recs = [as_lines(t) for t in templates]
    offsets = np.zeros(len(recs) + 1, dtype=np.int64)
    for i, r in enumerate(recs):
        offsets[i + 1] = offsets[i] + r.shape[0]
    flat = np.concatenate(recs, axis=0) if recs and offsets[-1] > 0 else np.zeros((0, 4), dtype=np.float32)
    return np.ascontiguousarray(flat, dtype=np.float32), offsets


def search(fm, templates, scene, max_tmpl_lines, max_scene_lines, kind=BATCH_OPTIMIZE, batch=10, nthreads=1,
           return_stats=False):
    flat, offsets = pack_templates(templates)
    s = as_lines(scene)
    out = C.POINTER(_Match)()
    stats = np.zeros(2, dtype=np.int64)
    n = lib().fdcmo_search(fm._h, _fp(flat), _lp(offsets), len(templates), _fp(s), s.shape[0], max_tmpl_lines,
                           max_scene_lines, kind, batch, nthreads, C.byref(out), _lp(stats))
    res = np.zeros(n, dtype=MATCH_DTYPE)
    if n:
        C.memmove(res.ctypes.data, out, n * MATCH_DTYPE.itemsize)
    lib().fdcmo_free_matches(out)
    return (res, stats) if return_stats else res


def search_concentric(fm, templates, scene, max_tmpl_lines, max_scene_lines, center, low, high, kind=BATCH_OPTIMIZE,
                      batch=10, nthreads=1):
    flat, offsets = pack_templates(templates)
    s = as_lines(scene)
    out = C.POINTER(_Match)()
    n = lib().fdcmo_search_concentric(fm._h, _fp(flat), _lp(offsets), len(templates), _fp(s), s.shape[0], max_tmpl_lines,
                                      max_scene_lines, center[0], center[1], low, high, kind, batch, nthreads,
                                      C.byref(out))
    res = np.zeros(n, dtype=MATCH_DTYPE)
    if n:
        C.memmove(res.ctypes.data, out, n * MATCH_DTYPE.itemsize)
    lib().fdcmo_free_matches(out)
    return res


def filter_in_range(lines, center, low, high):
    s = as_lines(lines)
    out = np.zeros(max(1, s.shape[0]), dtype=np.int64)
    n = lib().fdcmo_filter_in_range(_fp(s), s.shape[0], center[0], center[1], low, high, _lp(out))
    return out[:n].copy()


def concentric_search(tmpl, scene, max_tmpl_lines, max_scene_lines, center, low, high):
    t, s = as_lines(tmpl), as_lines(scene)
    out = np.zeros(2 * max(1, max_tmpl_lines * max_scene_lines), dtype=np.int64)
    n = lib().fdcmo_concentric_search(_fp(t), t.shape[0], _fp(s), s.shape[0], max_tmpl_lines, max_scene_lines,
                                      center[0], center[1], low, high, _lp(out))
    return out[: 2 * n].reshape(n, 2)


# ---- unit-level helpers used by the known-answer tests ----
def rasterize_vector(x, y):
    out = np.zeros(2, dtype=np.float32)
    lib().fdcmo_rasterize_vector(x, y, _fp(out))
    return out


def rasterize_line(line):
    l = np.ascontiguousarray(line, dtype=np.float32)
    xs = np.zeros(1 << 16, dtype=np.int64)
    ys = np.zeros(1 << 16, dtype=np.int64)
    n = lib().fdcmo_rasterize_line(_fp(l), _lp(xs), _lp(ys), xs.size)
    return np.stack([xs[:n], ys[:n]])


def clip_lines(lines, xmin, xmax, ymin, ymax):
    s = as_lines(lines)
    out = np.zeros_like(s)
    n = lib().fdcmo_clip_lines(_fp(s), s.shape[0], xmin, xmax, ymin, ymax, _fp(out))
    return from_lines(out[:n])


def draw_lines(img, lines, color):
    """img: (H, W) array; returns a new (H, W) float32 array."""
    a = np.array(np.asarray(img, dtype=np.float32).T, dtype=np.float32, order="C", copy=True)  # [x][y]
    s = as_lines(lines)
    lib().fdcmo_draw_lines(_fp(a), a.shape[1], a.shape[0], _fp(s), s.shape[0], color)
    return a.T


def distance_transform(lines, W, H, distance):
    s = as_lines(lines)
    out = np.zeros((W, H), dtype=np.float32)
    lib().fdcmo_distance_transform(_fp(s), s.shape[0], W, H, distance, _fp(out))
    return out.T


def column_pass_l2(img):
    a = np.array(np.asarray(img, dtype=np.float32).T, dtype=np.float32, order="C", copy=True)
    lib().fdcmo_column_pass_l2(_fp(a), a.shape[1], a.shape[0])
    return a.T


def line_integral(img, angle):
    a = np.array(np.asarray(img, dtype=np.float32).T, dtype=np.float32, order="C", copy=True)
    lib().fdcmo_line_integral(_fp(a), a.shape[1], a.shape[0], np.float32(angle))
    return a.T


def scene_centered_translation(lines, padding):
    s = as_lines(lines)
    t = np.zeros(2, dtype=np.float32)
    size = np.zeros(2, dtype=np.int64)
    lib().fdcmo_scene_centered_translation(_fp(s), s.shape[0], padding, _fp(t), _lp(size))
    return t, size


def minmax_translation(tmpl, align_vec, size, extra=(0.0, 0.0)):
    s = as_lines(tmpl) if np.size(tmpl) else np.zeros((0, 4), dtype=np.float32)
    out = np.zeros(2, dtype=np.float32)
    lib().fdcmo_minmax_translation(_fp(s), s.shape[0], align_vec[0], align_vec[1], size[0], size[1], extra[0],
                                   extra[1], _fp(out))
    return out


def closest_orientation(keys, line):
    k = np.ascontiguousarray(keys, dtype=np.float32)
    l = np.ascontiguousarray(line, dtype=np.float32)
    return lib().fdcmo_closest_orientation(_fp(k), k.size, _fp(l))


def propagate(keys, vol, coeff):
    """vol (m, W, H) [k][x][y]; returns the propagated copy."""
    k = np.ascontiguousarray(keys, dtype=np.float32)
    v = np.array(vol, dtype=np.float32, order="C", copy=True)
    lib().fdcmo_propagate(_fp(k), k.size, _fp(v), v.shape[1], v.shape[2], coeff)
    return v


def default_search(tmpl, scene, max_tmpl_lines, max_scene_lines):
    t, s = as_lines(tmpl), as_lines(scene)
    out = np.zeros(2 * max(1, max_tmpl_lines * max_scene_lines), dtype=np.int64)
    n = lib().fdcmo_default_search(_fp(t), t.shape[0], _fp(s), s.shape[0], max_tmpl_lines, max_scene_lines, _lp(out))
    return out[: 2 * n].reshape(n, 2)


def centered_range(c, n, maxlen):
    out = np.zeros(2, dtype=np.int64)
    lib().fdcmo_centered_range(c, n, maxlen, _lp(out))
    return tuple(out)


def align(tmpl_line, ref_line):
    t = np.ascontiguousarray(tmpl_line, dtype=np.float32)
    r = np.ascontiguousarray(ref_line, dtype=np.float32)
    out = np.zeros(12, dtype=np.float32)
    lib().fdcmo_align(_fp(t), _fp(r), _fp(out))
    return out[:6].reshape(2, 3), out[6:].reshape(2, 3)


def transform(lines, T):
    s = as_lines(lines)
    t = np.ascontiguousarray(T, dtype=np.float32).reshape(6)
    out = np.zeros_like(s)
    lib().fdcmo_transform(_fp(s), s.shape[0], _fp(t), _fp(out))
    return from_lines(out)


def optimize(fm, tmpl, align_vec, kind=BATCH_OPTIMIZE, batch=10):
    s = as_lines(tmpl)
    out = np.zeros(3, dtype=np.float32)
    ok = lib().fdcmo_optimize(fm._h, _fp(s), s.shape[0], align_vec[0], align_vec[1], kind, batch, _fp(out))
    return (float(out[0]), out[1:].copy()) if ok else None


def evaluate(fm, tmpl, translations):
    s = as_lines(tmpl)
    t = np.ascontiguousarray(translations, dtype=np.float32).reshape(-1, 2)
    out = np.zeros(t.shape[0], dtype=np.float32)
    lib().fdcmo_evaluate(fm._h, _fp(s), s.shape[0], _fp(t), t.shape[0], _fp(out))
    return out


def eigen_sum(v):
    v = np.ascontiguousarray(v, dtype=np.float32)
    return float(lib().fdcmo_eigen_sum(_fp(v), v.size))


# ---- unit-level entry points for the math.test.cpp known answers
def argsort_greater(v):
    v = np.ascontiguousarray(v, dtype=np.float32)
    out = np.zeros(v.size, dtype=np.int64)
    lib().fdcmo_argsort_greater(_fp(v), v.size, _lp(out))
    return out.tolist()


def binary_search_greater(sorted_desc, value):
    v = np.ascontiguousarray(sorted_desc, dtype=np.float32)
    f = lib().fdcmo_binary_search_greater
    f.restype = C.c_long
    f.argtypes = [C.POINTER(C.c_float), C.c_long, C.c_float]
    return int(f(_fp(v), v.size, float(value)))


def minmax_point(lines):
    s = as_lines(lines)
    out = np.zeros(4, dtype=np.float32)
    lib().fdcmo_minmax_point(_fp(s), s.shape[0], _fp(out))
    return out[:2].copy(), out[2:].copy()


def line_props(line):
    """(angle, length, normalised direction) of one line x1, y1, x2, y2."""
    l = np.ascontiguousarray(line, dtype=np.float32)
    out = np.zeros(4, dtype=np.float32)
    lib().fdcmo_line_props(_fp(l), _fp(out))
    return float(out[0]), float(out[1]), out[2:].copy()


def translate(lines, t):
    s = as_lines(lines)
    out = np.zeros_like(s)
    f = lib().fdcmo_translate
    f.argtypes = [C.POINTER(C.c_float), C.c_long, C.c_float, C.c_float, C.POINTER(C.c_float)]
    f(_fp(s), s.shape[0], float(t[0]), float(t[1]), _fp(out))
    return from_lines(out)


def combine(translation, T):
    t = np.ascontiguousarray(T, dtype=np.float32).reshape(6)
    out = np.zeros(6, dtype=np.float32)
    f = lib().fdcmo_combine
    f.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    f(float(translation[0]), float(translation[1]), _fp(t), _fp(out))
    return out.reshape(2, 3)


def sort_matches(records):
    """sortMatches (matchstrategy.h:46-50) on a copy of the MATCH_DTYPE records."""
    rec = np.array(records, dtype=MATCH_DTYPE, copy=True, order="C")
    lib().fdcmo_sort_matches(rec.ctypes.data, len(rec))
    return rec


def partial_sort_matches(records, k):
    """sortMatches(matches, maxNumCandidates) (matchstrategy.h:52-55) on a copy of the records."""
    rec = np.array(records, dtype=MATCH_DTYPE, copy=True, order="C")
    lib().fdcmo_partial_sort_matches(rec.ctypes.data, len(rec), int(k))
    return rec


def penalize(records, lengths, tau=None):
    """penalize<DefaultPenalty> (tau None) / penalize<ExponentialPenalty> on a copy of the records."""
    rec = np.array(records, dtype=MATCH_DTYPE, copy=True, order="C")
    lens = np.ascontiguousarray(lengths, dtype=np.float32)
    rc = lib().fdcmo_penalize(0 if tau is None else 1, 0.0 if tau is None else tau, rec.ctypes.data, len(rec), _fp(lens), len(lens))
    if rc:
        raise IndexError("templatelengths is not consistent with match template indices")
    return rec
