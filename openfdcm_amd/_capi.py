"""ctypes binding of libfdcm_hip.so (include/fdcm.h).

There is no CPU fallback: if the library is missing or a call fails, this raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FDCM_LIBRARY selects another build of the same library (the lab build of `make LAB=1`, for tools/)
LIB_PATH = os.environ.get("FDCM_LIBRARY") or os.path.join(_HERE, "libfdcm_hip.so")

FDCM_OK = 0
L2, L2_SQUARED, L1 = 0, 1, 2
DEFAULT_OPTIMIZE, BATCH_OPTIMIZE, INDULGENT_OPTIMIZE = 0, 1, 2
DEFAULT_PENALTY, EXPONENTIAL_PENALTY = 0, 1
SHARDED_ALWAYS_COLLECTIVE = 1
SHARDED_ALLOW_SAME_DEVICE = 2
SHARD_TEMPLATES, SHARD_FRAMES = 0, 1

MATCH_DTYPE = np.dtype([("tmpl_idx", "<i4"), ("score", "<f4"), ("transform", "<f4", (6,))])
assert MATCH_DTYPE.itemsize == 32


class FeaturemapInfo(C.Structure):
    _fields_ = [("width", C.c_int64), ("height", C.c_int64), ("depth", C.c_int64),
                ("scene_translation", C.c_float * 2), ("distance", C.c_int32),
                ("dt3_coeff", C.c_float), ("padding", C.c_float)]


class BuildTiming(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("seeds_ms", C.c_float), ("pass1_ms", C.c_float),
                ("pass2_ms", C.c_float), ("propagate_ms", C.c_float), ("integral_ms", C.c_float), ("span_ms", C.c_float)]


class SearchTiming(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("kernel_ms", C.c_float), ("candidates", C.c_int64),
                ("evaluations", C.c_int64)]


# every symbol include/fdcm.h declares: (name, restype, argtypes)
_fp, _i64p, _vp = C.POINTER(C.c_float), C.POINTER(C.c_int64), C.c_void_p
SYMBOLS = [
    ("fdcm_last_error", C.c_char_p, []),
    ("fdcm_version", C.c_char_p, []),
    ("fdcm_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("fdcm_set_device", C.c_int, [C.c_int]),
    ("fdcm_get_device", C.c_int, [C.POINTER(C.c_int)]),
    ("fdcm_featuremap_build", C.c_int, [_fp, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_int, C.POINTER(_vp)]),
    ("fdcm_featuremap_rebuild", C.c_int, [_vp, _fp, C.c_int64]),
    ("fdcm_featuremap_free", C.c_int, [_vp]),
    ("fdcm_featuremap_get_info", C.c_int, [_vp, C.POINTER(FeaturemapInfo)]),
    ("fdcm_featuremap_keys", C.c_int, [_vp, _fp]),
    ("fdcm_featuremap_slice", C.c_int, [_vp, C.c_int64, _fp]),
    ("fdcm_featuremap_device_volume", C.c_int, [_vp, C.POINTER(_vp)]),
    ("fdcm_featuremap_device_volume_stride", C.c_int, [_vp, C.POINTER(C.c_int64)]),
    ("fdcm_featuremap_last_timing", C.c_int, [_vp, C.POINTER(BuildTiming)]),
    ("fdcm_featuremap_stage_timing", C.c_int, [_vp, C.c_int]),
    ("fdcm_featuremap_from_slices", C.c_int, [_fp, C.c_int64, _fp, C.c_int64, C.c_int64, _fp, C.POINTER(_vp)]),
    ("fdcm_featuremap_build_staged", C.c_int,
     [_fp, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(_vp)]),
    ("fdcm_featuremap_minmax_translation", C.c_int, [_vp, _fp, C.c_int64, _fp, _fp]),
    ("fdcm_featuremap_minmax_translation_batch", C.c_int, [_vp, _fp, _i64p, C.c_int64, _fp, _fp]),
    ("fdcm_featuremap_evaluate", C.c_int, [_vp, _fp, _i64p, C.c_int64, _fp, _i64p, _fp]),
    ("fdcm_templates_create", C.c_int, [_fp, _i64p, C.c_int64, C.POINTER(_vp)]),
    ("fdcm_templates_free", C.c_int, [_vp]),
    ("fdcm_templates_count", C.c_int, [_vp, _i64p, _i64p]),
    ("fdcm_templates_lengths", C.c_int, [_vp, _fp]),
    ("fdcm_search", C.c_int, [_vp, _vp, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_int32,
                              C.POINTER(_vp), _i64p]),
    ("fdcm_search_capacity", C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int64, _i64p]),
    ("fdcm_search_device", C.c_int, [_vp, _vp, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_int32,
                                     _vp, _i64p]),
    ("fdcm_search_last_timing", C.c_int, [_vp, C.POINTER(SearchTiming)]),
    ("fdcm_matches_free", None, [_vp]),
    ("fdcm_blocks_to_host", C.c_int, [_vp, C.c_int32, C.c_int64, _vp, C.POINTER(_vp), C.POINTER(C.c_int64)]),
    ("fdcm_topk", C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int32, C.c_int, C.c_float, C.c_int64, C.POINTER(_vp), _i64p]),
    ("fdcm_pipeline_create", C.c_int, [C.c_int64, C.c_float, C.c_float, C.c_int, _vp, C.c_int64, C.c_int64, C.c_int,
                                       C.c_int64, C.c_int32, C.c_int, C.POINTER(_vp)]),
    ("fdcm_pipeline_submit", C.c_int, [_vp, _fp, C.c_int64, _vp, _i64p]),
    ("fdcm_pipeline_wait", C.c_int, [_vp, C.c_int64, C.POINTER(_vp), _i64p, C.POINTER(BuildTiming),
                                     C.POINTER(SearchTiming)]),
    ("fdcm_pipeline_slots", C.c_int, [_vp, C.POINTER(C.c_int)]),
    ("fdcm_pipeline_free", C.c_int, [_vp]),
    ("fdcm_sharded_create", C.c_int, [C.POINTER(C.c_int), C.c_int, _fp, _i64p, C.c_int64, C.c_int64, C.c_float, C.c_float,
                                      C.c_int, C.c_int, C.POINTER(_vp)]),
    ("fdcm_sharded_search", C.c_int, [_vp, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.POINTER(_vp), _i64p]),
    ("fdcm_sharded_search_topk", C.c_int, [_vp, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_float,
                                           C.c_int64, C.POINTER(_vp), _i64p]),
    ("fdcm_sharded_set_frames_in_flight", C.c_int, [_vp, C.c_int]),
    ("fdcm_sharded_set_mode", C.c_int, [_vp, C.c_int]),
    ("fdcm_sharded_submit", C.c_int, [_vp, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int64, _i64p]),
    ("fdcm_sharded_submit_topk", C.c_int, [_vp, _fp, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_float,
                                           C.c_int64, _i64p]),
    ("fdcm_sharded_wait", C.c_int, [_vp, C.c_int64, C.POINTER(_vp), _i64p]),
    ("fdcm_sharded_info", C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _i64p, _i64p, _i64p]),
    ("fdcm_sharded_last_timing", C.c_int, [_vp, C.c_int, C.POINTER(BuildTiming), C.POINTER(SearchTiming)]),
    ("fdcm_sharded_free", C.c_int, [_vp]),
    ("fdcm_filter_in_range", C.c_int, [_fp, C.c_int64, _fp, C.c_float, C.c_float, _i64p, _i64p]),
    ("fdcm_penalize", C.c_int, [C.c_int, C.c_float, _vp, C.c_int64, _fp, C.c_int64]),
    ("fdcm_sort_matches", C.c_int, [_vp, C.c_int64]),
    ("fdcm_partial_sort_matches", C.c_int, [_vp, C.c_int64, C.c_int64]),
    ("fdcm_lines_read", C.c_int, [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int64)]),
    ("fdcm_lines_write", C.c_int, [C.c_char_p, _fp, C.c_int64]),
    ("fdcm_lines_free", None, [C.POINTER(C.c_float)]),
    ("fdcm_selftest_atanf", C.c_int64, [C.c_uint32, C.c_uint32, C.c_uint64]),
    ("fdcm_orientation_bins_mode", C.c_int, []),
    ("fdcm_selftest_sweep_ranges", C.c_int, [C.c_int]),
    ("fdcm_selftest_sweep_order_counts", C.c_int, [_i64p, _i64p]),
    ("fdcm_selftest_sweep_steals", C.c_int, [_vp, _i64p]),
]

_lib = None


class FdcmError(RuntimeError):
    pass


def loaded():
    """True once libfdcm_hip.so (and with it /opt/rocm's HIP runtime) is in the process."""
    return _lib is not None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FdcmError(
                f"{LIB_PATH} is missing: build it with `make -C openfdcm_amd/csrc` (or __graft_entry__.build()); "
                "openfdcm_amd has no CPU fallback")
        l = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != FDCM_OK:
        raise FdcmError(f"libfdcm_hip: {lib().fdcm_last_error().decode()} (code {rc})")


def fptr(a):
    return a.ctypes.data_as(_fp)


def as_records(lines):
    """(4, N) array-like (the reference's LineArray) -> (N, 4) contiguous float32 records."""
    a = np.asarray(lines, dtype=np.float32)
    if a.ndim == 1 and a.size % 4 == 0:
        a = a.reshape(4, -1)
    if a.ndim != 2 or a.shape[0] != 4:
        raise ValueError(f"expected a (4, N) line array, got shape {a.shape}")
    return np.ascontiguousarray(a.T)


def pack_templates(templates):
    """list of (4, N_i) LineArrays -> ((sum N_i, 4) float32 records, int64 offsets of length len + 1)."""
    n = len(templates)
    offsets = np.zeros(n + 1, dtype=np.int64)
    if n and all(isinstance(t, np.ndarray) and t.ndim == 2 and t.shape[0] == 4 for t in templates):
        # the usual case in one concatenate instead of a transpose + copy per template
        np.cumsum([t.shape[1] for t in templates], out=offsets[1:])
        if offsets[-1] == 0:
            return np.zeros((0, 4), dtype=np.float32), offsets
        return np.ascontiguousarray(np.concatenate(templates, axis=1).T, dtype=np.float32), offsets
    recs = [as_records(t) for t in templates]
    for i, r in enumerate(recs):
        offsets[i + 1] = offsets[i] + r.shape[0]
    if recs and offsets[-1] > 0:
        flat = np.ascontiguousarray(np.concatenate(recs, axis=0), dtype=np.float32)
    else:
        flat = np.zeros((0, 4), dtype=np.float32)
    return flat, offsets
