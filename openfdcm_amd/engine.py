"""Thin object layer over the C ABI: HBM-resident feature maps and template sets."""
import ctypes as C

import numpy as np

from . import _capi as capi


def _adopt_matches(ptr, n):
    """Wrap a library-allocated fdcm_match array as a structured numpy array without copying; the
    allocation is released (fdcm_matches_free) when the array is garbage collected."""
    import weakref
    if not ptr or n == 0:
        if ptr:
            capi.lib().fdcm_matches_free(ptr)
        return np.zeros(0, dtype=capi.MATCH_DTYPE)
    raw = (C.c_char * (n * capi.MATCH_DTYPE.itemsize)).from_address(ptr.value)
    res = np.frombuffer(raw, dtype=capi.MATCH_DTYPE)
    weakref.finalize(raw, capi.lib().fdcm_matches_free, C.c_void_p(ptr.value))
    return res


class DeviceFeatureMap:
    """Owns an fdcm_featuremap handle (DT3 volume resident in HBM)."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle) if not isinstance(handle, C.c_void_p) else handle
        self.refresh()

    def refresh(self):
        info = capi.FeaturemapInfo()
        capi.check(capi.lib().fdcm_featuremap_get_info(self._h, C.byref(info)))
        self.width, self.height, self.depth = info.width, info.height, info.depth
        self.scene_translation = np.array(list(info.scene_translation), dtype=np.float32)
        self.distance = info.distance
        self.keys = np.zeros(self.depth, dtype=np.float32)
        if self.depth:
            capi.check(capi.lib().fdcm_featuremap_keys(self._h, capi.fptr(self.keys)))

    @classmethod
    def build(cls, scene, depth=30, coeff=5.0, padding=2.2, distance=capi.L2, stop_after=3):
        rec = capi.as_records(scene)
        h = C.c_void_p()
        if stop_after == 3:
            rc = capi.lib().fdcm_featuremap_build(capi.fptr(rec), rec.shape[0], int(depth), float(coeff),
                                                  float(padding), int(distance), C.byref(h))
        else:
            rc = capi.lib().fdcm_featuremap_build_staged(capi.fptr(rec), rec.shape[0], int(depth), float(coeff),
                                                         float(padding), int(distance), int(stop_after), C.byref(h))
        capi.check(rc)
        return cls(h)

    @classmethod
    def from_volume(cls, keys, volume, scene_translation):
        """volume: (depth, W, H) float32, [k][x][y]."""
        keys = np.ascontiguousarray(keys, dtype=np.float32)
        vol = np.ascontiguousarray(volume, dtype=np.float32)
        st = np.ascontiguousarray(scene_translation, dtype=np.float32)
        if vol.size == 0:
            m, W, H = len(keys), 0, 0
        else:
            m, W, H = vol.shape
        h = C.c_void_p()
        capi.check(capi.lib().fdcm_featuremap_from_slices(capi.fptr(keys), m, capi.fptr(vol), W, H, capi.fptr(st),
                                                          C.byref(h)))
        return cls(h)

    def rebuild(self, scene):
        rec = capi.as_records(scene)
        capi.check(capi.lib().fdcm_featuremap_rebuild(self._h, capi.fptr(rec), rec.shape[0]))
        self.refresh()

    def slice(self, k):
        """Slice k as an (H, W) array (column-major in memory, as the reference's RawImage)."""
        out = np.zeros((self.width, self.height), dtype=np.float32)
        capi.check(capi.lib().fdcm_featuremap_slice(self._h, int(k), capi.fptr(out)))
        return out.T

    def volume(self):
        """(depth, W, H) float32 copy of the device volume, [k][x][y]."""
        out = np.zeros((self.depth, self.width, self.height), dtype=np.float32)
        for k in range(self.depth):
            capi.check(capi.lib().fdcm_featuremap_slice(self._h, k, capi.fptr(out[k])))
        return out

    def device_pointer(self):
        """Address of the volume in HBM; layout: element k * device_slice_stride() + ((x // 4) * H + y) * 4 + x % 4."""
        p = C.c_void_p()
        capi.check(capi.lib().fdcm_featuremap_device_volume(self._h, C.byref(p)))
        return p.value

    def device_slice_stride(self):
        n = C.c_int64()
        capi.check(capi.lib().fdcm_featuremap_device_volume_stride(self._h, C.byref(n)))
        return n.value

    def minmax_translation(self, tmpl, align_vec):
        """FeatureMap::minmaxTranslation (featuremap.h:113-115 -> dt3cpu.cpp:30-75,119-124): (negative, positive)
        multiplier limits of align_vec for the (4, N) template; runs on the device."""
        rec = capi.as_records(tmpl)
        av = np.ascontiguousarray(align_vec, dtype=np.float32).reshape(2)
        out = np.zeros(2, dtype=np.float32)
        capi.check(capi.lib().fdcm_featuremap_minmax_translation(self._h, capi.fptr(rec), rec.shape[0], capi.fptr(av),
                                                                 capi.fptr(out)))
        return out

    def minmax_translation_batch(self, templates, align_vecs):
        """One launch for a list of templates with one align vector each -> (T, 2) float32."""
        flat, offsets = capi.pack_templates(templates)
        av = np.ascontiguousarray(align_vecs, dtype=np.float32).reshape(len(offsets) - 1, 2)
        out = np.zeros((len(offsets) - 1, 2), dtype=np.float32)
        capi.check(capi.lib().fdcm_featuremap_minmax_translation_batch(
            self._h, capi.fptr(flat), offsets.ctypes.data_as(C.POINTER(C.c_int64)), len(offsets) - 1, capi.fptr(av),
            capi.fptr(out)))
        return out

    def evaluate(self, templates, translations):
        """FeatureMap::evaluate (featuremap.h:117-120 -> dt3cpu.cpp:126-179): templates = list of (4, N) arrays,
        translations = per template a list / (n, 2) array of (x, y); returns a list of float32 arrays.  One launch."""
        flat, offsets = capi.pack_templates(templates)
        if len(translations) != len(offsets) - 1:
            raise ValueError("one list of translations per template is required")
        trs = [np.ascontiguousarray(t, dtype=np.float32).reshape(-1, 2) for t in translations]
        toff = np.zeros(len(trs) + 1, dtype=np.int64)
        for i, t in enumerate(trs):
            toff[i + 1] = toff[i] + t.shape[0]
        tflat = np.ascontiguousarray(np.concatenate(trs, axis=0)) if trs and toff[-1] else np.zeros((0, 2), dtype=np.float32)
        scores = np.zeros(int(toff[-1]), dtype=np.float32)
        capi.check(capi.lib().fdcm_featuremap_evaluate(
            self._h, capi.fptr(flat), offsets.ctypes.data_as(C.POINTER(C.c_int64)), len(offsets) - 1, capi.fptr(tflat),
            toff.ctypes.data_as(C.POINTER(C.c_int64)), capi.fptr(scores)))
        return [scores[toff[i]:toff[i + 1]].copy() for i in range(len(trs))]

    def stage_timing(self, on):
        """Device-side times cost an event between the kernels: True / 1 per-stage times (default), 2 the build's and the
        search's spans only, False / 0 none (build_timing() / search_timing() carry host time and counters only)."""
        capi.check(capi.lib().fdcm_featuremap_stage_timing(self._h, int(on)))

    def build_timing(self):
        t = capi.BuildTiming()
        capi.check(capi.lib().fdcm_featuremap_last_timing(self._h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in t._fields_}

    def search_timing(self):
        t = capi.SearchTiming()
        capi.check(capi.lib().fdcm_search_last_timing(self._h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in t._fields_}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            capi.lib().fdcm_featuremap_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceTemplates:
    """Owns an fdcm_templates handle: a list of LineArrays resident in HBM."""

    def __init__(self, templates, _packed=None):
        flat, offsets = _packed if _packed is not None else capi.pack_templates(templates)
        self.count = len(offsets) - 1
        h = C.c_void_p()
        capi.check(capi.lib().fdcm_templates_create(capi.fptr(flat), offsets.ctypes.data_as(C.POINTER(C.c_int64)),
                                                    self.count, C.byref(h)))
        self._h = h

    def lengths(self):
        out = np.zeros(self.count, dtype=np.float32)
        capi.check(capi.lib().fdcm_templates_lengths(self._h, capi.fptr(out)))
        return out

    def capacity(self, n_scene, max_tmpl_lines, max_scene_lines):
        cap = C.c_int64()
        capi.check(capi.lib().fdcm_search_capacity(self._h, n_scene, max_tmpl_lines, max_scene_lines, C.byref(cap)))
        return cap.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            capi.lib().fdcm_templates_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def search_raw(fm, templates, scene, max_tmpl_lines, max_scene_lines, optimizer=capi.BATCH_OPTIMIZE, batch_size=10,
               tmpl_index_base=0):
    """Run the search and return the raw matches as a structured array (capi.MATCH_DTYPE)."""
    rec = capi.as_records(scene)
    out = C.c_void_p()
    n = C.c_int64()
    capi.check(capi.lib().fdcm_search(fm._h, templates._h, capi.fptr(rec), rec.shape[0], int(max_tmpl_lines),
                                      int(max_scene_lines), int(optimizer), int(batch_size), int(tmpl_index_base),
                                      C.byref(out), C.byref(n)))
    return _adopt_matches(out, n.value)


def search_into(fm, templates, scene, max_tmpl_lines, max_scene_lines, optimizer, batch_size, tmpl_index_base,
                device_ptr):
    """Search leaving the matches in a caller-provided device buffer; returns the count."""
    rec = capi.as_records(scene)
    n = C.c_int64()
    capi.check(capi.lib().fdcm_search_device(fm._h, templates._h, capi.fptr(rec), rec.shape[0], int(max_tmpl_lines),
                                             int(max_scene_lines), int(optimizer), int(batch_size),
                                             int(tmpl_index_base), C.c_void_p(device_ptr), C.byref(n)))
    return n.value


class FramePipeline:
    """Owns an fdcm_pipeline: `slots` frames in flight, each running rebuild + search on its own HIP
    stream and host worker thread (include/fdcm.h, "frame pipeline")."""

    def __init__(self, templates, depth=30, coeff=5.0, padding=2.2, distance=capi.L2, max_tmpl_lines=4,
                 max_scene_lines=4, optimizer=capi.BATCH_OPTIMIZE, batch_size=10, tmpl_index_base=0, slots=2):
        self.templates = templates  # keeps the handle alive
        h = C.c_void_p()
        capi.check(capi.lib().fdcm_pipeline_create(int(depth), float(coeff), float(padding), int(distance),
                                                   templates._h, int(max_tmpl_lines), int(max_scene_lines),
                                                   int(optimizer), int(batch_size), int(tmpl_index_base), int(slots),
                                                   C.byref(h)))
        self._h = h
        self.slots = int(slots)
        self.last_build_timing = None
        self.last_search_timing = None

    def submit(self, scene, device_ptr=None, prepared=False):
        """Queue one frame; returns its ticket.  `scene` is a (4, N) LineArray, or with prepared=True the
        (N, 4) float32 records capi.as_records() returns."""
        rec = scene if prepared else capi.as_records(scene)
        t = C.c_int64()
        capi.check(capi.lib().fdcm_pipeline_submit(self._h, capi.fptr(rec), rec.shape[0],
                                                   C.c_void_p(device_ptr) if device_ptr else None, C.byref(t)))
        return t.value

    def wait(self, ticket, to_host=True):
        """Block until frame `ticket` is complete.  Returns the raw matches (structured array) when the frame
        was submitted without a device buffer, else the match count."""
        out, n = C.c_void_p(), C.c_int64()
        bt, st = capi.BuildTiming(), capi.SearchTiming()
        capi.check(capi.lib().fdcm_pipeline_wait(self._h, int(ticket), C.byref(out), C.byref(n), C.byref(bt),
                                                 C.byref(st)))
        self.last_build_timing = {k: getattr(bt, k) for k, _ in bt._fields_}
        self.last_search_timing = {k: getattr(st, k) for k, _ in st._fields_}
        if not out:
            return n.value
        return _adopt_matches(out, n.value)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            capi.lib().fdcm_pipeline_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def topk(fm, templates, k, penalty=None, tau=1.0, tmpl_index_base=0, device_ptr=None, n=0):
    """penalize + sort_matches + [:k] on the device (include/fdcm.h, "device tail").

    Works on the raw matches of the last search_raw() on `fm`, or on a device buffer filled by
    search_into() (`device_ptr`, `n` records).  penalty: None, capi.DEFAULT_PENALTY or
    capi.EXPONENTIAL_PENALTY (with tau).  Returns a structured array (capi.MATCH_DTYPE) of min(k, n)
    records, ascending penalised score, ties in positional order."""
    out, n_out = C.c_void_p(), C.c_int64()
    capi.check(capi.lib().fdcm_topk(fm._h, templates._h, C.c_void_p(device_ptr) if device_ptr else None, int(n),
                                    int(tmpl_index_base), -1 if penalty is None else int(penalty), float(tau), int(k),
                                    C.byref(out), C.byref(n_out)))
    return _adopt_matches(out, n_out.value)


class ShardedEngine:
    """Template shards over several GPUs of this node from one process (fdcm_sharded_* in include/fdcm.h): every
    device rebuilds the DT3 volume and searches a contiguous template range; the records (or the k best of every
    shard) reach the first device in one grouped RCCL send/recv per frame.  Returns what the single-device calls
    return for the whole template list."""

    def __init__(self, templates, devices=None, n_devices=None, depth=30, coeff=5.0, padding=2.2, distance=capi.L2,
                 always_collective=False, allow_same_device=False):
        flat, offsets = capi.pack_templates(templates)
        if devices is not None:
            n_devices = len(devices)
            dev = (C.c_int * n_devices)(*devices)
        else:
            n_devices = n_devices or 1
            dev = None
        h = C.c_void_p()
        capi.check(capi.lib().fdcm_sharded_create(dev, n_devices, capi.fptr(flat) if flat.size else None,
                                                  offsets.ctypes.data_as(C.POINTER(C.c_int64)), len(templates), depth, coeff,
                                                  padding, distance, (capi.SHARDED_ALWAYS_COLLECTIVE if always_collective else 0) |
                                                  (capi.SHARDED_ALLOW_SAME_DEVICE if allow_same_device else 0),
                                                  C.byref(h)))
        self._h, self.n_devices, self.n_templates = h, n_devices, len(templates)

    def info(self):
        n = C.c_int()
        devs = (C.c_int * self.n_devices)()
        begin = (C.c_int64 * (self.n_devices + 1))()
        coll, moved = C.c_int64(), C.c_int64()
        capi.check(capi.lib().fdcm_sharded_info(self._h, C.byref(n), devs, begin, C.byref(coll), C.byref(moved)))
        return {"devices": list(devs), "shard_begin": list(begin), "collectives": coll.value, "bytes_moved": moved.value}

    def search(self, scene, max_tmpl_lines, max_scene_lines, optimizer=capi.BATCH_OPTIMIZE, batch_size=10):
        rec = capi.as_records(scene)
        out, n = C.c_void_p(), C.c_int64()
        capi.check(capi.lib().fdcm_sharded_search(self._h, capi.fptr(rec) if rec.size else None, rec.shape[0], max_tmpl_lines,
                                                  max_scene_lines, optimizer, batch_size, C.byref(out), C.byref(n)))
        return _adopt_matches(out, n.value)

    def search_topk(self, scene, max_tmpl_lines, max_scene_lines, k, penalty=None, tau=1.0, optimizer=capi.BATCH_OPTIMIZE,
                    batch_size=10):
        rec = capi.as_records(scene)
        out, n = C.c_void_p(), C.c_int64()
        capi.check(capi.lib().fdcm_sharded_search_topk(self._h, capi.fptr(rec) if rec.size else None, rec.shape[0],
                                                       max_tmpl_lines, max_scene_lines, optimizer, batch_size,
                                                       -1 if penalty is None else penalty, tau, k, C.byref(out), C.byref(n)))
        return _adopt_matches(out, n.value)

    def set_mode(self, mode):
        """capi.SHARD_TEMPLATES (default: every frame on every device, template ranges, one exchange per frame) or
        capi.SHARD_FRAMES (ticket t whole on device t % n_devices over the whole template list, no exchange).  Only while
        no frame is in flight; tickets restart at 0."""
        capi.check(capi.lib().fdcm_sharded_set_mode(self._h, int(mode)))

    def set_frames_in_flight(self, n_frames):
        """Frame slots per device (1..16); only while no frame is in flight."""
        capi.check(capi.lib().fdcm_sharded_set_frames_in_flight(self._h, int(n_frames)))

    def submit(self, scene, max_tmpl_lines, max_scene_lines, optimizer=capi.BATCH_OPTIMIZE, batch_size=10, k=None,
               penalty=None, tau=1.0, prepared=False):
        """Queue one frame on every device and return its ticket (k given: top-k mode)."""
        rec = scene if prepared else capi.as_records(scene)
        t = C.c_int64()
        if k is None:
            capi.check(capi.lib().fdcm_sharded_submit(self._h, capi.fptr(rec) if rec.size else None, rec.shape[0], max_tmpl_lines,
                                                      max_scene_lines, optimizer, batch_size, C.byref(t)))
        else:
            capi.check(capi.lib().fdcm_sharded_submit_topk(self._h, capi.fptr(rec) if rec.size else None, rec.shape[0],
                                                           max_tmpl_lines, max_scene_lines, optimizer, batch_size,
                                                           -1 if penalty is None else penalty, tau, k, C.byref(t)))
        return t.value

    def wait(self, ticket):
        """Collect a frame: its exchange runs here while the devices compute the frames submitted after it."""
        out, n = C.c_void_p(), C.c_int64()
        capi.check(capi.lib().fdcm_sharded_wait(self._h, int(ticket), C.byref(out), C.byref(n)))
        return _adopt_matches(out, n.value)

    def timing(self, shard=0):
        bt, st = capi.BuildTiming(), capi.SearchTiming()
        capi.check(capi.lib().fdcm_sharded_last_timing(self._h, shard, C.byref(bt), C.byref(st)))
        return ({k: getattr(bt, k) for k, _ in bt._fields_}, {k: getattr(st, k) for k, _ in st._fields_})

    def close(self):
        if self._h:
            capi.check(capi.lib().fdcm_sharded_free(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
