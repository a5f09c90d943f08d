// fdcm_capi.cpp -- extern "C" entry points of libfdcm_hip.so (include/fdcm.h).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <numeric>
#include <thread>

#include "fdcm_internal.h"
#include "fdcm_sweep.h"

namespace fdcm {
int device_cus(int device) {
    static std::mutex mu;
    static int cached[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    int& c = cached[device & 63];
    if (c <= 0) FDCM_HIP(hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, device));
    return c;
}
}  // namespace fdcm

namespace fdcm {
const char* last_error_cstr();
static thread_local int g_device = 0;

template <class F>
static int guarded(F&& f) {
    try {
        f();
        return FDCM_OK;
    } catch (const HipError& e) {
        set_error(std::string("HIP error: ") + hipGetErrorString(e.code) + " in " + e.what);
        return FDCM_EHIP;
    } catch (const std::string& s) {
        set_error(s);
        return FDCM_EINVAL;
    } catch (const std::exception& e) {
        set_error(e.what());
        return FDCM_EINTERNAL;
    } catch (...) {
        set_error("unknown error");
        return FDCM_EINTERNAL;
    }
}

static void require(bool ok, const char* msg) {
    if (!ok) throw std::string(msg);
}

static void upload_keys_only(fdcm_featuremap* fm) {
    // feature maps adopted from caller slices carry no build plan: keep the keys where search expects them
    fm->off_keys = 0;
    const size_t bytes = std::max<size_t>(16, fm->keys.size() * sizeof(float));
    fm->plan.reserve(bytes);
    if (!fm->keys.empty())
        FDCM_HIP(hipMemcpy(fm->plan.p, fm->keys.data(), fm->keys.size() * sizeof(float), hipMemcpyHostToDevice));
}

static void destroy(fdcm_featuremap* fm) {
    if (!fm) return;
    (void)hipSetDevice(fm->device);
    if (fm->stream) (void)hipStreamSynchronize(fm->stream);
    if (fm->prep_stream) (void)hipStreamSynchronize(fm->prep_stream);
    fm->vol.release(); fm->ivol.release(); fm->bitmap.release(); fm->coldesc.release(); fm->colmask.release(); fm->offtab.release(); fm->stack.release(); fm->plan.release(); fm->stage.release();
    fm->s_scene.release(); fm->s_pairs.release(); fm->s_records.release(); fm->s_flags.release(); fm->s_out.release(); fm->s_work.release(); fm->s_tail.release(); fm->s_tail_out.release(); fm->s_eval.release();
    fm->s_counter.release(); fm->s_stage.release(); fm->s_cnt.release(); fm->s_bins.release(); fm->s_bins_stage.release();
    if (fm->timing.created)
        for (auto& e : fm->timing.ev) (void)hipEventDestroy(e);
    if (fm->stream) (void)hipStreamDestroy(fm->stream);
    if (fm->prep_stream) { (void)hipStreamDestroy(fm->prep_stream); (void)hipEventDestroy(fm->prep_done); }
    delete fm;
}
}  // namespace fdcm

using namespace fdcm;

extern "C" {

const char* fdcm_last_error(void) { return last_error_cstr(); }
const char* fdcm_version(void) { return "openfdcm_amd 0.1.0 (gfx950)"; }

int fdcm_device_count(int* count) {
    return guarded([&] {
        require(count != nullptr, "count is null");
        FDCM_HIP(hipGetDeviceCount(count));
    });
}

int fdcm_set_device(int device) {
    return guarded([&] {
        FDCM_HIP(hipSetDevice(device));
        g_device = device;
    });
}

int fdcm_get_device(int* device) {
    return guarded([&] {
        require(device != nullptr, "device is null");
        *device = g_device;
    });
}

int fdcm_featuremap_build_staged(const float* scene_lines, int64_t n_lines, int64_t depth, float dt3_coeff,
                                 float padding, int distance, int stop_after, fdcm_featuremap** out) {
    fdcm_featuremap* fm = nullptr;
    int rc = guarded([&] {
        require(out != nullptr, "out is null");
        require(n_lines >= 0 && (n_lines == 0 || scene_lines), "bad scene_lines");
        require(depth >= 0, "depth must be >= 0");
        require(distance >= FDCM_L2 && distance <= FDCM_L1, "unknown distance");
        require(stop_after >= 1 && stop_after <= 3, "stop_after must be 1..3");
        fm = new fdcm_featuremap();
        fm->device = g_device;
        fm->depth_param = depth; fm->coeff = dt3_coeff; fm->padding = padding; fm->distance = distance;
        BuildPlan plan;
        make_build_plan(scene_lines, depth > 0 ? n_lines : 0, depth, dt3_coeff, padding, plan);
        run_build(fm, plan, stop_after);
        *out = fm;
    });
    if (rc != FDCM_OK) { destroy(fm); if (out) *out = nullptr; }
    return rc;
}

int fdcm_featuremap_build(const float* scene_lines, int64_t n_lines, int64_t depth, float dt3_coeff, float padding,
                          int distance, fdcm_featuremap** out) {
    return fdcm_featuremap_build_staged(scene_lines, n_lines, depth, dt3_coeff, padding, distance, 3, out);
}

int fdcm_featuremap_rebuild(fdcm_featuremap* fm, const float* scene_lines, int64_t n_lines) {
    return guarded([&] {
        require(fm != nullptr, "featuremap is null");
        require(n_lines >= 0 && (n_lines == 0 || scene_lines), "bad scene_lines");
        BuildPlan plan;
        make_build_plan(scene_lines, fm->depth_param > 0 ? n_lines : 0, fm->depth_param, fm->coeff, fm->padding, plan);
        run_build(fm, plan, 3);
    });
}

int fdcm_featuremap_free(fdcm_featuremap* fm) {
    destroy(fm);
    return FDCM_OK;
}

int fdcm_featuremap_get_info(const fdcm_featuremap* fm, fdcm_featuremap_info* info) {
    return guarded([&] {
        require(fm && info, "null argument");
        info->width = fm->W; info->height = fm->H; info->depth = fm->m;
        info->scene_translation[0] = fm->tx; info->scene_translation[1] = fm->ty;
        info->distance = fm->distance; info->dt3_coeff = fm->coeff; info->padding = fm->padding;
    });
}

int fdcm_featuremap_keys(const fdcm_featuremap* fm, float* keys) {
    return guarded([&] {
        require(fm && (keys || fm->keys.empty()), "null argument");
        if (!fm->keys.empty()) std::memcpy(keys, fm->keys.data(), fm->keys.size() * sizeof(float));
    });
}

int fdcm_featuremap_slice(const fdcm_featuremap* fm, int64_t k, float* out_host) {
    return guarded([&] {
        require(fm && out_host, "null argument");
        require(k >= 0 && k < fm->m, "slice index out of range");
        finish_build(const_cast<fdcm_featuremap*>(fm));
        FDCM_HIP(hipSetDevice(fm->device));
        const size_t npix = (size_t)fm->W * fm->H;
        if (!fm->current_interleaved()) {  // a partial build (tests of the stages) that stopped before the propagation
            FDCM_HIP(hipMemcpy(out_host, fm->vol.as<float>() + (size_t)k * npix, npix * sizeof(float), hipMemcpyDeviceToHost));
            return;
        }
        const size_t sl = ivol_slice_floats(fm->W, fm->H);
        std::vector<float> tmp(sl);
        FDCM_HIP(hipMemcpy(tmp.data(), fm->current() + (size_t)k * sl, sl * sizeof(float), hipMemcpyDeviceToHost));
        for (int64_t x = 0; x < fm->W; ++x)
            for (int64_t y = 0; y < fm->H; ++y) out_host[(size_t)x * fm->H + y] = tmp[ivol_index((int)x, (int)y, fm->H)];
    });
}

int fdcm_featuremap_device_volume(const fdcm_featuremap* fm, const float** device_ptr) {
    return guarded([&] {
        require(fm && device_ptr, "null argument");
        finish_build(const_cast<fdcm_featuremap*>(fm));  // the caller may read it from any stream
        *device_ptr = fm->current();
    });
}

int fdcm_featuremap_device_volume_stride(const fdcm_featuremap* fm, int64_t* floats_per_slice) {
    return guarded([&] {
        require(fm && floats_per_slice, "null argument");
        *floats_per_slice = fm->current_interleaved() ? (int64_t)ivol_slice_floats(fm->W, fm->H) : fm->W * fm->H;
    });
}

int fdcm_featuremap_last_timing(const fdcm_featuremap* fm, fdcm_build_timing* t) {
    return guarded([&] {
        require(fm && t, "null argument");
        finish_build(const_cast<fdcm_featuremap*>(fm));
        *t = fm->last_build;
    });
}

int fdcm_featuremap_stage_timing(fdcm_featuremap* fm, int on) {
    return guarded([&] {
        require(fm != nullptr, "featuremap is null");
        require(on >= 0 && on <= 2, "stage timing: 0 (off), 1 (per stage) or 2 (totals only)");
        fm->want_stage_events = on;
    });
}

int fdcm_featuremap_from_slices(const float* keys, int64_t depth, const float* volume_host, int64_t width,
                                int64_t height, const float scene_translation[2], fdcm_featuremap** out) {
    fdcm_featuremap* fm = nullptr;
    int rc = guarded([&] {
        require(out && scene_translation, "null argument");
        require(depth >= 0 && width >= 0 && height >= 0, "negative size");
        require(depth == 0 || (keys && volume_host), "null keys/volume");
        for (int64_t i = 1; i < depth; ++i) require(keys[i - 1] < keys[i], "keys must be strictly ascending (std::map order)");
        fm = new fdcm_featuremap();
        fm->device = g_device;
        FDCM_HIP(hipSetDevice(fm->device));
        fm->depth_param = depth; fm->m = depth; fm->W = width; fm->H = height;
        fm->tx = scene_translation[0]; fm->ty = scene_translation[1];
        fm->keys.assign(keys, keys + depth);
        require(width <= 16384 && height <= 16384, "feature size above 16384 is not supported");
        const size_t sl = ivol_slice_floats(width, height);
        if (depth && sl) {  // into the interleaved layout the search reads, one slice at a time
            fm->vol.reserve((size_t)depth * sl * sizeof(float));
            std::vector<float> tmp(sl, 0.f);
            for (int64_t k = 0; k < depth; ++k) {
                const float* src = volume_host + (size_t)k * width * height;
                for (int64_t x = 0; x < width; ++x)
                    for (int64_t y = 0; y < height; ++y) tmp[ivol_index((int)x, (int)y, height)] = src[(size_t)x * height + y];
                FDCM_HIP(hipMemcpy(fm->vol.as<float>() + (size_t)k * sl, tmp.data(), sl * sizeof(float), hipMemcpyHostToDevice));
            }
        }
        fm->vol_stage = 3;
        upload_keys_only(fm);
        *out = fm;
    });
    if (rc != FDCM_OK) { destroy(fm); if (out) *out = nullptr; }
    return rc;
}

// ------------------------------------------------------------------------------------------ feature-map seam
static void check_offsets(const int64_t* off, int64_t n, const char* what) {
    require(off != nullptr, what);
    require(off[0] == 0, "offsets[0] must be 0");
    for (int64_t i = 0; i < n; ++i) require(off[i] <= off[i + 1], "offsets must be ascending");
}

int fdcm_featuremap_minmax_translation_batch(const fdcm_featuremap* fm, const float* tmpl_lines, const int64_t* line_offsets,
                                             int64_t n_templates, const float* align_vecs, float* out_minmax) {
    return guarded([&] {
        require(fm != nullptr, "featuremap is null");
        require(n_templates >= 0, "negative template count");
        if (n_templates == 0) return;
        check_offsets(line_offsets, n_templates, "line_offsets is null");
        require(line_offsets[n_templates] == 0 || tmpl_lines, "tmpl_lines is null");
        require(align_vecs && out_minmax, "null align_vecs/out_minmax");
        run_minmax(const_cast<fdcm_featuremap*>(fm), tmpl_lines, line_offsets, n_templates, align_vecs, out_minmax);
    });
}

int fdcm_featuremap_minmax_translation(const fdcm_featuremap* fm, const float* tmpl_lines, int64_t n_lines,
                                       const float align_vec[2], float out_minmax[2]) {
    const int64_t off[2] = {0, n_lines};
    if (n_lines < 0) { set_error("negative line count"); return FDCM_EINVAL; }
    return fdcm_featuremap_minmax_translation_batch(fm, tmpl_lines, off, 1, align_vec, out_minmax);
}

int fdcm_featuremap_evaluate(const fdcm_featuremap* fm, const float* tmpl_lines, const int64_t* line_offsets,
                             int64_t n_templates, const float* translations, const int64_t* translation_offsets,
                             float* scores_out) {
    return guarded([&] {
        require(fm != nullptr, "featuremap is null");
        require(n_templates >= 0, "negative template count");
        if (n_templates == 0) return;
        check_offsets(line_offsets, n_templates, "line_offsets is null");
        check_offsets(translation_offsets, n_templates, "translation_offsets is null");
        require(line_offsets[n_templates] == 0 || tmpl_lines, "tmpl_lines is null");
        require(translation_offsets[n_templates] == 0 || (translations && scores_out), "null translations/scores_out");
        run_evaluate(const_cast<fdcm_featuremap*>(fm), tmpl_lines, line_offsets, n_templates, translations, translation_offsets,
                     scores_out);
    });
}

// ------------------------------------------------------------------------------------------ templates
int fdcm_templates_create(const float* lines, const int64_t* offsets, int64_t n_templates, fdcm_templates** out) {
    fdcm_templates* t = nullptr;
    int rc = guarded([&] {
        require(out != nullptr, "out is null");
        require(n_templates >= 0 && (n_templates == 0 || offsets), "bad offsets");
        t = new fdcm_templates();
        t->device = g_device;
        FDCM_HIP(hipSetDevice(t->device));
        t->T = n_templates;
        t->offsets.assign(n_templates + 1, 0);
        for (int64_t i = 0; i <= n_templates && offsets; ++i) t->offsets[i] = offsets[i];
        require(t->offsets[0] == 0, "offsets[0] must be 0");
        for (int64_t i = 0; i < n_templates; ++i) require(t->offsets[i] <= t->offsets[i + 1], "offsets must be ascending");
        t->n_lines = t->offsets[n_templates];
        require(t->n_lines == 0 || lines, "lines is null");
        t->lines.assign(lines, lines + 4 * t->n_lines);
        t->lengths.resize((size_t)t->n_lines);
        t->sorted.resize((size_t)t->n_lines);
        for (int64_t i = 0; i < t->n_lines; ++i) {  // getLength, math.h:306-308
            const float dx = lines[4 * i + 2] - lines[4 * i], dy = lines[4 * i + 3] - lines[4 * i + 1];
            t->lengths[i] = std::sqrt(dx * dx + dy * dy);
        }
        // argsort(tmpl_lengths, std::greater<>()), defaultsearch.cpp:35 / math.h:106-116: scene independent
        std::vector<long> ind;
        for (int64_t i = 0; i < n_templates; ++i) {
            const int64_t l0 = t->offsets[i], n = t->offsets[i + 1] - l0;
            t->max_lines = std::max(t->max_lines, n);
            ind.resize((size_t)n);
            std::iota(ind.begin(), ind.end(), 0);
            const float* len = t->lengths.data() + l0;
            std::sort(ind.begin(), ind.end(), [len](long const i1, long const i2) { return len[i1] > len[i2]; });
            for (int64_t j = 0; j < n; ++j) t->sorted[l0 + j] = (int32_t)ind[j];
        }
        t->d_lines.reserve(std::max<size_t>(16, t->lines.size() * 4));
        t->d_offsets.reserve((size_t)(n_templates + 1) * 8);
        t->d_lengths.reserve(std::max<size_t>(16, t->lengths.size() * 4));
        t->d_sorted.reserve(std::max<size_t>(16, t->sorted.size() * 4));
        if (t->n_lines) {
            FDCM_HIP(hipMemcpy(t->d_lines.p, t->lines.data(), t->lines.size() * 4, hipMemcpyHostToDevice));
            FDCM_HIP(hipMemcpy(t->d_lengths.p, t->lengths.data(), t->lengths.size() * 4, hipMemcpyHostToDevice));
            FDCM_HIP(hipMemcpy(t->d_sorted.p, t->sorted.data(), t->sorted.size() * 4, hipMemcpyHostToDevice));
        }
        FDCM_HIP(hipMemcpy(t->d_offsets.p, t->offsets.data(), (size_t)(n_templates + 1) * 8, hipMemcpyHostToDevice));
        *out = t;
    });
    if (rc != FDCM_OK) { if (t) fdcm_templates_free(t); if (out) *out = nullptr; }
    return rc;
}

int fdcm_templates_free(fdcm_templates* t) {
    if (!t) return FDCM_OK;
    (void)hipSetDevice(t->device);
    t->d_lines.release(); t->d_offsets.release(); t->d_lengths.release(); t->d_sorted.release();
    delete t;
    return FDCM_OK;
}

int fdcm_templates_count(const fdcm_templates* t, int64_t* n_templates, int64_t* n_lines) {
    return guarded([&] {
        require(t != nullptr, "templates is null");
        if (n_templates) *n_templates = t->T;
        if (n_lines) *n_lines = t->n_lines;
    });
}

int fdcm_templates_lengths(const fdcm_templates* t, float* lengths) {
    return guarded([&] {
        require(t && (lengths || t->T == 0), "null argument");
        // getTemplateLengths, math.h:319-324: getLength(tmpl).sum() (Eigen redux order)
        for (int64_t i = 0; i < t->T; ++i) {
            const float* v = t->lengths.data() + t->offsets[i];
            const int64_t n = t->offsets[i + 1] - t->offsets[i];
            float res = 0.f;
            if (n > 0) {
                const int64_t a2 = (n / 8) * 8, a1 = (n / 4) * 4;
                if (a1) {
                    float p0[4] = {v[0], v[1], v[2], v[3]};
                    if (a1 > 4) {
                        float p1[4] = {v[4], v[5], v[6], v[7]};
                        for (int64_t idx = 8; idx < a2; idx += 8)
                            for (int l = 0; l < 4; ++l) { p0[l] = p0[l] + v[idx + l]; p1[l] = p1[l] + v[idx + 4 + l]; }
                        for (int l = 0; l < 4; ++l) p0[l] = p0[l] + p1[l];
                        if (a1 > a2) for (int l = 0; l < 4; ++l) p0[l] = p0[l] + v[a2 + l];
                    }
                    res = (p0[0] + p0[2]) + (p0[1] + p0[3]);
                    for (int64_t idx = a1; idx < n; ++idx) res = res + v[idx];
                } else {
                    res = v[0];
                    for (int64_t idx = 1; idx < n; ++idx) res = res + v[idx];
                }
            }
            lengths[i] = res;
        }
    });
}

// ------------------------------------------------------------------------------------------ search
int fdcm_search_capacity(const fdcm_templates* templates, int64_t n_scene_lines, int64_t max_tmpl_lines,
                         int64_t max_scene_lines, int64_t* capacity) {
    return guarded([&] {
        require(templates && capacity, "null argument");
        *capacity = search_capacity(templates, n_scene_lines, max_tmpl_lines, max_scene_lines);
    });
}

static void check_search_args(const fdcm_featuremap* fm, const fdcm_templates* t, const float* scene, int64_t n_scene,
                              int64_t maxT, int64_t maxS, int optimizer, int64_t batch) {
    require(fm && t, "null featuremap/templates");
    require(n_scene >= 0 && (n_scene == 0 || scene), "bad scene_lines");
    require(maxT >= 0 && maxS >= 0, "negative search window");
    require(optimizer >= FDCM_DEFAULT_OPTIMIZE && optimizer <= FDCM_INDULGENT_OPTIMIZE, "unknown optimizer");
    require(optimizer != FDCM_BATCH_OPTIMIZE || batch >= 1, "batch_size must be >= 1");
    require(fm->device == t->device, "featuremap and templates live on different devices");
}

int fdcm_search_device(const fdcm_featuremap* fm, const fdcm_templates* templates, const float* scene_lines,
                       int64_t n_scene_lines, int64_t max_tmpl_lines, int64_t max_scene_lines, int optimizer,
                       int64_t batch_size, int32_t tmpl_index_base, fdcm_match* out_device, int64_t* n_out) {
    return guarded([&] {
        check_search_args(fm, templates, scene_lines, n_scene_lines, max_tmpl_lines, max_scene_lines, optimizer, batch_size);
        require(out_device && n_out, "null output");
        run_search(const_cast<fdcm_featuremap*>(fm), templates, scene_lines, n_scene_lines, max_tmpl_lines,
                   max_scene_lines, optimizer, batch_size, tmpl_index_base, out_device, nullptr, n_out);
    });
}

int fdcm_search(const fdcm_featuremap* fm, const fdcm_templates* templates, const float* scene_lines,
                int64_t n_scene_lines, int64_t max_tmpl_lines, int64_t max_scene_lines, int optimizer,
                int64_t batch_size, int32_t tmpl_index_base, fdcm_match** out, int64_t* n_out) {
    return guarded([&] {
        check_search_args(fm, templates, scene_lines, n_scene_lines, max_tmpl_lines, max_scene_lines, optimizer, batch_size);
        require(out && n_out, "null output");
        fdcm_featuremap* f = const_cast<fdcm_featuremap*>(fm);
        *out = nullptr;
        try {
            run_search(f, templates, scene_lines, n_scene_lines, max_tmpl_lines, max_scene_lines, optimizer, batch_size,
                       tmpl_index_base, nullptr, out, n_out);
        } catch (...) {
            result_release(*out);
            *out = nullptr;
            throw;
        }
        if (!*out) *out = result_acquire(sizeof(fdcm_match));  // no candidates: an empty (non-null) array
        f->last_n_out = *n_out;
    });
}

int fdcm_topk(fdcm_featuremap* fm, const fdcm_templates* templates, const fdcm_match* matches_device, int64_t n,
              int32_t tmpl_index_base, int penalty, float tau, int64_t k, fdcm_match** out, int64_t* n_out) {
    return guarded([&] {
        require(fm && templates && out && n_out, "null argument");
        require(penalty >= -1 && penalty <= FDCM_EXPONENTIAL_PENALTY, "unknown penalty");
        require(fm->device == templates->device, "featuremap and templates live on different devices");
        require(k >= 0, "k must be >= 0");
        if (!matches_device) {  // the matches of the last fdcm_search on this handle
            matches_device = fm->s_out.as<fdcm_match>();
            n = fm->last_n_out;
        }
        require(n >= 0 && (n == 0 || matches_device), "bad matches");
        try {
            run_topk(fm, templates, matches_device, n, tmpl_index_base, penalty, tau, k, out, n_out);
        } catch (...) {
            result_release(*out);
            *out = nullptr;
            throw;
        }
    });
}

int fdcm_search_last_timing(const fdcm_featuremap* fm, fdcm_search_timing* t) {
    return guarded([&] {
        require(fm && t, "null argument");
        *t = fm->last_search;
    });
}

void fdcm_matches_free(fdcm_match* m) { result_release(m); }

int fdcm_blocks_to_host(const void* blocks_device, int32_t n_blocks, int64_t capacity_records, void* stream, fdcm_match** out,
                        int64_t* n_out) {
    if (out) *out = nullptr;
    if (n_out) *n_out = 0;
    int rc = guarded([&] {
        require(out && n_out && n_blocks >= 0 && capacity_records >= 0 && (n_blocks == 0 || blocks_device), "bad arguments");
        require((int64_t)n_blocks * capacity_records < (int64_t)1 << 40, "too many records");
        if (n_blocks == 0 || capacity_records == 0) { *out = result_acquire(64); return; }
        // the kernel runs on the device that holds the blocks (whatever this thread's current device is), and the
        // caller's current device is left as it was
        int prev = -1;
        (void)hipGetDevice(&prev);
        hipPointerAttribute_t attr{};
        const int dev = hipPointerGetAttributes(&attr, blocks_device) == hipSuccess ? attr.device : g_device;
        (void)hipGetLastError();
        FDCM_HIP(hipSetDevice(dev));
        try {
            blocks_to_host((hipStream_t)stream, blocks_device, n_blocks, capacity_records, out, n_out);
        } catch (...) {
            if (prev >= 0) (void)hipSetDevice(prev);
            throw;
        }
        if (prev >= 0) (void)hipSetDevice(prev);
    });
    if (rc != FDCM_OK && out && *out) { result_release(*out); *out = nullptr; }
    return rc;
}

int fdcm_filter_in_range(const float* lines, int64_t n_lines, const float center[2], float low_boundary,
                         float high_boundary, int64_t* out_indices, int64_t* n_out) {
    return guarded([&] {
        require(n_lines >= 0 && (n_lines == 0 || lines) && center && n_out && (n_lines == 0 || out_indices), "bad arguments");
        int64_t k = 0;
        for (int64_t i = 0; i < n_lines; ++i) {  // filterInRange, concentricrange.h:73-84
            const float* p = lines + 4 * i;
            const float cx = (p[2] + p[0]) / 2 - center[0], cy = (p[3] + p[1]) / 2 - center[1];
            const float rad = std::sqrt(cx * cx + cy * cy);
            if (rad > (low_boundary - FLT_EPSILON) && rad < high_boundary) out_indices[k++] = i;
        }
        *n_out = k;
    });
}

// ------------------------------------------------------------------------------------------ tail
int fdcm_penalize(int penalty, float tau, fdcm_match* matches, int64_t n, const float* template_lengths,
                  int64_t n_templates) {
    return guarded([&] {
        require(n >= 0 && (n == 0 || matches), "bad matches");
        require(penalty == FDCM_DEFAULT_PENALTY || penalty == FDCM_EXPONENTIAL_PENALTY, "unknown penalty");
        // defaultpenalty.cpp:37-41 / exponentialpenalty.cpp:42-46; templatelengths.at() -> out_of_range
        for (int64_t i = 0; i < n; ++i)
            if (matches[i].tmpl_idx < 0 || matches[i].tmpl_idx >= n_templates)
                throw std::string("In penalize, the size of templatelengths is not consistent with match template indices");
        if (penalty == FDCM_DEFAULT_PENALTY) {
            for (int64_t i = 0; i < n; ++i) matches[i].score = matches[i].score / std::max(template_lengths[matches[i].tmpl_idx], 1e-6f);
            return;
        }
        // The divisor pow(len, tau) depends on the template only: one powf per template that occurs, not per match
        // (the same float operands give the same float result, so the scores are the reference's bit for bit).
        if (n < n_templates) {
            for (int64_t i = 0; i < n; ++i)
                matches[i].score = matches[i].score / std::pow(std::max(template_lengths[matches[i].tmpl_idx], 1e-6f), tau);
            return;
        }
        std::vector<float> div((size_t)n_templates);
        for (int64_t t = 0; t < n_templates; ++t) div[(size_t)t] = std::pow(std::max(template_lengths[t], 1e-6f), tau);
        for (int64_t i = 0; i < n; ++i) matches[i].score = matches[i].score / div[(size_t)matches[i].tmpl_idx];
    });
}

// sortMatches (matchstrategy.h:46-50): std::sort by score -- unstable, and which of two equal scores comes first is whatever
// libstdc++'s introsort does (a frame of 27 000 matches always holds a few equal scores, so nothing but that algorithm gives the
// reference's list).  Its moves depend on the comparisons only, so (a) sorting (score, position) pairs and gathering the records
// ends in the same permutation, and (b) so does running it on a few threads: introsort recurses on the right part of every
// partition and loops on the left, the parts never exchange elements again, and the closing insertion sort never moves an
// element across a partition boundary (everything left of it is <= everything right of it).  Here the first three partition
// levels hand their right parts to a small persistent pool and every part then runs libstdc++'s own std::__introsort_loop +
// std::__final_insertion_sort: 1.27 -> 0.49 ms for 27 025 records on the GPU box's host (tools/sim/sort_pool_bench.cpp;
// tests/test_matchlist.py compares with std::sort on the reference's Match structs, ties, +-0 and +-inf included).  Lists below
// 8192 records, lists with a NaN score (no strict weak order: the serial scan's behaviour is its own) and calls that find the
// pool busy take the plain std::sort.
namespace {
#if defined(__GLIBCXX__)
struct SortKey { float score; uint32_t pos; };
struct SortKeyLess { bool operator()(const SortKey& a, const SortKey& b) const { return a.score < b.score; } };
class SortPool {
    std::mutex mu;
    std::condition_variable cv_task, cv_done;
    std::deque<std::function<void()>> q;
    int pending = 0;
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_task.wait(lk, [&] { return !q.empty(); });
            auto f = std::move(q.front());
            q.pop_front();
            lk.unlock();
            f();
            lk.lock();
            if (--pending == 0) cv_done.notify_all();
        }
    }
public:
    std::mutex in_use;  // one sort at a time
    explicit SortPool(int helpers) {
        for (int i = 0; i < helpers; ++i) std::thread([this] { loop(); }).detach();  // (they live as long as the process: the pool is never destroyed)
    }
    void submit(std::function<void()> f) {
        { std::lock_guard<std::mutex> lk(mu); ++pending; q.push_back(std::move(f)); }
        cv_task.notify_one();
    }
    void help_and_wait() {  // the caller's thread works the queue too, then waits for the helpers' last tasks
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            if (!q.empty()) {
                auto f = std::move(q.front());
                q.pop_front();
                lk.unlock();
                f();
                lk.lock();
                if (--pending == 0) cv_done.notify_all();
                continue;
            }
            if (pending == 0) return;
            cv_done.wait(lk, [&] { return pending == 0 || !q.empty(); });
        }
    }
};
SortPool* sort_pool() {
    static SortPool* p = [] {
        const unsigned hc = std::thread::hardware_concurrency();
        return hc >= 4 ? new SortPool((int)std::min(7u, hc - 1)) : nullptr;
    }();
    return p;
}
void sort_part(SortKey* first, SortKey* last, long depth, int levels, SortPool* P) {
    auto cmp = __gnu_cxx::__ops::__iter_comp_iter(SortKeyLess{});
    while (levels > 0 && last - first > 2048 && depth > 0) {  // std::__introsort_loop's own steps, the right part to the pool
        --depth; --levels;
        SortKey* cut = std::__unguarded_partition_pivot(first, last, cmp);
        P->submit([=] { sort_part(cut, last, depth, levels, P); });
        last = cut;
    }
    std::__introsort_loop(first, last, depth, cmp);
    std::__final_insertion_sort(first, last, cmp);
}
bool sort_matches_parallel(fdcm_match* matches, int64_t n) {
    if (n < 8192 || n > (int64_t)UINT32_MAX) return false;
    SortPool* P = sort_pool();
    if (!P) return false;
    std::unique_lock<std::mutex> use(P->in_use, std::try_to_lock);
    if (!use.owns_lock()) return false;
    std::vector<SortKey> key((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        if (matches[i].score != matches[i].score) return false;  // NaN
        key[(size_t)i] = {matches[i].score, (uint32_t)i};
    }
    sort_part(key.data(), key.data() + n, (long)std::__lg(n) * 2, 3, P);
    P->help_and_wait();
    std::vector<fdcm_match> out((size_t)n);
    for (int64_t i = 0; i < n; ++i) out[(size_t)i] = matches[key[(size_t)i].pos];
    std::memcpy(matches, out.data(), (size_t)n * sizeof(fdcm_match));
    return true;
}
#else
bool sort_matches_parallel(fdcm_match*, int64_t) { return false; }
#endif
}  // namespace

int fdcm_sort_matches(fdcm_match* matches, int64_t n) {
    return guarded([&] {
        require(n >= 0 && (n == 0 || matches), "bad matches");
        if (sort_matches_parallel(matches, n)) return;
        std::sort(matches, matches + n, [](const fdcm_match& a, const fdcm_match& b) { return a.score < b.score; });
    });
}

// sortMatches(matches, maxNumCandidates), matchstrategy.h:52-55: std::partial_sort of the first min(k, n) places
int fdcm_partial_sort_matches(fdcm_match* matches, int64_t n, int64_t max_num_candidates) {
    return guarded([&] {
        require(n >= 0 && (n == 0 || matches), "bad matches");
        require(max_num_candidates >= 0, "max_num_candidates must be >= 0");
        std::partial_sort(matches, matches + std::min<int64_t>(max_num_candidates, n), matches + n,
                          [](const fdcm_match& a, const fdcm_match& b) { return a.score < b.score; });
    });
}

// ------------------------------------------------------------------------------------------ line files (fdcm_lineio.cpp)
int fdcm_lines_read(const char* path, float** lines, int64_t* n_lines) {
    return guarded([&] {
        require(path && lines && n_lines, "null argument");
        *lines = nullptr; *n_lines = 0;
        lines_read(path, lines, n_lines);
    });
}
int fdcm_lines_write(const char* path, const float* lines, int64_t n_lines) {
    return guarded([&] {
        require(path && n_lines >= 0 && (n_lines == 0 || lines), "bad arguments");
        lines_write(path, lines, n_lines);
    });
}
void fdcm_lines_free(float* lines) { std::free(lines); }

// ------------------------------------------------------------------------------------------ self checks
int64_t fdcm_selftest_atanf(uint32_t first, uint32_t stride, uint64_t count) {
    if (stride == 0) stride = 1;
    unsigned nt = std::max(1u, std::thread::hardware_concurrency());
    std::vector<uint64_t> bad(nt, 0);
    std::vector<std::thread> th;
    for (unsigned w = 0; w < nt; ++w)
        th.emplace_back([&, w] {
            for (uint64_t i = w; i < count; i += nt) {
                const uint32_t u = first + (uint32_t)(i * stride);
                const float x = f_from_bits(u);
                const float a = atanf_glibc(x), b = atanf(x);
                if (bits_from_f(a) != bits_from_f(b) && !(f_isnan(a) && f_isnan(b))) ++bad[w];
            }
        });
    for (auto& t : th) t.join();
    uint64_t total = 0;
    for (auto b : bad) total += b;
    return (int64_t)total;
}

int fdcm_orientation_bins_mode(void) { return fdcm::orientation_bins_on_host() ? 1 : 0; }
int fdcm_selftest_sweep_order_counts(int64_t* from_history, int64_t* from_proxy) {
    return guarded([&] {
        require(from_history && from_proxy, "null argument");
        fdcm::sweep_order_counts(from_history, from_proxy);
    });
}
int fdcm_selftest_sweep_steals(fdcm_featuremap* fm, int64_t* count) {
    return guarded([&] {
        require(fm && count, "null argument");
        *count = 0;
        if (!fm->sweep_steals) return;
        finish_build(fm);
        FDCM_HIP(hipSetDevice(fm->device));
        int v = 0;
        FDCM_HIP(hipMemcpy(&v, fm->sweep_steals, sizeof(int), hipMemcpyDeviceToHost));
        *count = v;
    });
}
int fdcm_selftest_sweep_ranges(int n_seeded_columns) { return fdcm::sweep_ranges(n_seeded_columns, fdcm::sweep_min_cols()); }

}  // extern "C"
