// fdcm_seam.hip -- the reference's feature-map plug-in seam on the HBM volume: the functions a type deriving from
// FeatureMapInstance must specialise besides getFeatureSize (featuremap.h:27-52), i.e. what FeatureMapModel<T>
// forwards to (featuremap.h:80-92) and what every optimiser calls (defaultoptimize.cpp:26,49-64,
// batchoptimize.cpp:27,58-62):
//   minmaxTranslation<Dt3Cpu>  dt3cpu.cpp:30-75,119-124   -> k_minmax    (one wave per template)
//   evaluate<Dt3Cpu>           dt3cpu.cpp:126-179         -> k_evaluate  (one wave per template and 32 translations)
// Both are batched over templates: one launch per call.  The scores are added in Eigen's order by the same
// pair_score() the search kernel uses (fdcm_score.h), so they are the reference's bits.
#include <cmath>
#include <cstring>

#include "fdcm_internal.h"
#include "fdcm_score.h"

namespace fdcm {

// minmaxPoint (math.h:166-171) of every template + detail::minmaxTranslation (dt3cpu.cpp:30-75) with the feature
// map's size and scene translation.  align: one vector per template.
__global__ void __launch_bounds__(256) k_minmax(const float* __restrict__ lines, const long long* __restrict__ offsets, int T,
                                                const float2* __restrict__ align, float W, float H, float tx, float ty,
                                                float2* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;  // wave-uniform
    const long long l0 = offsets[t];
    const int n = (int)(offsets[t + 1] - l0);
    float mnx = f_inf(), mny = f_inf(), mxx = -f_inf(), mxy = -f_inf();
    for (int i = lane; i < n; i += 64) {
        const float* p = lines + (l0 + i) * 4;
        mnx = std_min(mnx, std_min(p[0], p[2])); mxx = std_max(mxx, std_max(p[0], p[2]));
        mny = std_min(mny, std_min(p[1], p[3])); mxy = std_max(mxy, std_max(p[1], p[3]));
    }
    mnx = wave_min_f(mnx); mny = wave_min_f(mny); mxx = wave_max_f(mxx); mxy = wave_max_f(mxy);
    if (lane == 0) {
        float lo, hi;
        minmax_translation(mnx, mny, mxx, mxy, align[t].x, align[t].y, W, H, tx, ty, lo, hi);
        out[t] = make_float2(lo, hi);
    }
}

struct EvalItem { int t, first, count, pad; };  // translations [first, first + count) of template t; count <= 32

// evaluate<Dt3Cpu>, dt3cpu.cpp:126-179.  lines5: x1, y1, x2, y2 and the orientation bin of every template line (the
// bins come from the host: closestOrientation with the host libm, dt3cpu.cpp:144-148, as in the reference).  The
// reference reads the image unchecked (the optimisers stay inside it through minmaxTranslation); here a translation
// that puts an end point outside the image scores NaN instead of reading out of bounds.
__global__ void __launch_bounds__(256) k_evaluate(const float* __restrict__ vol, int W, int H, float tx, float ty,
                                                  const float* __restrict__ lines5, const long long* __restrict__ offsets,
                                                  const float2* __restrict__ trans, const EvalItem* __restrict__ items,
                                                  int n_items, int lds_lines, float* __restrict__ scores) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int it = blockIdx.x * 4 + wave;
    if (it >= n_items) return;  // wave-uniform; waves never synchronise with each other
    const EvalItem e = items[it];
    float* L = lds + (size_t)wave * 5 * lds_lines;
    const long long l0 = offsets[e.t];
    const int n = (int)(offsets[e.t + 1] - l0);
    for (int i = lane; i < 5 * n; i += 64) L[i] = lines5[l0 * 5 + i];
    const int h = lane >> 5, slot = lane & 31;
    const bool act = slot < e.count;
    // translate(tmpl, sceneTranslation + translation), dt3cpu.cpp:153
    const float2 tr = act ? trans[e.first + slot] : make_float2(0.f, 0.f);
    const float offx = tx + tr.x, offy = ty + tr.y;
    bool inside = true;
    for (int i = h; i < n; i += 2) {  // the two lanes of a translation check alternate lines
        const float* l = L + 5 * i;
        const float px[2] = {l[0] + offx, l[2] + offx}, py[2] = {l[1] + offy, l[3] + offy};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            // cast<int>() of a value outside int's range is undefined in the reference; here it counts as outside
            const bool fin = px[c] > -1.f && px[c] < (float)W && py[c] > -1.f && py[c] < (float)H;  // false for NaN
            const int x = fin ? (int)px[c] : -1, y = fin ? (int)py[c] : -1;
            inside = inside && x >= 0 && x < W && y >= 0 && y < H;
        }
    }
    inside = inside && __shfl_xor((int)inside, 32) != 0;
    // (64-bit addresses: the seam serves volumes of any size and is not the hot path; L[5 i + 4] holds the bin)
    const VolRef V = make_volref(vol, ivol_slice_floats(W, H), 0, false);
    const float s = pair_score<false>(V, L, n, offx, offy, (unsigned)H, h, act && inside);
    if (act && h == 0) scores[e.first + slot] = inside ? s : f_nan();
}

void run_minmax(fdcm_featuremap* fm, const float* lines, const int64_t* offsets, int64_t T, const float* align, float* out) {
    if (T == 0) return;
    std::lock_guard<std::mutex> turn(fm->seam_mutex);  // concurrent callers of one feature map (a pool's tasks) take turns
    FDCM_HIP(hipSetDevice(fm->device));
    if (!fm->stream) FDCM_HIP(hipStreamCreateWithFlags(&fm->stream, hipStreamNonBlocking));
    hipStream_t st = fm->stream;
    const int64_t n_lines = offsets[T];
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_lines = 0, o_off = al((size_t)n_lines * 16), o_al = o_off + al((size_t)(T + 1) * 8), o_out = o_al + al((size_t)T * 8),
                 total = o_out + al((size_t)T * 8);
    fm->s_eval.reserve(total);
    char* d = (char*)fm->s_eval.p;
    if (n_lines) FDCM_HIP(hipMemcpyAsync(d + o_lines, lines, (size_t)n_lines * 16, hipMemcpyHostToDevice, st));
    FDCM_HIP(hipMemcpyAsync(d + o_off, offsets, (size_t)(T + 1) * 8, hipMemcpyHostToDevice, st));
    FDCM_HIP(hipMemcpyAsync(d + o_al, align, (size_t)T * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_minmax, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, st, (const float*)(d + o_lines),
                       (const long long*)(d + o_off), (int)T, (const float2*)(d + o_al), (float)fm->W, (float)fm->H, fm->tx, fm->ty,
                       (float2*)(d + o_out));
    FDCM_HIP(hipGetLastError());
    FDCM_HIP(hipMemcpyAsync(out, d + o_out, (size_t)T * 8, hipMemcpyDeviceToHost, st));
    FDCM_HIP(hipStreamSynchronize(st));
}

void run_evaluate(fdcm_featuremap* fm, const float* lines, const int64_t* offsets, int64_t T, const float* translations,
                  const int64_t* tr_offsets, float* scores) {
    if (T == 0 || tr_offsets[T] == 0) return;
    if (fm->vol_stage != 3) throw std::string("the feature map holds a partial build (no line integral): nothing to evaluate");
    if (fm->m == 0 || fm->W == 0 || fm->H == 0) throw std::string("evaluate on an empty feature map");
    std::lock_guard<std::mutex> turn(fm->seam_mutex);  // concurrent callers of one feature map (a pool's tasks) take turns
    finish_build(fm);
    FDCM_HIP(hipSetDevice(fm->device));
    if (!fm->stream) FDCM_HIP(hipStreamCreateWithFlags(&fm->stream, hipStreamNonBlocking));
    hipStream_t st = fm->stream;
    const int64_t n_lines = offsets[T], n_tr = tr_offsets[T];
    if (n_tr > 0x7fffffffll || n_lines > 0x7fffffffll) throw std::string("too many translations or lines in one evaluate call");
    // closestOrientation per template line, once per call (dt3cpu.cpp:144-148), with the host libm like the reference
    std::vector<float> l5((size_t)n_lines * 5);
    int64_t max_lines = 1;
    for (int64_t i = 0; i < n_lines; ++i) {
        const float* p = lines + 4 * i;
        const float angle = std::atan((p[3] - p[1]) / (p[2] - p[0]));  // getAngle, math.h:295-299
        const int bin = closest_orientation(fm->keys.data(), (int)fm->m, angle);
        float* q = &l5[(size_t)i * 5];
        q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; q[3] = p[3];
        std::memcpy(q + 4, &bin, 4);
    }
    std::vector<EvalItem> items;
    for (int64_t t = 0; t < T; ++t) {
        max_lines = std::max(max_lines, offsets[t + 1] - offsets[t]);
        for (int64_t f = tr_offsets[t]; f < tr_offsets[t + 1]; f += 32)
            items.push_back(EvalItem{(int)t, (int)f, (int)std::min<int64_t>(32, tr_offsets[t + 1] - f), 0});
    }
    const size_t lds = (size_t)4 * 5 * (size_t)max_lines * sizeof(float);
    if (lds > 160 * 1024) throw std::string("a template has too many lines for the evaluate kernel's LDS staging (2048 at most)");
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_l5 = 0, o_off = al(l5.size() * 4), o_tr = o_off + al((size_t)(T + 1) * 8), o_it = o_tr + al((size_t)n_tr * 8),
                 o_sc = o_it + al(items.size() * sizeof(EvalItem)), total = o_sc + al((size_t)n_tr * 4);
    fm->s_eval.reserve(total);
    char* d = (char*)fm->s_eval.p;
    if (n_lines) FDCM_HIP(hipMemcpyAsync(d + o_l5, l5.data(), l5.size() * 4, hipMemcpyHostToDevice, st));
    FDCM_HIP(hipMemcpyAsync(d + o_off, offsets, (size_t)(T + 1) * 8, hipMemcpyHostToDevice, st));
    FDCM_HIP(hipMemcpyAsync(d + o_tr, translations, (size_t)n_tr * 8, hipMemcpyHostToDevice, st));
    FDCM_HIP(hipMemcpyAsync(d + o_it, items.data(), items.size() * sizeof(EvalItem), hipMemcpyHostToDevice, st));
    if (lds > 64 * 1024)
        FDCM_HIP(hipFuncSetAttribute((const void*)k_evaluate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_evaluate, dim3((unsigned)((items.size() + 3) / 4)), dim3(256), lds, st, (const float*)fm->vol.as<float>(),
                       (int)fm->W, (int)fm->H, fm->tx, fm->ty, (const float*)(d + o_l5), (const long long*)(d + o_off),
                       (const float2*)(d + o_tr), (const EvalItem*)(d + o_it), (int)items.size(), (int)max_lines, (float*)(d + o_sc));
    FDCM_HIP(hipGetLastError());
    FDCM_HIP(hipMemcpyAsync(scores, d + o_sc, (size_t)n_tr * 4, hipMemcpyDeviceToHost, st));
    FDCM_HIP(hipStreamSynchronize(st));  // (the host vectors above stay alive until here)
}

}  // namespace fdcm
