// fdcm_search.hip -- search<DefaultMatch> + DefaultSearch + DefaultOptimize/BatchOptimize on gfx950
// (defaultmatch.cpp:32-89, defaultsearch.cpp:29-49, batchoptimize.cpp:6-123, dt3cpu.cpp:119-179).
//
// One wavefront per aligned candidate (template t, template line j, scene line i, alignment).
// The wave builds the candidate itself -- align(), transform(), orientation bins with the glibc
// atanf restatement, bounding box, rasterizeVector, minmaxTranslation -- keeps the aligned lines in
// LDS, and then replays the optimiser's batches: lane b scores translation multiplier k0 + b as
// sum_i |I[bin_i](p1_i + t) - I[bin_i](p2_i + t)| with two 4-byte gathers per line from the DT3
// volume, added in Eigen's VectorXf::sum() order so the float32 score -- and therefore every
// early-exit decision -- is the reference's bit for bit.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <numeric>

#include "fdcm_internal.h"

namespace fdcm {

struct SearchParams {
    // feature map
    const float* vol;
    const float* keys;
    int W, H, m;
    float tx, ty;
    // templates
    const float* tlines;        // 4 floats per line
    const long long* toffsets;  // T+1
    const float* tlengths;      // per line
    const int* tsorted;         // per template: local line indices by descending length
    int T;
    // scene
    const float* slines;
    const float* s_sorted_len;
    const int* s_sorted_idx;
    int n_s;
    // strategy
    int maxT, maxS, window;
    int optimizer;
    long long batch;
    int base;
    // candidates
    const long long* cand_offsets;  // T+1
    long long ncand;
    int lds_lines;  // capacity (lines) of the per-wave LDS area
    // outputs
    fdcm_match* records;
    int* flags;
    unsigned long long* counters;  // [0] translations evaluated, [1] volume reads
};

__device__ __forceinline__ float wave_min_f(float v) {
    for (int d = 32; d >= 1; d >>= 1) v = std_min(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
    for (int d = 32; d >= 1; d >>= 1) v = std_max(v, __shfl_xor(v, d));
    return v;
}

// evaluate<Dt3Cpu> for one translation, dt3cpu.cpp:153-175.  L = per-wave LDS lines
// (x1,y1,x2,y2,bin), off = sceneTranslation + translation.
__device__ __forceinline__ float line_value(const float* __restrict__ vol, const float* L, int i, float offx,
                                            float offy, size_t W, size_t H) {
    const float* l = L + 5 * i;
    const int x1 = (int)(l[0] + offx), y1 = (int)(l[1] + offy);  // translate then cast<int>()
    const int x2 = (int)(l[2] + offx), y2 = (int)(l[3] + offy);
    const size_t sb = (size_t)__float_as_int(l[4]) * W;
    const float a = vol[(sb + (size_t)x1) * H + (size_t)y1];
    const float b = vol[(sb + (size_t)x2) * H + (size_t)y2];
    return f_abs(a - b);
}

// score_per_line.sum(): Eigen 3.4.0 redux (Redux.h, LinearVectorizedTraversal, Packet4f):
// two packet accumulators over blocks of 8, an optional trailing packet, predux as
// (p0+p2)+(p1+p3), then the scalar tail in order.
__device__ __forceinline__ float score_translation(const float* __restrict__ vol, const float* L, int n, float offx,
                                                   float offy, size_t W, size_t H) {
    if (n == 0) return 0.f;
    const int aligned2 = (n / 8) * 8, aligned = (n / 4) * 4;
    float res;
    if (aligned) {
        float p0[4], p1[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) p0[l] = line_value(vol, L, l, offx, offy, W, H);
        if (aligned > 4) {
#pragma unroll
            for (int l = 0; l < 4; ++l) p1[l] = line_value(vol, L, 4 + l, offx, offy, W, H);
            for (int idx = 8; idx < aligned2; idx += 8) {
                float a[8];
#pragma unroll
                for (int l = 0; l < 8; ++l) a[l] = line_value(vol, L, idx + l, offx, offy, W, H);
#pragma unroll
                for (int l = 0; l < 4; ++l) { p0[l] = p0[l] + a[l]; p1[l] = p1[l] + a[4 + l]; }
            }
#pragma unroll
            for (int l = 0; l < 4; ++l) p0[l] = p0[l] + p1[l];
            if (aligned > aligned2) {
#pragma unroll
                for (int l = 0; l < 4; ++l) p0[l] = p0[l] + line_value(vol, L, aligned2 + l, offx, offy, W, H);
            }
        }
        res = (p0[0] + p0[2]) + (p0[1] + p0[3]);
        for (int idx = aligned; idx < n; ++idx) res = res + line_value(vol, L, idx, offx, offy, W, H);
    } else {
        res = line_value(vol, L, 0, offx, offy, W, H);
        for (int idx = 1; idx < n; ++idx) res = res + line_value(vol, L, idx, offx, offy, W, H);
    }
    return res;
}

__global__ void __launch_bounds__(256) k_search(const SearchParams P) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long cand = (long long)blockIdx.x * (blockDim.x >> 6) + wave;
    if (cand >= P.ncand) return;  // wave-uniform
    float* L = lds + (size_t)wave * P.lds_lines * 5;

    // ---- which candidate: template t, sorted template line j, window slot wi, alignment flip
    int lo = 0, hi = P.T;  // last t with cand_offsets[t] <= cand
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (P.cand_offsets[mid] <= cand) lo = mid; else hi = mid;
    }
    const int t = lo;
    const int local = (int)(cand - P.cand_offsets[t]);
    const int flip = local & 1, pair = local >> 1;
    const int j = pair / P.window, wi = pair - j * P.window;
    const long long l0 = P.toffsets[t];
    const int n_t = (int)(P.toffsets[t + 1] - l0);
    // establishSearchStrategy<DefaultSearch>, defaultsearch.cpp:38-46
    const int tl_local = P.tsorted[l0 + j];
    const float tlen = P.tlengths[l0 + tl_local];
    const int centre = binary_search_greater(P.s_sorted_len, P.n_s, tlen);
    int rb, re;
    centered_range(centre, P.n_s, P.maxS, rb, re);
    const int scene_idx = P.s_sorted_idx[rb + wi];
    float tl[4], sl[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { tl[c] = P.tlines[(l0 + tl_local) * 4 + c]; sl[c] = P.slines[(size_t)scene_idx * 4 + c]; }
    // align + transform, defaultmatch.cpp:59-67
    float T1[6], T2[6], T[6];
    align_pair(tl, sl, T1, T2);
#pragma unroll
    for (int c = 0; c < 6; ++c) T[c] = flip ? T2[c] : T1[c];
    float mnx = f_inf(), mny = f_inf(), mxx = -f_inf(), mxy = -f_inf();
    for (int i = lane; i < n_t; i += 64) {
        const float* p = P.tlines + (l0 + i) * 4;
        const float x1 = (T[0] * p[0] + T[1] * p[1]) + T[2], y1 = (T[3] * p[0] + T[4] * p[1]) + T[5];
        const float x2 = (T[0] * p[2] + T[1] * p[3]) + T[2], y2 = (T[3] * p[2] + T[4] * p[3]) + T[5];
        const float angle = atanf_glibc((y2 - y1) / (x2 - x1));  // getAngle, math.h:295-299
        const int bin = closest_orientation(P.keys, P.m, angle);  // dt3cpu.cpp:144-148
        float* d = L + 5 * i;
        d[0] = x1; d[1] = y1; d[2] = x2; d[3] = y2; d[4] = __int_as_float(bin);
        mnx = std_min(mnx, std_min(x1, x2)); mxx = std_max(mxx, std_max(x1, x2));
        mny = std_min(mny, std_min(y1, y2)); mxy = std_max(mxy, std_max(y1, y2));
    }
    mnx = wave_min_f(mnx); mny = wave_min_f(mny); mxx = wave_max_f(mxx); mxy = wave_max_f(mxy);
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes of the wave are done (single wave owns L)
    __builtin_amdgcn_wave_barrier();

    // ---- optimize<BatchOptimize / DefaultOptimize> for this candidate
    bool valid = true;
    // align_vec = normalize(scene_line), defaultmatch.cpp:59
    float ax = sl[2] - sl[0], ay = sl[3] - sl[1];
    {
        const float nrm = sqrtf(ax * ax + ay * ay);
        ax = ax / nrm; ay = ay / nrm;
    }
    if (relatively_equal(f_abs(ax) + f_abs(ay), 0.f)) valid = false;  // batchoptimize.cpp:20-23
    float savx = 0.f, savy = 0.f, min_mul = 0.f, max_mul = 0.f;
    if (valid) {
        rasterize_vector(ax, ay, savx, savy);  // :26
        minmax_translation(mnx, mny, mxx, mxy, savx, savy, (float)P.W, (float)P.H, P.tx, P.ty, min_mul, max_mul);  // :27
        if (!f_isfinite(min_mul) || !f_isfinite(max_mul)) valid = false;  // :30-33
    }
    float best = 0.f;
    long long best_k = 0;
    unsigned long long n_eval = 0;
    if (valid) {
        const size_t W = (size_t)P.W, H = (size_t)P.H;
        // initial score at translation (0,0), :36
        float init = 0.f;
        if (lane == 0) init = score_translation(P.vol, L, n_t, P.tx + 0.f, P.ty + 0.f, W, H);
        init = __shfl(init, 0);
        n_eval += 1;
        best = init;
        float back = init;  // scores.back(): NOT reset between the two directions (batchoptimize.cpp:73)
        const long long B = P.optimizer == FDCM_BATCH_OPTIMIZE ? P.batch : 1;
        for (int dir = 1; dir >= -1; dir -= 2) {
            const long long lim = dir > 0 ? (long long)max_mul : (long long)min_mul;  // static_cast<long>
            for (long long k0 = dir; dir > 0 ? k0 <= lim : k0 >= lim; k0 += dir * B) {
                // batch = k0, k0+dir, ... limited by B entries and by lim
                long long nb = dir > 0 ? (lim - k0 + 1) : (k0 - lim + 1);
                if (nb > B) nb = B;
                // first argmin and last element over the batch, in chunks of 64 lanes
                float bmin = 0.f, blast = 0.f;
                long long bmin_k = 0;
                for (long long c0 = 0; c0 < nb; c0 += 64) {
                    const long long kk = k0 + dir * (c0 + lane);
                    const bool act = (c0 + lane) < nb;
                    float sc = f_inf();
                    if (act) {
                        const float trx = (float)kk * savx, try_ = (float)kk * savy;  // :58 / :81
                        sc = score_translation(P.vol, L, n_t, P.tx + trx, P.ty + try_, W, H);
                    }
                    const int nact = (int)((nb - c0) < 64 ? (nb - c0) : 64);
                    // std::min_element: first minimum
                    float m = sc;
                    for (int d = 32; d >= 1; d >>= 1) m = std_min(m, __shfl_xor(m, d));
                    const unsigned long long eq = __ballot(act && sc == m);
                    const int arg = eq ? (__ffsll((long long)eq) - 1) : 0;
                    const float cmin = __shfl(sc, arg);
                    if (c0 == 0 || cmin < bmin) { bmin = cmin; bmin_k = k0 + dir * (c0 + arg); }
                    blast = __shfl(sc, nact - 1);
                }
                n_eval += (unsigned long long)nb;
                if (bmin > back) break;                 // :65 / :88
                back = bmin;                            // keep (translation, score)
                if (bmin < best) { best = bmin; best_k = bmin_k; }  // first argmin over kept scores, :97
                if (P.optimizer == FDCM_BATCH_OPTIMIZE && bmin < blast) break;  // :70 / :93
            }
        }
    }
    if (lane == 0) {
        fdcm_match r;
        r.tmpl_idx = P.base + t;
        r.score = best;
        // combine(translation, transform), math.h:427-432; translation = float(k) * scaled_align_vec
        // (the kept translation for k = 0 is the literal Point2{0,0} of batchoptimize.cpp:47)
        const float trx = best_k == 0 ? 0.f : (float)best_k * savx, try_ = best_k == 0 ? 0.f : (float)best_k * savy;
        r.transform[0] = T[0]; r.transform[1] = T[1]; r.transform[2] = T[2] + trx;
        r.transform[3] = T[3]; r.transform[4] = T[4]; r.transform[5] = T[5] + try_;
        if (valid) P.records[cand] = r;
        P.flags[cand] = valid ? 1 : 0;
        if (n_eval) {
            atomicAdd(&P.counters[0], n_eval);
            atomicAdd(&P.counters[1], n_eval * 2ull * (unsigned long long)n_t);
        }
    }
}

// Positional compaction of the valid records (defaultmatch.cpp:76-86): one block walks the
// candidate list in chunks of 1024 with a running offset.
__global__ void __launch_bounds__(1024) k_compact(const fdcm_match* __restrict__ records, const int* __restrict__ flags,
                                                  long long n, fdcm_match* __restrict__ out,
                                                  unsigned long long* __restrict__ counters) {
    __shared__ int wsum[16];
    __shared__ long long running;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) running = 0;
    __syncthreads();
    for (long long c0 = 0; c0 < n; c0 += 1024) {
        const long long i = c0 + tid;
        const int f = i < n ? flags[i] : 0;
        int incl = f;
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        const long long base = running;
        if (f) out[base + wbase + incl - 1] = records[i];
        __syncthreads();
        if (tid == 1023) running = base + wbase + incl;
        __syncthreads();
    }
    if (tid == 0) counters[2] = (unsigned long long)running;
}

int64_t search_capacity(const fdcm_templates* t, int64_t n_scene, int64_t maxT, int64_t maxS) {
    int64_t total = 0;
    const int64_t window = std::min<int64_t>(maxS, n_scene);
    for (int64_t i = 0; i < t->T; ++i) {
        const int64_t nt = t->offsets[i + 1] - t->offsets[i];
        total += 2 * std::min<int64_t>(nt, maxT) * window;
    }
    return total;
}

void run_search(fdcm_featuremap* fm, const fdcm_templates* t, const float* scene, int64_t n_scene, int64_t maxT,
                int64_t maxS, int optimizer, int64_t batch, int32_t base, fdcm_match* out_device, int64_t* n_out) {
    const auto t0 = std::chrono::steady_clock::now();
    *n_out = 0;
    fm->last_search = fdcm_search_timing{};
    // early-outs of search<DefaultMatch>, defaultmatch.cpp:40-41
    if (t->T == 0 || n_scene == 0 || (fm->W == 0 && fm->H == 0)) return;
    FDCM_HIP(hipSetDevice(fm->device));
    if (!fm->stream) FDCM_HIP(hipStreamCreateWithFlags(&fm->stream, hipStreamNonBlocking));
    if (!fm->timing.created) {
        for (auto& e : fm->timing.ev) FDCM_HIP(hipEventCreate(&e));
        fm->timing.created = true;
    }
    hipStream_t st = fm->stream;
    const int n_s = (int)n_scene;
    const int window = (int)std::min<int64_t>(maxS, n_scene);
    // ---- scene side of establishSearchStrategy (defaultsearch.cpp:32-36): lengths, argsort by
    // descending length with std::sort (same comparator and index type as the reference)
    std::vector<float> slen((size_t)n_s);
    for (int i = 0; i < n_s; ++i) {
        const float dx = scene[4 * i + 2] - scene[4 * i], dy = scene[4 * i + 3] - scene[4 * i + 1];
        slen[i] = std::sqrt(dx * dx + dy * dy);
    }
    std::vector<long> sidx((size_t)n_s);
    std::iota(sidx.begin(), sidx.end(), 0);
    std::sort(sidx.begin(), sidx.end(), [&slen](long const i1, long const i2) { return slen[i1] > slen[i2]; });
    // candidate offsets per template
    std::vector<long long> coff((size_t)t->T + 1, 0);
    for (int64_t i = 0; i < t->T; ++i) {
        const int64_t nt = t->offsets[i + 1] - t->offsets[i];
        coff[i + 1] = coff[i] + 2 * std::min<int64_t>(nt, maxT) * window;
    }
    const long long ncand = coff[t->T];
    fm->last_search.candidates = ncand;
    if (ncand == 0) return;
    // ---- stage + upload: scene lines | sorted lengths | sorted idx | candidate offsets
    auto align16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_lines = 0, o_len = align16((size_t)n_s * 16), o_idx = o_len + align16((size_t)n_s * 4),
                 o_coff = o_idx + align16((size_t)n_s * 4), blob = o_coff + align16(coff.size() * 8);
    fm->s_stage.reserve(blob);
    fm->s_scene.reserve(blob);
    char* hs = (char*)fm->s_stage.p;
    std::memcpy(hs + o_lines, scene, (size_t)n_s * 16);
    float* hl = (float*)(hs + o_len);
    int* hi = (int*)(hs + o_idx);
    for (int i = 0; i < n_s; ++i) { hl[i] = slen[sidx[i]]; hi[i] = (int)sidx[i]; }
    std::memcpy(hs + o_coff, coff.data(), coff.size() * 8);
    FDCM_HIP(hipMemcpyAsync(fm->s_scene.p, hs, blob, hipMemcpyHostToDevice, st));
    fm->s_records.reserve((size_t)ncand * sizeof(fdcm_match));
    fm->s_flags.reserve((size_t)ncand * sizeof(int));
    fm->s_counter.reserve(64);
    FDCM_HIP(hipMemsetAsync(fm->s_counter.p, 0, 64, st));
    // keys live at the end of the build plan blob; for adopted volumes they are uploaded there too
    SearchParams P{};
    P.vol = fm->vol.as<float>();
    P.keys = (const float*)((const char*)fm->plan.p + fm->off_keys);
    P.W = (int)fm->W; P.H = (int)fm->H; P.m = (int)fm->m; P.tx = fm->tx; P.ty = fm->ty;
    P.tlines = t->d_lines.as<float>();
    P.toffsets = t->d_offsets.as<long long>();
    P.tlengths = t->d_lengths.as<float>();
    P.tsorted = t->d_sorted.as<int>();
    P.T = (int)t->T;
    const char* ds = (const char*)fm->s_scene.p;
    P.slines = (const float*)(ds + o_lines);
    P.s_sorted_len = (const float*)(ds + o_len);
    P.s_sorted_idx = (const int*)(ds + o_idx);
    P.n_s = n_s;
    P.maxT = (int)maxT; P.maxS = (int)maxS; P.window = window;
    P.optimizer = optimizer; P.batch = batch < 1 ? 1 : batch; P.base = base;
    P.cand_offsets = (const long long*)(ds + o_coff);
    P.ncand = ncand;
    P.lds_lines = (int)std::max<int64_t>(1, t->max_lines);
    P.records = fm->s_records.as<fdcm_match>();
    P.flags = fm->s_flags.as<int>();
    P.counters = fm->s_counter.as<unsigned long long>();
    const size_t lds = (size_t)4 * P.lds_lines * 5 * sizeof(float);
    if (lds > 160 * 1024) throw std::string("template with too many lines for the per-wave LDS area");
    if (lds > 64 * 1024)
        FDCM_HIP(hipFuncSetAttribute((const void*)k_search, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t* ev = fm->timing.ev;
    FDCM_HIP(hipEventRecord(ev[6], st));
    hipLaunchKernelGGL(k_search, dim3((unsigned)((ncand + 3) / 4)), dim3(256), lds, st, P);
    fdcm_match* dst = out_device;
    if (!dst) {
        fm->s_out.reserve((size_t)ncand * sizeof(fdcm_match));
        dst = fm->s_out.as<fdcm_match>();
    }
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, st, P.records, P.flags, ncand, dst, P.counters);
    FDCM_HIP(hipEventRecord(ev[7], st));
    FDCM_HIP(hipGetLastError());
    unsigned long long hc[3] = {0, 0, 0};
    FDCM_HIP(hipMemcpyAsync(hc, fm->s_counter.p, sizeof hc, hipMemcpyDeviceToHost, st));
    FDCM_HIP(hipStreamSynchronize(st));
    *n_out = (int64_t)hc[2];
    fm->last_search.evaluations = (int64_t)hc[0];
    FDCM_HIP(hipEventElapsedTime(&fm->last_search.kernel_ms, ev[6], ev[7]));
    fm->last_search.total_ms =
        std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace fdcm
