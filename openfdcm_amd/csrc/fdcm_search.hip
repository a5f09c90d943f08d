// fdcm_search.hip -- search<DefaultMatch> + DefaultSearch + DefaultOptimize/BatchOptimize on gfx950
// (defaultmatch.cpp:32-89, defaultsearch.cpp:29-49, batchoptimize.cpp:6-123, dt3cpu.cpp:119-179).
//
// k_pairs resolves DefaultSearch once per (template, template line): binary search in the sorted
// scene lengths + window -> {template line, scene line} pairs.  k_search then runs one wavefront
// per aligned candidate (template t, pair, alignment) with no workgroup-level synchronisation:
// each wave has a private LDS area (keys | aligned lines, 5 floats per line | score window),
// builds its candidate itself -- align(), transform(), orientation bins with the glibc atanf
// restatement, bounding box, rasterizeVector, minmaxTranslation -- and replays the optimiser.
// Results go to a positional record/flag/evaluation-count slot per candidate (no atomics);
// k_chunk_counts + k_scatter compact them in the reference's order -- for a host-output search straight into the
// caller's pinned buffer (no copy command: see run_search).
//
// Scoring: sum_i |I[bin_i](p1_i + t) - I[bin_i](p2_i + t)| with two 4-byte gathers per line from the integrated
// volume, whose interleaved layout (ivol_index) puts 4 x 4 pixels into a 64-byte sector.  Two lanes share one translation (the two packet
// accumulators of Eigen's VectorXf::sum()), so one round scores up to 32 translations with all
// gathers of a lane in flight together.  The first round scores translation 0 and the first
// WIN multipliers of BOTH directions; the reference's exit rule (batches, the un-reset
// scores.back(), first-minimum ties) is then replayed on the scores, and further rounds run only
// for candidates whose descent continues.  Scores are added in Eigen's order, so they -- and
// every decision -- are the reference's bit for bit; translations scored speculatively but never
// reached by the rule are simply not read.
#include <algorithm>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <cstring>
#include <numeric>
#include <thread>

#include "fdcm_internal.h"
#include "fdcm_score.h"

namespace fdcm {

struct SearchParams {
    // feature map
    const float* vol;    // the integrated volume, [k][x/4][y][x%4]
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
    int batch;  // >= 1
    int win;    // multipliers scored per direction and round (multiple of batch when batch <= 15)
    int base;
    // candidates
    const long long* cand_offsets;  // T+1
    const int2* pairs;              // per template, per (j, wi): {template line, scene line}
    int pairs_stride;               // maxT * window
    int bpt;                        // workgroups per template (template-major order, when there is no work list)
    const int2* work;               // valid pairs {template, pair slot} grouped by scene line (or null)
    long long ncand;                // candidates in total
    int nblocks;                    // k_search workgroups
    int xcd_parts;                  // 1: every XCD takes its own contiguous part of the work list
    int lds_lines;                  // capacity (lines) of the LDS template / aligned-line areas
    const unsigned short* host_bins; // [candidate][lds_lines] orientation bins computed with the host libm, or null (see run_search)
    // outputs
    fdcm_match* records;
    int* flags;
    int* evals;
    unsigned long long* counters;  // [0] translations evaluated by the rule, [1] unused, [2] matches
#ifdef FDCM_LAB
    unsigned long long* lab;  // 8 stamps per candidate (make LAB=1, FDCM_SEARCH_LAB=1)
#endif
};

#ifndef FDCM_SEARCH_WPB
#define FDCM_SEARCH_WPB 4
#endif
static constexpr int kWavesPerBlock = FDCM_SEARCH_WPB;

struct OptState {
    VolRef V;        // the integrated volume
    const float* L;  // aligned lines of the candidate (LDS)
    float* sc;       // score window (LDS): [0, WIN) positive, [WIN, 2 WIN) negative, [2 WIN] translation 0
    int n_t, lane, B, WIN;
    bool batch_rule;
    bool reset_back;  // IndulgentOptimize: the negative direction compares against the initial score again
    unsigned H;
    float tx, ty, savx, savy;
    int lim_p, lim_n;  // multiplier limits: 32 bits are enough, see k_search
#ifdef FDCM_LAB
    unsigned long long* lab;
#endif
};
#ifdef FDCM_LAB
#define SEARCH_STAMP(ptr, i) do { if ((ptr) && (threadIdx.x & 63) == 0) (ptr)[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SEARCH_STAMP(ptr, i) do {} while (0)
#endif

// Score multipliers k_from, k_from + dir, ... (cnt of them) into sc[dst ..]; with_zero additionally
// scores translation (0,0) into sc[2 WIN].  32 translations per gather round.
template <bool BUF32>
__device__ __forceinline__ void score_range(const OptState& o, int dir, int k_from, int cnt, int dst, bool with_zero) {
    const int h = o.lane >> 5, slot = o.lane & 31;  // neighbouring lanes = neighbouring translations
    const int total = cnt + (with_zero ? 1 : 0);
    for (int s0 = 0; s0 < total; s0 += 32) {
        const int idx = s0 + slot - (with_zero ? 1 : 0);  // -1 = the zero translation
        const bool act = idx < cnt;
        const int k = idx < 0 ? 0 : k_from + dir * idx;
        // translation = float(k) * scaled_align_vec (:58/:81); Point2{0,0} for the initial score (:36)
        const float trx = idx < 0 ? 0.f : (float)k * o.savx, try_ = idx < 0 ? 0.f : (float)k * o.savy;
        const float s = pair_score<BUF32>(o.V, o.L, o.n_t, o.tx + trx, o.ty + try_, o.H, h, act);
        if (act && h == 0) o.sc[idx < 0 ? 2 * o.WIN : dst + idx] = s;
    }
}

// optimize<BatchOptimize / DefaultOptimize / IndulgentOptimize> for one candidate (batchoptimize.cpp:36-98,
// defaultoptimize.cpp:36-66, indulgentoptimize.cpp:33-77): speculative scoring in windows, literal replay of
// the rule.  IndulgentOptimize walks like DefaultOptimize (a passed-through score is scored again at the
// same multiplier until the allowance is used up, then the walk breaks) but starts the negative direction
// from the initial score again.
template <bool BUF32>
__device__ __forceinline__ void optimise(const OptState& o, float& best, int& best_k, int& n_eval) {
    const int WIN = o.WIN, B = o.B;
    const int h = o.lane >> 5, slot = o.lane & 31;  // neighbouring lanes = neighbouring translations
    // ---- round 1: translation 0 and the first WIN multipliers of both directions
    const int have_p = min(WIN, o.lim_p >= 1 ? o.lim_p : 0);
    const int have_n = min(WIN, o.lim_n <= -1 ? -o.lim_n : 0);
    if (1 + have_p + have_n <= 32) {
        // one gather round: slot 0 = zero, then positives, then negatives
        const int idx = slot - 1;
        const bool is_p = idx >= 0 && idx < have_p, is_n = idx >= have_p && idx < have_p + have_n;
        const bool act = slot == 0 || is_p || is_n;
        const int k = is_p ? 1 + idx : (is_n ? -1 - (idx - have_p) : 0);
        const float trx = slot == 0 ? 0.f : (float)k * o.savx, try_ = slot == 0 ? 0.f : (float)k * o.savy;
        const float s = pair_score<BUF32>(o.V, o.L, o.n_t, o.tx + trx, o.ty + try_, o.H, h, act);
        if (act && h == 0) o.sc[slot == 0 ? 2 * WIN : (is_p ? idx : WIN + (idx - have_p))] = s;
    } else {
        score_range<BUF32>(o, +1, 1, have_p, 0, true);
        score_range<BUF32>(o, -1, -1, have_n, WIN, false);
    }
    const float init = o.sc[2 * WIN];
    SEARCH_STAMP(o.lab, 3);
    n_eval += 1;
    best = init;
    float back = init;  // scores.back(): NOT reset between the two directions (batchoptimize.cpp:73)
    for (int dir = 1; dir >= -1; dir -= 2) {
        if (dir < 0 && o.reset_back) back = init;  // indulgentoptimize.cpp:59-63
        const int lim = dir > 0 ? o.lim_p : o.lim_n;
        const int off = dir > 0 ? 0 : WIN;
        int win0 = dir;                             // multiplier held in sc[off]
        int have = dir > 0 ? have_p : have_n;       // multipliers available from win0 on
        for (int k0 = dir; dir > 0 ? k0 <= lim : k0 >= lim; k0 += dir * B) {
            int nb = dir > 0 ? (lim - k0 + 1) : (k0 - lim + 1);
            if (nb > B) nb = B;
            // std::min_element over the batch (first minimum) and its last element, :63-70 / :86-93
            float bmin = 0.f, blast = 0.f;
            int bmin_k = 0;
            for (int c0 = 0; c0 < nb;) {
                int rel = dir > 0 ? (k0 + c0 - win0) : (win0 - (k0 - c0));  // index in the window
                if (rel >= have) {  // the rule walks on: score the next window of this direction
                    win0 = k0 + dir * c0;
                    const int left = dir > 0 ? (lim - win0 + 1) : (win0 - lim + 1);
                    have = min(WIN, left);
                    score_range<BUF32>(o, dir, win0, have, off, false);
                    rel = 0;
                }
                const int take = min(nb - c0, have - rel);
                // the piece's scores one per lane: its first minimum is a wave reduction, not a walk through LDS
                for (int e0 = 0; e0 < take; e0 += 64) {
                    const int n = min(64, take - e0);
                    const float s = o.lane < n ? o.sc[off + rel + e0 + o.lane] : f_inf();
                    const float mn = wave_min_f(s);
                    const int first = __ffsll((long long)__ballot(o.lane < n && s == mn)) - 1;
                    if ((c0 == 0 && e0 == 0) || mn < bmin) { bmin = mn; bmin_k = k0 + dir * (c0 + e0 + first); }
                    blast = __shfl(s, n - 1);
                }
                c0 += take;
            }
            n_eval += nb;
            if (bmin > back) break;                 // :65 / :88
            back = bmin;                            // keep (translation, score)
            if (bmin < best) { best = bmin; best_k = bmin_k; }  // first argmin over kept scores, :97
            if (o.batch_rule && bmin < blast) break;  // :70 / :93
        }
    }
}

// establishSearchStrategy<DefaultSearch> (defaultsearch.cpp:38-46) for every (template, j-th longest
// template line): binary search of its length among the scene lengths, centred window of scene lines.
__global__ void k_pairs(const SearchParams P) {
    const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = P.maxT;
    if (gidx >= P.T * per) return;
    const int t = gidx / per, j = gidx - t * per;
    const long long l0 = P.toffsets[t];
    const int n_t = (int)(P.toffsets[t + 1] - l0);
    int2* out = const_cast<int2*>(P.pairs) + (size_t)t * P.pairs_stride + (size_t)j * P.window;
    if (j >= n_t) {  // fewer lines than max_tmpl_lines: no pairs in these slots
        for (int wi = 0; wi < P.window; ++wi) out[wi] = make_int2(-1, -1);
        return;
    }
    const int tl_local = P.tsorted[l0 + j];
    const float tlen = P.tlengths[l0 + tl_local];
    const int centre = binary_search_greater(P.s_sorted_len, P.n_s, tlen);
    int rb, re;
    centered_range(centre, P.n_s, P.maxS, rb, re);
    for (int wi = 0; wi < P.window; ++wi) out[wi] = make_int2(tl_local, P.s_sorted_idx[rb + wi]);
}

// Work list: the valid (template, pair slot) entries grouped by scene line.  A candidate's gathers
// fall inside the template's bounding box laid along its scene line (plus the walk of the
// optimiser), a few MB over all slices, and DefaultSearch sends most candidates to the few scene
// lines whose length is close to the templates' longest lines.  In template-major order consecutive
// waves touch unrelated regions and every L2 keeps missing; grouped by scene line, the waves in
// flight at any time share one or two regions.  Output positions are
// independent of the processing order (k_chunk_counts / k_scatter compact by candidate position).
// One workgroup: counting sort by scene line (hashed into kWorkBins) with LDS atomics; the order
// inside a group is arbitrary.
static constexpr int kWorkBins = 8192;
// bin of a pair: its scene line.  (Splitting the two alignments of a pair into separate groups was
// measured slower: 0.37-0.41 ms against 0.30 ms.)
__device__ __forceinline__ int work_key(int scene_idx) { return scene_idx & (kWorkBins - 1); }
// take a unique rank inside bins[key]: one LDS atomic per lane (same-address lanes serialise inside
// the LDS unit, ~1 per cycle, which is far cheaper than aggregating them with ballots and shuffles)
__device__ __forceinline__ int work_rank(int* bins, int key, bool valid, int lane) {
    (void)lane;
    return valid ? atomicAdd(&bins[key], 1) : 0;
}
// exclusive scan of the bins in place: kWorkBins / 1024 bins per thread + a block scan of the sums
__device__ __forceinline__ void work_scan(int* bins, int* partial, int tid) {
    constexpr int PER = kWorkBins / 1024;
    int loc[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { loc[q] = sum; sum += bins[tid * PER + q]; }
    partial[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = tid >= d ? partial[tid - d] : 0;
        __syncthreads();
        partial[tid] += v;
        __syncthreads();
    }
    const int excl = partial[tid] - sum;
#pragma unroll
    for (int q = 0; q < PER; ++q) bins[tid * PER + q] = excl + loc[q];
}

// Up to 16384 slots: one workgroup, 16 slots per thread kept in registers (all loads in flight at once, no second
// pass over memory).
__global__ void __launch_bounds__(1024) k_worklist(const SearchParams P, long long n_slots, int2* __restrict__ work) {
    __shared__ int bins[kWorkBins];
    __shared__ int partial[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < kWorkBins; i += 1024) bins[i] = 0;
    {
        constexpr int RR = 16;
        int2 pr[RR];
        int rk[RR];
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            const long long i = (long long)r * 1024 + tid;
            pr[r] = i < n_slots ? P.pairs[i] : make_int2(-1, -1);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RR; ++r) rk[r] = work_rank(bins, work_key(pr[r].y), pr[r].x >= 0, lane);
        __syncthreads();
        work_scan(bins, partial, tid);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            if (pr[r].x < 0) continue;
            const long long i = (long long)r * 1024 + tid;
            const int t = (int)(i / P.pairs_stride);
            work[bins[work_key(pr[r].y)] + rk[r]] = make_int2(t, (int)(i - (long long)t * P.pairs_stride));
        }
    }
}

// More than 16384 slots: the same counting sort over several workgroups.  Each block takes 16384 slots:
//   k_wl_count    block-local histogram in LDS, one global atomicAdd per (block, non-empty bin)
//   k_wl_starts   exclusive scan of the global histogram (one block) -> first index of every bin; cursors zeroed
//   k_wl_scatter  block-local ranks again (same loads), one global atomicAdd per (block, non-empty bin) reserves the
//                 block's range inside the bin; the order inside a bin is arbitrary, as in the one-block version
static constexpr int kWlSlotsPerBlock = 16 * 1024;
__global__ void __launch_bounds__(1024) k_wl_count(const SearchParams P, long long n_slots, int* __restrict__ ghist) {
    __shared__ int bins[kWorkBins];
    const int tid = threadIdx.x;
    for (int i = tid; i < kWorkBins; i += 1024) bins[i] = 0;
    __syncthreads();
    const long long b0 = (long long)blockIdx.x * kWlSlotsPerBlock;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long i = b0 + (long long)r * 1024 + tid;
        const int2 pr = i < n_slots ? P.pairs[i] : make_int2(-1, -1);
        if (pr.x >= 0) atomicAdd(&bins[work_key(pr.y)], 1);
    }
    __syncthreads();
    for (int i = tid; i < kWorkBins; i += 1024)
        if (bins[i]) atomicAdd(&ghist[i], bins[i]);
}
__global__ void __launch_bounds__(1024) k_wl_starts(int* __restrict__ ghist, int* __restrict__ cursor) {
    __shared__ int bins[kWorkBins];
    __shared__ int partial[1024];
    const int tid = threadIdx.x;
    for (int i = tid; i < kWorkBins; i += 1024) { bins[i] = ghist[i]; cursor[i] = 0; }
    __syncthreads();
    work_scan(bins, partial, tid);
    __syncthreads();
    for (int i = tid; i < kWorkBins; i += 1024) ghist[i] = bins[i];  // now the first index of every bin
}
__global__ void __launch_bounds__(1024) k_wl_scatter(const SearchParams P, long long n_slots, const int* __restrict__ starts,
                                                     int* __restrict__ cursor, int2* __restrict__ work) {
    __shared__ int bins[kWorkBins];   // local counts, then the block's first index inside each bin
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < kWorkBins; i += 1024) bins[i] = 0;
    __syncthreads();
    const long long b0 = (long long)blockIdx.x * kWlSlotsPerBlock;
    int2 pr[16];
    int rk[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long i = b0 + (long long)r * 1024 + tid;
        pr[r] = i < n_slots ? P.pairs[i] : make_int2(-1, -1);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) rk[r] = work_rank(bins, work_key(pr[r].y), pr[r].x >= 0, lane);
    __syncthreads();
    for (int i = tid; i < kWorkBins; i += 1024) {
        const int c = bins[i];
        bins[i] = c ? starts[i] + atomicAdd(&cursor[i], c) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (pr[r].x < 0) continue;
        const long long i = b0 + (long long)r * 1024 + tid;
        const int t = (int)(i / P.pairs_stride);
        work[bins[work_key(pr[r].y)] + rk[r]] = make_int2(t, (int)(i - (long long)t * P.pairs_stride));
    }
}

#ifndef FDCM_SEARCH_WPE
#define FDCM_SEARCH_WPE 4
#endif
template <bool BUF32>
__global__ void __launch_bounds__(64 * kWavesPerBlock, FDCM_SEARCH_WPE) k_search(const SearchParams P) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t, local;
    if (P.work) {
        // List order = launch order: workgroups go round-robin to the XCDs, so all XCDs work on the same
        // scene line at the same time.  (Giving every XCD its own contiguous part of the list was
        // measured slower: 0.32-0.39 ms against 0.27-0.29 ms.)
        const int per_xcd = (P.nblocks + 7) >> 3;
        const long long w = P.xcd_parts ? ((long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * kWavesPerBlock + wave
                                        : (long long)blockIdx.x * kWavesPerBlock + wave;
        if (w >= P.ncand) return;  // wave-uniform; waves never synchronise with each other
        const int2 e = P.work[w >> 1];  // the two alignments of a pair run next to each other
        t = e.x;
        local = e.y * 2 + (int)(w & 1);
    } else {
        t = blockIdx.x / P.bpt;
        local = (blockIdx.x - t * P.bpt) * kWavesPerBlock + wave;
    }
    const long long l0 = P.toffsets[t];
    const int n_t = (int)(P.toffsets[t + 1] - l0);
    const int count_t = 2 * min(n_t, P.maxT) * P.window;
    if (local >= count_t) return;  // wave-uniform; waves never synchronise with each other
    // ---- per-wave LDS: keys | aligned lines | scores
    float* s_keys = lds + (size_t)wave * (P.m + 5 * P.lds_lines + 2 * P.win + 1);
    float* L = s_keys + P.m;
    float* sc = L + 5 * P.lds_lines;
    for (int i = lane; i < P.m; i += 64) s_keys[i] = P.keys[i];
    const long long cand = P.cand_offsets[t] + local;
#ifdef FDCM_LAB
    unsigned long long* lab = P.lab ? P.lab + 8 * cand : nullptr;
    if (lab && lane == 0) { lab[0] = __builtin_amdgcn_s_memtime(); lab[6] = wall_clock64(); }
#endif
    const VolRef V = make_volref(P.vol, ivol_slice_floats(P.W, P.H), P.m, BUF32);

    // ---- which candidate: sorted template line j, window slot wi, alignment flip
    const int flip = local & 1, pair = local >> 1;
    const int2 pr = P.pairs[(size_t)t * P.pairs_stride + pair];
    const int tl_local = pr.x, scene_idx = pr.y;
    float tl[4], sl[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { tl[c] = P.tlines[(l0 + tl_local) * 4 + c]; sl[c] = P.slines[(size_t)scene_idx * 4 + c]; }
    const float* s_tl = P.tlines + l0 * 4;  // template lines straight from HBM (coalesced 16 B per lane)
#ifdef FDCM_LAB
    if (lab && tl[0] + sl[0] != 1.2345e-30f) SEARCH_STAMP(lab, 1);
#endif
    // align + transform, defaultmatch.cpp:59-67
    float T1[6], T2[6], T[6];
    align_pair(tl, sl, T1, T2);
#pragma unroll
    for (int c = 0; c < 6; ++c) T[c] = flip ? T2[c] : T1[c];
    float mnx = f_inf(), mny = f_inf(), mxx = -f_inf(), mxy = -f_inf();
    for (int i = lane; i < n_t; i += 64) {
        const float* p = s_tl + 4 * i;
        const float x1 = (T[0] * p[0] + T[1] * p[1]) + T[2], y1 = (T[3] * p[0] + T[4] * p[1]) + T[5];
        const float x2 = (T[0] * p[2] + T[1] * p[3]) + T[2], y2 = (T[3] * p[2] + T[4] * p[3]) + T[5];
        int bin;
        if (P.host_bins) {
            bin = (int)P.host_bins[(size_t)cand * P.lds_lines + i];
        } else {
            const float angle = atanf_glibc((y2 - y1) / (x2 - x1));  // getAngle, math.h:295-299
            bin = closest_orientation(s_keys, P.m, angle);  // dt3cpu.cpp:144-148
        }
        float* d = L + 5 * i;
        d[0] = x1; d[1] = y1; d[2] = x2; d[3] = y2; d[4] = line_slice_word(V, bin);
        mnx = std_min(mnx, std_min(x1, x2)); mxx = std_max(mxx, std_max(x1, x2));
        mny = std_min(mny, std_min(y1, y2)); mxy = std_max(mxy, std_max(y1, y2));
    }
    mnx = wave_min_f(mnx); mny = wave_min_f(mny); mxx = wave_max_f(mxx); mxy = wave_max_f(mxy);
#ifdef FDCM_LAB
    if (lab && mnx != 1.2345e-30f) SEARCH_STAMP(lab, 2);
#endif

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
    int best_k = 0, n_eval = 0;
    if (valid) {
        OptState o;
        o.V = V; o.L = L; o.sc = sc; o.n_t = n_t; o.H = (unsigned)P.H; o.tx = P.tx; o.ty = P.ty;
        o.savx = savx; o.savy = savy; o.lane = lane;
#ifdef FDCM_LAB
        o.lab = lab;
#endif
        o.B = P.optimizer == FDCM_BATCH_OPTIMIZE ? P.batch : 1; o.WIN = P.win; o.batch_rule = P.optimizer == FDCM_BATCH_OPTIMIZE;
        o.reset_back = P.optimizer == FDCM_INDULGENT_OPTIMIZE;
        // static_cast<long>(max_mul / min_mul), batchoptimize.cpp:51,74
        // One component of the scaled align vector is exactly +-1 (rasterizeVector), so the admissible multipliers
        // are bounded by the feature size (< 2^14): 32-bit multipliers walk exactly like the reference's longs, and
        // float(k) is one conversion instead of a 64-bit sequence.  (The clamp only keeps the arithmetic defined.)
        const long long lp = (long long)max_mul, ln = (long long)min_mul;
        o.lim_p = (int)std::min<long long>(std::max<long long>(lp, -(1ll << 30)), 1ll << 30);
        o.lim_n = (int)std::min<long long>(std::max<long long>(ln, -(1ll << 30)), 1ll << 30);
        optimise<BUF32>(o, best, best_k, n_eval);
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
        // translations the reference's rule evaluated (reduced by the compaction kernels: one
        // contended atomic per candidate would serialise the whole grid at ~12 ns each)
        P.evals[cand] = (int)n_eval;
#ifdef FDCM_LAB
        if (lab) { lab[4] = __builtin_amdgcn_s_memtime(); lab[7] = wall_clock64(); lab[5] = (unsigned long long)n_eval; }
#endif
    }
}

// Positional compaction of the valid records (defaultmatch.cpp:76-86) in two small kernels:
// per-chunk counts, then each chunk sums the counts before it, scans its flags and scatters.
static constexpr int kChunk = 1024;
__global__ void __launch_bounds__(256) k_chunk_counts(const int* __restrict__ flags, const int* __restrict__ evals,
                                                      long long n, int* __restrict__ counts, int* __restrict__ evsums) {
    __shared__ int part[4], epart[4];
    const long long c0 = (long long)blockIdx.x * kChunk;
    int s = 0, e = 0;
    for (int i = threadIdx.x; i < kChunk; i += 256) {
        if (c0 + i < n) { s += flags[c0 + i]; e += evals[c0 + i]; }
    }
    for (int d = 32; d >= 1; d >>= 1) { s += __shfl_xor(s, d); e += __shfl_xor(e, d); }
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = s; epart[threadIdx.x >> 6] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
        evsums[blockIdx.x] = epart[0] + epart[1] + epart[2] + epart[3];
    }
}
__global__ void __launch_bounds__(1024) k_scatter(const fdcm_match* __restrict__ records, const int* __restrict__ flags,
                                                  long long n, const int* __restrict__ counts,
                                                  const int* __restrict__ evsums, int nchunks,
                                                  fdcm_match* __restrict__ out, unsigned long long* __restrict__ counters,
                                                  fdcm_match* __restrict__ out2, unsigned long long* __restrict__ counters2) {
    __shared__ long long wsum[16];
    __shared__ long long chunk_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long s = 0;
    for (int i = tid; i < (int)blockIdx.x; i += 1024) s += counts[i];
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    if (tid == 0) {
        long long b = 0;
        for (int w = 0; w < 16; ++w) b += wsum[w];
        chunk_base = b;
    }
    __syncthreads();
    const long long i = (long long)blockIdx.x * kChunk + tid;
    const int f = i < n ? flags[i] : 0;
    int incl = f;
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    long long wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wsum[w];
    if (f) {
        const fdcm_match r = records[i];
        out[chunk_base + wbase + incl - 1] = r;
        if (out2) out2[chunk_base + wbase + incl - 1] = r;  // pinned host memory: the caller's copy, written in place
    }
    if ((int)blockIdx.x == nchunks - 1) {
        if (tid == 1023) {
            counters[2] = (unsigned long long)(chunk_base + wbase + incl);
            if (counters2) counters2[2] = (unsigned long long)(chunk_base + wbase + incl);
        }
        long long ev = 0;
        for (int i2 = tid; i2 < nchunks; i2 += 1024) ev += evsums[i2];
        for (int d = 32; d >= 1; d >>= 1) ev += __shfl_xor(ev, d);
        __syncthreads();
        if (lane == 0) wsum[wave] = ev;
        __syncthreads();
        if (tid == 0) {
            long long tot = 0;
            for (int w = 0; w < 16; ++w) tot += wsum[w];
            counters[0] = (unsigned long long)tot;
            if (counters2) counters2[0] = (unsigned long long)tot;
        }
    }
}

// The same compaction in one launch, for searches of at most kCompactChunks chunks (65 536 candidates): a block sums the
// flags of the chunks before its own itself (at most 256 KB from L2) instead of reading per-chunk counts that an earlier
// launch left -- one kernel boundary and one small kernel less on a blocking frame.
static constexpr int kCompactChunks = 64;
__global__ void __launch_bounds__(1024) k_compact(const fdcm_match* __restrict__ records, const int* __restrict__ flags,
                                                  const int* __restrict__ evals, long long n, int nchunks,
                                                  fdcm_match* __restrict__ out, unsigned long long* __restrict__ counters,
                                                  fdcm_match* __restrict__ out2, unsigned long long* __restrict__ counters2) {
    __shared__ long long wsum[16];
    __shared__ long long chunk_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long i = (long long)blockIdx.x * kChunk + tid;
    const int f = i < n ? flags[i] : 0;
    fdcm_match r{};
    if (f) r = records[i];  // in flight while the earlier chunks are summed
    long long s = 0;
    {
        const int4* f4 = reinterpret_cast<const int4*>(flags);  // whole chunks: multiples of 1024 ints from an aligned base
        const long long n4 = (long long)blockIdx.x * (kChunk / 4);
        for (long long q = tid; q < n4; q += 1024) { const int4 v = f4[q]; s += v.x + v.y + v.z + v.w; }
    }
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    if (tid == 0) {
        long long b = 0;
        for (int w = 0; w < 16; ++w) b += wsum[w];
        chunk_base = b;
    }
    int incl = f;
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    __syncthreads();
    const long long cb = chunk_base;
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    long long wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wsum[w];
    if (f) {
        out[cb + wbase + incl - 1] = r;
        if (out2) out2[cb + wbase + incl - 1] = r;  // pinned host memory: the caller's copy, written in place
    }
    if ((int)blockIdx.x == nchunks - 1) {
        if (tid == 1023) {
            counters[2] = (unsigned long long)(cb + wbase + incl);
            if (counters2) counters2[2] = (unsigned long long)(cb + wbase + incl);
        }
        long long ev = 0;
        for (long long i2 = tid; i2 < n; i2 += 1024) ev += evals[i2];
        for (int d = 32; d >= 1; d >>= 1) ev += __shfl_xor(ev, d);
        __syncthreads();
        if (lane == 0) wsum[wave] = ev;
        __syncthreads();
        if (tid == 0) {
            long long tot = 0;
            for (int w = 0; w < 16; ++w) tot += wsum[w];
            counters[0] = (unsigned long long)tot;
            if (counters2) counters2[0] = (unsigned long long)tot;
        }
    }
}

int64_t search_capacity(const fdcm_templates* t, int64_t n_scene, int64_t maxT, int64_t maxS) {
    int64_t total = 0;
    // the reference takes size_t limits and applies min(tmpl.cols(), maxTmplLines) / min(scene, maxSceneLines)
    // (defaultsearch.cpp:38, defaultsearch.h:42-46): "all lines" values such as 2^40 are legal
    maxT = std::min<int64_t>(std::max<int64_t>(maxT, 0), std::max<int64_t>(t->max_lines, 0));
    const int64_t window = std::min<int64_t>(std::max<int64_t>(maxS, 0), n_scene);
    for (int64_t i = 0; i < t->T; ++i) {
        const int64_t nt = t->offsets[i + 1] - t->offsets[i];
        total += 2 * std::min<int64_t>(nt, maxT) * window;
    }
    return total;
}

bool orientation_bins_on_host() {
    static const bool on_host = [] {
        const bool forced = getenv("FDCM_FORCE_HOST_BINS") != nullptr;
        const bool differs = fdcm_selftest_atanf(0, 65537, (1ull << 32) / 65537) != 0;
        if (forced || differs)
            fprintf(stderr, "libfdcm_hip: orientation bins of the candidates come from this machine's libm on host threads (%s); "
                            "searches are slower than with the device's atanf (fdcm_orientation_bins_mode() == 1)\n",
                    differs ? "its atanf differs from the glibc 2.35 restatement the device runs" : "FDCM_FORCE_HOST_BINS is set");
        return forced || differs;
    }();
    return on_host;
}

// The workspaces run_search takes, sized for (template set, scene size, window): the same expressions, nothing queued.
void reserve_search(fdcm_featuremap* fm, const fdcm_templates* t, int64_t n_scene, int64_t maxT, int64_t maxS) {
    if (!t || t->T == 0 || n_scene <= 0) return;
    FDCM_HIP(hipSetDevice(fm->device));
    maxT = std::min<int64_t>(std::max<int64_t>(maxT, 0), std::max<int64_t>(t->max_lines, 0));
    maxS = std::min<int64_t>(std::max<int64_t>(maxS, 0), n_scene);
    const int64_t ncand = search_capacity(t, n_scene, maxT, maxS);
    if (ncand == 0) return;
    auto align16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t n_s = (size_t)n_scene;
    const size_t blob = align16(n_s * 16) + 2 * align16(n_s * 4) + align16(((size_t)t->T + 1) * 8);
    const int nchunks = (int)((ncand + kChunk - 1) / kChunk);
    fm->s_stage.reserve(blob);
    fm->s_scene.reserve(blob);
    fm->s_records.reserve((size_t)ncand * sizeof(fdcm_match));
    fm->s_flags.reserve(2 * ((size_t)ncand + (size_t)nchunks) * sizeof(int));
    fm->s_counter.reserve(64);
    fm->s_cnt.reserve(64);
    const size_t pairs_stride = (size_t)(maxT * maxS);
    fm->s_pairs.reserve(std::max<size_t>(16, (size_t)t->T * pairs_stride * sizeof(int2)));
    fm->s_work.reserve((((size_t)(ncand / 2) * sizeof(int2) + 15) & ~(size_t)15) + 2 * (size_t)kWorkBins * sizeof(int));
    if (orientation_bins_on_host()) {
        const size_t stride = (size_t)std::max<int64_t>(1, t->max_lines);
        fm->s_bins_stage.reserve((size_t)ncand * stride * sizeof(unsigned short));
        fm->s_bins.reserve((size_t)ncand * stride * sizeof(unsigned short));
    }
}

void run_search(fdcm_featuremap* fm, const fdcm_templates* t, const float* scene, int64_t n_scene, int64_t maxT,
                int64_t maxS, int optimizer, int64_t batch, int32_t base, fdcm_match* out_device, fdcm_match** out_host,
                int64_t* n_out) {
    *n_out = 0;
    fm->last_search = fdcm_search_timing{};
    // early-outs of search<DefaultMatch>, defaultmatch.cpp:40-41
    if (t->T == 0 || n_scene == 0 || (fm->W == 0 && fm->H == 0)) return;
    // The orientation bins of the aligned template lines come from the device restatement of glibc 2.35's atanf, the
    // scene's from the host libm (as in the reference): on a host whose libm differs (a newer glibc ships a correctly
    // rounded atanf) the two would silently disagree.  The first search checks a sample (every 65537th bit pattern, a
    // few ms); on a mismatch the bins of every candidate's lines are computed on the host instead -- same transform,
    // this machine's atanf, a few host threads, overlapping the build that is still running on the device -- so that
    // parity is defined against the box's own libm, as the reference's would be.  FDCM_FORCE_HOST_BINS=1 forces it.
    const bool host_bins_needed = orientation_bins_on_host();
    FDCM_HIP(hipSetDevice(fm->device));
    if (!fm->stream) FDCM_HIP(hipStreamCreateWithFlags(&fm->stream, hipStreamNonBlocking));
    if (!fm->timing.created) {
        for (auto& e : fm->timing.ev) FDCM_HIP(hipEventCreate(&e));
        fm->timing.created = true;
    }
    hipStream_t st = fm->stream;
    const int n_s = (int)n_scene;
    // Clamp the limits before any cast or sizing: the reference takes size_t and applies min() (defaultsearch.cpp:38,
    // defaultsearch.h:42-46), so DefaultSearch(10**12, 10**12) means "all lines".
    maxT = std::min<int64_t>(std::max<int64_t>(maxT, 0), std::max<int64_t>(t->max_lines, 0));
    maxS = std::min<int64_t>(std::max<int64_t>(maxS, 0), n_scene);
    const int window = (int)maxS;
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
    int64_t cpt_max = 0;
    for (int64_t i = 0; i < t->T; ++i) {
        const int64_t nt = t->offsets[i + 1] - t->offsets[i];
        const int64_t c = 2 * std::min<int64_t>(nt, maxT) * window;
        coff[i + 1] = coff[i] + c;
        cpt_max = std::max(cpt_max, c);
    }
    const long long ncand = coff[t->T];
    fm->last_search.candidates = ncand;
    if (ncand == 0) return;
    // ---- every check that can refuse the search, before the first command is queued (a refusal must not leave an upload
    // in flight on the preparation stream that the next call's staging would overwrite)
    if (fm->vol_stage != 3) throw std::string("the feature map holds a partial build (no line integral): nothing to search");
    const int64_t B = optimizer == FDCM_BATCH_OPTIMIZE ? std::max<int64_t>(1, batch) : 1;
    if (B > 4096) throw std::string("batch_size above 4096 is not supported");
    const int win = B <= 15 ? (int)((15 / B) * B) : (int)B;  // multipliers scored per direction and round: whole batches, 15 at most when they fit one round
    const int lds_lines = (int)std::max<int64_t>(1, t->max_lines);
    const size_t lds = (size_t)kWavesPerBlock * ((size_t)fm->m + 5 * (size_t)lds_lines + 2 * (size_t)win + 1) * sizeof(float);
    if (lds > 160 * 1024) throw std::string("scene/template too large for the search kernel's LDS staging");
    if (host_bins_needed && fm->m > 65535) throw std::string("host-side orientation bins need depth <= 65535");
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
    // The search's preparation (scene upload, k_pairs, the work list) needs nothing of the volume: on a handle that has the GPU
    // to itself it goes to a second stream, beside the kernels of a build that is still running on `st` (a blocking
    // rebuild -> search spends ~25 us less); k_search waits for it through an event.  A slot of a frame pipeline keeps one
    // stream (its frames already run beside each other, and every stream takes a hardware queue).
    hipStream_t sp = st;
    if (!fm->shares_gpu) {
        if (!fm->prep_stream) {
            FDCM_HIP(hipStreamCreateWithFlags(&fm->prep_stream, hipStreamNonBlocking));
            FDCM_HIP(hipEventCreateWithFlags(&fm->prep_done, hipEventDisableTiming));
        }
        sp = fm->prep_stream;
    }
    FDCM_HIP(hipMemcpyAsync(fm->s_scene.p, hs, blob, hipMemcpyHostToDevice, sp));
    const int nchunks = (int)((ncand + kChunk - 1) / kChunk);
    fm->s_records.reserve((size_t)ncand * sizeof(fdcm_match));
    fm->s_flags.reserve(2 * ((size_t)ncand + (size_t)nchunks) * sizeof(int));
    fm->s_counter.reserve(64);
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
    P.optimizer = optimizer;
    P.batch = (int)B;
    P.win = win;
    P.base = base;
    P.cand_offsets = (const long long*)(ds + o_coff);
    P.bpt = (int)((cpt_max + kWavesPerBlock - 1) / kWavesPerBlock);
    P.ncand = ncand;
    static const int env_xcd_parts = getenv("FDCM_SEARCH_XCD_PARTS") ? atoi(getenv("FDCM_SEARCH_XCD_PARTS")) : 0;  // tuning override, read once
    P.xcd_parts = env_xcd_parts;
    P.lds_lines = lds_lines;
    P.records = fm->s_records.as<fdcm_match>();
    P.flags = fm->s_flags.as<int>();
    P.evals = P.flags + ncand + nchunks;
    P.counters = fm->s_counter.as<unsigned long long>();
    P.pairs_stride = (int)(maxT * window);
    fm->s_pairs.reserve(std::max<size_t>(16, (size_t)t->T * P.pairs_stride * sizeof(int2)));
    P.pairs = fm->s_pairs.as<int2>();
    // volumes below 4 GB (every BASELINE config but 5) are addressed through one buffer descriptor with 32-bit offsets
    static const bool env_flat = getenv("FDCM_SEARCH_FLAT") != nullptr;  // measurement: 64-bit flat addresses always
    const bool buf32 = !env_flat && (size_t)fm->m * ivol_slice_floats(fm->W, fm->H) * sizeof(float) < ((size_t)1 << 32);
    if (lds > 64 * 1024)
        FDCM_HIP(hipFuncSetAttribute(buf32 ? (const void*)k_search<true> : (const void*)k_search<false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t* ev = fm->timing.ev;
    const bool timed = fm->want_stage_events != 0;  // fdcm_featuremap_stage_timing(fm, 0): no events around the search either
    if (timed) FDCM_HIP(hipEventRecord(ev[6], st));
    hipLaunchKernelGGL(k_pairs, dim3((unsigned)(((size_t)t->T * maxT + 255) / 256)), dim3(256), 0, sp, P);
    const long long n_slots = (long long)t->T * P.pairs_stride;
    static const bool no_worklist = getenv("FDCM_SEARCH_TEMPLATE_MAJOR") != nullptr;  // tuning override
    if (!no_worklist && n_slots <= 0x7fffffffll) {
        const size_t work_bytes = ((size_t)(ncand / 2) * sizeof(int2) + 15) & ~(size_t)15;
        fm->s_work.reserve(work_bytes + 2 * (size_t)kWorkBins * sizeof(int));
        if (n_slots <= kWlSlotsPerBlock) {
            hipLaunchKernelGGL(k_worklist, dim3(1), dim3(1024), 0, sp, P, n_slots, fm->s_work.as<int2>());
        } else {
            int* ghist = (int*)((char*)fm->s_work.p + work_bytes);
            int* cursor = ghist + kWorkBins;
            const unsigned nb = (unsigned)((n_slots + kWlSlotsPerBlock - 1) / kWlSlotsPerBlock);
            FDCM_HIP(hipMemsetAsync(ghist, 0, (size_t)kWorkBins * sizeof(int), sp));
            hipLaunchKernelGGL(k_wl_count, dim3(nb), dim3(1024), 0, sp, P, n_slots, ghist);
            hipLaunchKernelGGL(k_wl_starts, dim3(1), dim3(1024), 0, sp, ghist, cursor);
            hipLaunchKernelGGL(k_wl_scatter, dim3(nb), dim3(1024), 0, sp, P, n_slots, ghist, cursor, fm->s_work.as<int2>());
        }
        P.work = fm->s_work.as<int2>();
        P.nblocks = (int)(((ncand + kWavesPerBlock - 1) / kWavesPerBlock + 7) / 8 * 8);
    } else {
        P.work = nullptr;
        P.nblocks = (int)((size_t)t->T * P.bpt);
    }
    if (sp != st) {  // k_search (and the host bins' upload) behind the preparation
        FDCM_HIP(hipEventRecord(fm->prep_done, sp));
        FDCM_HIP(hipStreamWaitEvent(st, fm->prep_done, 0));
    }
    if (host_bins_needed) {
        const size_t stride = (size_t)P.lds_lines;
        fm->s_bins_stage.reserve((size_t)ncand * stride * sizeof(unsigned short));
        fm->s_bins.reserve((size_t)ncand * stride * sizeof(unsigned short));
        unsigned short* hb = (unsigned short*)fm->s_bins_stage.p;
        const float* keys = fm->keys.data();
        const int mkeys = (int)fm->m;
        std::vector<float> sorted_len((size_t)n_s);
        for (int i = 0; i < n_s; ++i) sorted_len[i] = slen[sidx[i]];
        const unsigned nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        std::atomic<int64_t> next{0};
        const fdcm_templates* tt = t;
        for (unsigned w = 0; w < nthreads; ++w)
            th.emplace_back([&, tt] {
                for (int64_t ti; (ti = next.fetch_add(1)) < tt->T;) {
                    const int64_t l0 = tt->offsets[ti];
                    const int n_t = (int)(tt->offsets[ti + 1] - l0);
                    const int jmax = (int)std::min<int64_t>(n_t, maxT);
                    for (int j = 0; j < jmax; ++j) {  // establishSearchStrategy<DefaultSearch>, as k_pairs
                        const int tl_local = tt->sorted[(size_t)l0 + j];
                        const float tlen = tt->lengths[(size_t)l0 + tl_local];
                        const int centre = binary_search_greater(sorted_len.data(), n_s, tlen);
                        int rb, re;
                        centered_range(centre, n_s, (int)maxS, rb, re);
                        for (int wi = 0; wi < window; ++wi) {
                            const int scene_idx = (int)sidx[rb + wi];
                            float T1[6], T2[6];
                            align_pair(&tt->lines[(size_t)(l0 + tl_local) * 4], scene + (size_t)scene_idx * 4, T1, T2);
                            for (int flip = 0; flip < 2; ++flip) {
                                const float* T = flip ? T2 : T1;
                                const long long cand = coff[ti] + 2ll * (j * window + wi) + flip;
                                unsigned short* out = hb + (size_t)cand * stride;
                                for (int i = 0; i < n_t; ++i) {
                                    const float* p = &tt->lines[(size_t)(l0 + i) * 4];
                                    const float x1 = (T[0] * p[0] + T[1] * p[1]) + T[2], y1 = (T[3] * p[0] + T[4] * p[1]) + T[5];
                                    const float x2 = (T[0] * p[2] + T[1] * p[3]) + T[2], y2 = (T[3] * p[2] + T[4] * p[3]) + T[5];
                                    const float angle = std::atan((y2 - y1) / (x2 - x1));  // the host libm, like the scene's bins
                                    out[i] = (unsigned short)closest_orientation(keys, mkeys, angle);
                                }
                            }
                        }
                    }
                }
            });
        for (auto& x : th) x.join();
        FDCM_HIP(hipMemcpyAsync(fm->s_bins.p, hb, (size_t)ncand * stride * sizeof(unsigned short), hipMemcpyHostToDevice, st));
        P.host_bins = fm->s_bins.as<unsigned short>();
    }
#ifdef FDCM_LAB
    static const bool env_lab = getenv("FDCM_SEARCH_LAB") != nullptr;
    static DevBuf labbuf;
    P.lab = nullptr;
    if (env_lab) { labbuf.reserve((size_t)ncand * 64); FDCM_HIP(hipMemsetAsync(labbuf.p, 0, (size_t)ncand * 64, st)); P.lab = labbuf.as<unsigned long long>(); }
    if (lab_skip("search")) {} else
#endif
    if (buf32) hipLaunchKernelGGL(k_search<true>, dim3((unsigned)P.nblocks), dim3(64 * kWavesPerBlock), lds, st, P);
    else hipLaunchKernelGGL(k_search<false>, dim3((unsigned)P.nblocks), dim3(64 * kWavesPerBlock), lds, st, P);
#ifdef FDCM_LAB
    if (env_lab) {
        FDCM_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> h((size_t)ncand * 8);
        FDCM_HIP(hipMemcpy(h.data(), labbuf.p, h.size() * 8, hipMemcpyDeviceToHost));
        double d[4] = {0, 0, 0, 0}, life = 0, ev = 0;
        unsigned long long w0 = ~0ull, w1 = 0;
        long long n = 0;
        for (long long c = 0; c < ncand; ++c) {
            const unsigned long long* e = &h[(size_t)c * 8];
            if (!e[4] || !e[3]) continue;
            d[0] += (double)(e[1] - e[0]); d[1] += (double)(e[2] - e[1]); d[2] += (double)(e[3] - e[2]); d[3] += (double)(e[4] - e[3]);
            life += (double)(e[7] - e[6]); ev += (double)e[5];
            w0 = std::min(w0, e[6]); w1 = std::max(w1, e[7]);
            ++n;
        }
        if (n) fprintf(stderr, "[search lab] %lld waves: loads %.0f  lines+bins %.0f  round1 %.0f  rule+rounds %.0f cycles; life %.2f us (100 MHz clock), span %.1f us, evals/wave %.1f\n",
                       n, d[0] / n, d[1] / n, d[2] / n, d[3] / n, life / n / 100.0, (double)(w1 - w0) / 100.0, ev / n);
    }
#endif
    fdcm_match* dst = out_device;
    if (!dst) {
        // host output: one extra record behind the candidates' capacity carries the counters, so that the
        // matches and their count come back with a single copy
        fm->s_out.reserve((size_t)(ncand + 1) * sizeof(fdcm_match));
        dst = fm->s_out.as<fdcm_match>();
        P.counters = reinterpret_cast<unsigned long long*>(dst + ncand);
    }
    int* d_counts = P.flags + ncand;
    int* d_evsums = P.evals + ncand;
    // The matches (host-output search) and the counters reach the host without a copy command: k_scatter writes them
    // into pinned host memory itself.  (A hipMemcpyAsync behind the kernels stalled for 8 - 12 ms about once per 100 ms
    // of running; the kernels before it and the build were on time.)
    fdcm_match* host_out = nullptr;
    unsigned long long* host_cnt = nullptr;
    if (out_host) {
        *out_host = result_acquire((size_t)(ncand + 1) * sizeof(fdcm_match));
        FDCM_HIP(hipHostGetDevicePointer((void**)&host_out, *out_host, 0));
        host_cnt = reinterpret_cast<unsigned long long*>(host_out + ncand);
    } else {
        fm->s_cnt.reserve(64);
        FDCM_HIP(hipHostGetDevicePointer((void**)&host_cnt, fm->s_cnt.p, 0));
    }
    static const bool env_two_step = getenv("FDCM_SEARCH_COMPACT2") != nullptr;  // the tests' switch: the two-kernel form at every size
    if (nchunks <= kCompactChunks && !env_two_step) {
        hipLaunchKernelGGL(k_compact, dim3((unsigned)nchunks), dim3(1024), 0, st, P.records, P.flags, P.evals, ncand, nchunks, dst, P.counters,
                           host_out, host_cnt);
    } else {
        hipLaunchKernelGGL(k_chunk_counts, dim3((unsigned)nchunks), dim3(256), 0, st, P.flags, P.evals, ncand, d_counts, d_evsums);
        hipLaunchKernelGGL(k_scatter, dim3((unsigned)nchunks), dim3(1024), 0, st, P.records, P.flags, ncand, d_counts, d_evsums,
                           nchunks, dst, P.counters, host_out, host_cnt);
    }
    if (timed) FDCM_HIP(hipEventRecord(ev[7], st));
    FDCM_HIP(hipGetLastError());
    unsigned long long hc[3] = {0, 0, 0};
    FDCM_HIP(hipStreamSynchronize(st));
    finish_build(fm);  // a build queued before this search is complete as well: collect its timings
    std::memcpy(hc, out_host ? (const void*)(*out_host + ncand) : (const void*)fm->s_cnt.p, sizeof hc);
    *n_out = (int64_t)hc[2];
    fm->last_search.evaluations = (int64_t)hc[0];
    if (timed) FDCM_HIP(hipEventElapsedTime(&fm->last_search.kernel_ms, ev[6], ev[7]));
    // the search's own span on the device: kernels and download are one thing since the compaction writes the matches into
    // host memory itself (host preparation overlaps a build that is still running, and the wait for that build is not the
    // search's time)
    fm->last_search.total_ms = fm->last_search.kernel_ms;
}

}  // namespace fdcm
