// fdcm_build.hip -- DT3 feature-map build on gfx950 (buildCpuFeaturemap<D>, dt3cpu.h:174-234).
//
// Volume layout in HBM, from the sweeps on: interleaved, [k][x/4][y][x%4] (ivol_index, fdcm_internal.h): 16 bytes hold 4
// neighbouring columns of one row.  The sweeps write the transforms into `vol`, the propagation reads them and writes
// `ivol`, the line integral reads `ivol` and writes its sums back into `vol`, which is what the search gathers from.
//
// Kernels (W x H = feature size, m = slices, V = 4*m*W*H bytes):
//   K0 k_seeds          clipped scene lines -> seed bitmap (1 bit per pixel, bits along y)   ~V/32
//   K1 k_coldesc_tile   per column and 64-row chunk: seed bits + nearest seed before/after   ~V/16
//   K2 L2 / L2^2        both 1-D passes in one sweep along x, pass 1 from the descriptors: fdcm_sweep.hip (column ranges of
//                       equal count, merged) where W^2 + H^2 <= 2^24, fdcm_sweep_literal.hip (one wave per chunk) above   write V
//      L1  k_l1_*       both sweeps with one pass over the volume (minima per word from the descriptors, carries, word by word)   write V
//   K3 k_propagate_reg<M> / k_propagate   orientation propagation, 4m steps per pixel in
//                       registers (generic depth: LDS) (+ sqrt for L2)                        read V, write V
//   K4 k_integral       directional prefix sum per slice, one sequential float chain per row / column of 16-byte
//                       units; shallow and steep sweeps (the latter through LDS tiles) in one launch  read V, write V
// Compiled with -ffp-contract=off; divide and sqrt are the correctly rounded forms.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "fdcm_build_dev.h"
#include "fdcm_internal.h"
#include "fdcm_quotient.h"
#include "fdcm_sweep.h"

namespace fdcm {


// ------------------------------------------------------------------------------------------ K0
// drawLines (drawing.h:111-125): one block per clipped line, threads over its raster points.
__global__ void k_seeds(const RasterLine* __restrict__ lines, unsigned long long* __restrict__ bitmap, int W, int H,
                        int HW64) {
    const RasterLine r = lines[blockIdx.x];
    for (int i = threadIdx.x; i < r.n; i += blockDim.x) {
        const float fx = lin_spaced_value(r.xmode, r.xlow, r.xhigh, r.xstep, r.n, i);
        const float fy = lin_spaced_value(r.ymode, r.ylow, r.yhigh, r.ystep, r.n, i);
        const long x = (long)roundf(fx);  // .round().cast<Eigen::Index>(): half away from zero
        const long y = (long)roundf(fy);
        if (x < 0 || x >= W || y < 0 || y >= H) continue;  // the reference would write out of bounds
        atomicOr(&bitmap[((size_t)r.slice * W + x) * HW64 + (y >> 6)], 1ull << (y & 63));
    }
}

// ------------------------------------------------------------------------------------------ K1
// Pass 1 of distanceTransform (imgproc.h:178 / :186 along y).  On a 0 / FLT_MAX image the
// lower-envelope pass yields exactly the squared distance to the nearest seed of the column
// (every envelope owner is a seed and owns itself), or FLT_MAX for a seedless column; the L1
// sweeps yield the plain distance.  Both are integers < 2^24, so any exact method gives the
// reference's bits.  One wave per column (k, x); lanes are 64 consecutive y.
// The seed words are cleared as they are read (when one pass reads each word once), so the next build of
// the same size starts from a zero bitmap without a separate fill.
__global__ void __launch_bounds__(256) k_coldesc(unsigned long long* __restrict__ bitmap,
                                                 ColDesc* __restrict__ desc, int W, int HW64, long ncols) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long col = (long)blockIdx.x * (blockDim.x >> 6) + wave;  // = k * W + x
    if (col >= ncols) return;
    const long k = col / W, x = col - k * W;
    unsigned long long* bw = bitmap + (size_t)col * HW64;
    const int ngroups = (HW64 + 63) >> 6;  // groups of 64 words = 4096 rows
    int carry_prev = INT_MIN;              // last seed row in earlier groups
    for (int g = 0; g < ngroups; ++g) {
        const int wi = g * 64 + lane;
        const unsigned long long word = wi < HW64 ? bw[wi] : 0ull;
        if (ngroups == 1 && word) bw[wi] = 0ull;
        const int last_i = word ? wi * 64 + 63 - __clzll(word) : INT_MIN;
        const int first_i = word ? wi * 64 + (__ffsll((long long)word) - 1) : INT_MAX;
        int carry_next = INT_MAX;  // first seed row in later groups (only when H > 4096)
        for (int g2 = ngroups - 1; g2 > g; --g2) {
            const int wj = g2 * 64 + lane;
            const unsigned long long w2 = wj < HW64 ? bw[wj] : 0ull;
            carry_next = min(carry_next, wave_min(w2 ? wj * 64 + (__ffsll((long long)w2) - 1) : INT_MAX));
        }
        ColDesc d;
        d.word = word;
        const int pv = max(wave_scan_max_excl(last_i, lane), carry_prev);
        const int nx = min(wave_scan_min_excl_rev(first_i, lane), carry_next);
        d.prev = pv == INT_MIN ? -kFar : pv;
        d.next = nx == INT_MAX ? kFar : nx;
        if (wi < HW64) desc[((size_t)k * HW64 + wi) * W + x] = d;
        carry_prev = max(carry_prev, wave_max(last_i));
    }
}

// The same for H <= 4096 (one group of words per column), with coalesced stores: a block takes 64 neighbouring
// columns of a slice, SEG lanes (16 / 32 / 64: the words of a column) scan one column each, 64 / SEG columns per wave
// pass, and the descriptors go through LDS so that every store covers the block's columns of one chunk (1 KB or 512 B
// contiguous; the wave-per-column kernel writes 16 bytes every W * 16).
// The seeds come straight from the slice's clipped lines (drawLines, drawing.h:111-125: k_seeds' points): every block
// rasterises the lines of its slice that can reach its XT columns into a bitmap tile in LDS -- no seed bitmap in memory, no
// launch of its own for a few thousand points, no atomics on HBM.
template <int SEG, int XT>  // XT columns per block: 64, or 32 when a column has 64 words (32 KB of LDS instead of 64)
__global__ void __launch_bounds__(256) k_coldesc_tile(const RasterLine* __restrict__ lines, const int* __restrict__ slice_first,
                                                      ColDesc* __restrict__ desc, int W, int H, int HW64, unsigned* __restrict__ colmask) {
    extern __shared__ uint4 tile[];  // [word][XT columns], rows padded by one unit (bank spread); behind it the seed bits [word][XT]
    constexpr int STR = XT + 1;
    unsigned long long* bits = reinterpret_cast<unsigned long long*>(tile + (size_t)HW64 * STR);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = lane & (SEG - 1), ci = lane / SEG;
    constexpr int CPP = 64 / SEG, CPW = XT / 4;  // columns per wave pass, columns per wave
    const long k = blockIdx.y;
    const int x0 = blockIdx.x * XT;
    for (int i = threadIdx.x; i < HW64 * XT; i += 256) bits[i] = 0ull;
    __syncthreads();
    for (int li = slice_first[k]; li < slice_first[k + 1]; ++li) {
        const RasterLine r = lines[li];
        // the line's columns: between its two ends (a constant when xmode == 0), half a pixel of rounding on either side
        const float xa = r.xmode == 0 ? r.xlow : fminf(r.xlow, r.xhigh), xb = r.xmode == 0 ? r.xlow : fmaxf(r.xlow, r.xhigh);
        if (xb + 1.f < (float)x0 || xa - 1.f > (float)(x0 + XT - 1)) continue;
        for (int i = threadIdx.x; i < r.n; i += 256) {
            const float fx = lin_spaced_value(r.xmode, r.xlow, r.xhigh, r.xstep, r.n, i);
            const long x = (long)roundf(fx);  // .round().cast<Eigen::Index>(): half away from zero
            if (x < x0 || x >= x0 + XT || x >= W) continue;
            const float fy = lin_spaced_value(r.ymode, r.ylow, r.yhigh, r.ystep, r.n, i);
            const long y = (long)roundf(fy);
            if (y < 0 || y >= H) continue;  // the reference would write out of bounds
            atomicOr(&bits[(y >> 6) * XT + (int)(x - x0)], 1ull << (y & 63));
        }
    }
    __syncthreads();
    for (int pass = 0; pass < CPW / CPP; ++pass) {
        const int xl = wave * CPW + pass * CPP + ci, x = x0 + xl;
        const bool valid = x < W && wi < HW64;
        const unsigned long long word = valid ? bits[wi * XT + xl] : 0ull;
        const int last_i = word ? wi * 64 + 63 - __clzll(word) : INT_MIN;
        const int first_i = word ? wi * 64 + (__ffsll((long long)word) - 1) : INT_MAX;
        int pmax = last_i, smin = first_i;  // inclusive scans inside the SEG lanes of a column
#pragma unroll
        for (int d = 1; d < SEG; d <<= 1) {
            const int a = __shfl_up(pmax, d), b = __shfl_down(smin, d);
            if (wi >= d) pmax = max(pmax, a);
            if (wi + d < SEG) smin = min(smin, b);
        }
        const int pe = __shfl_up(pmax, 1), se = __shfl_down(smin, 1);
        const int pv = wi == 0 ? INT_MIN : pe, nx = wi == SEG - 1 ? INT_MAX : se;
        if (wi < HW64)
            tile[wi * STR + xl] = make_uint4((unsigned)(word & 0xffffffffull), (unsigned)(word >> 32),
                                             (unsigned)(pv == INT_MIN ? -kFar : pv), (unsigned)(nx == INT_MAX ? kFar : nx));
    }
    __syncthreads();
    uint4* out = reinterpret_cast<uint4*>(desc);
    for (int idx = threadIdx.x; idx < HW64 * XT; idx += 256) {
        const int w = idx / XT, xl = idx - w * XT;
        if (x0 + xl < W) out[((size_t)k * HW64 + w) * W + x0 + xl] = tile[w * STR + xl];
    }
    // The slice's seeded columns, one bit per column ((W + 63) / 64 words of 64 bits per slice, written as 32-bit halves):
    // the L2 sweep skips the others, and every one of its workgroups used to rebuild this mask from the descriptors.
    if (colmask && wave == 0) {
        const bool seeded = lane < XT && x0 + lane < W && !desc_seedless(tile[min(lane, XT - 1)]);  // chunk 0's descriptor says it for the column
        const unsigned long long mk = __ballot(seeded);
        unsigned* dst = colmask + ((size_t)k * ((W + 63) >> 6) + (x0 >> 6)) * 2;
        if (XT == 64) { if (lane < 2) dst[lane] = lane ? (unsigned)(mk >> 32) : (unsigned)mk; }
        else {
            const int half = (x0 >> 5) & 1;
            if (lane == 0) dst[half] = (unsigned)mk;
            if (lane == 1 && half == 0 && x0 + 32 >= W) dst[1] = 0u;  // no block for the word's upper half
        }
    }
}

// ---- The two L1 sweeps with ONE pass over the volume (round 3).
// The forward sweep is F[x] = min(c[x], F[x-1] + 1) on the column values c (exact integers, or FLT_MAX for a seedless column,
// which absorbs every +1 and +-x), the backward one R[x] = min(F[x], R[x+1] + 1) on its result (imgproc.h:138-145).  As two
// kernels they write V, read V and write V again; the 24 GB of config 5 went at the rate of a plain copy (2.4 + 4.9 ms).
// Both recurrences only carry one number per row across a cut: F[x0 - 1] = min over x' < x0 of (c[x'] - x') + (x0 - 1), and
// for the backward one R[x1 + 1] may be replaced by B[x1 + 1] = min over x' > x1 of (c[x'] + x') - (x1 + 1), the plain
// backward scan of c (R[x1] = min(F[x1], F[x1+1] + 1, B[x1+1] + 1) and F[x1+1] + 1 = min(c[x1+1] + 1, F[x1] + 2) is never
// below min(F[x1], B[x1+1] + 1)).  All terms are integers below 2^24 or FLT_MAX: every float operation is exact, so the
// regrouping changes no bit.  So: the two minima per (row, 64-column word) from the descriptors alone (k_l1_word_mins),
// their exclusive prefix / suffix minima over the words of a row (k_l1_carries, 3 % of V), and then every word by itself:
// c from the descriptors, forward from its carry into 64 registers, backward from its carry, one store (k_l1_word).
__device__ __forceinline__ float l1_column_value(const ColDesc& dcur, unsigned long long nz, int j, int lane, int y) {
    if ((nz >> j) & 1ull) {  // a seed inside the chunk's 64 rows (wave-uniform): the bit scans
        unsigned long long wc;
        int pc, nc;
        desc_lane(dcur, j, wc, pc, nc);
        return column_value<false>(wc, pc, nc, lane, y);
    }
    const int pc = __builtin_amdgcn_readlane(dcur.prev, j), nc = __builtin_amdgcn_readlane(dcur.next, j);
    const int d = min(y - pc, nc - y);
    return d >= (1 << 29) ? FLT_MAX : (float)d;  // no seed in the whole column
}
__global__ void __launch_bounds__(256) k_l1_word_mins(const ColDesc* __restrict__ desc, float2* __restrict__ mins, int W, int HW64,
                                                      int nwords, long nwaves) {
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (wid >= nwaves) return;
    const long kc = wid / nwords;  // slice * HW64 + chunk
    const int w = (int)(wid - kc * nwords), c = (int)(kc % HW64), y = c * 64 + lane;
    const int x0 = w * 64, jn = min(64, W - x0);
    const ColDesc dcur = desc[(size_t)kc * W + min(x0 + lane, W - 1)];
    const unsigned long long valid = jn == 64 ? ~0ull : (1ull << jn) - 1ull;
    const unsigned long long nz = __builtin_amdgcn_ballot_w64(dcur.word != 0ull) & valid;
    // A column without a seed inside the chunk is worth min(y - prev, next - y) in every row y of the chunk, so over such
    // columns  min (c - x) = min(y - max (prev + x), min (next - x) - y)  and  min (c + x) = min(y - max (prev - x), min (next + x) - y):
    // four reductions over the word's descriptors (lane = column) and six instructions per row, for 64 columns x 18.  The
    // integers are exact (missing sides are 2^30 away; a result of that size means "no seed at all" = FLT_MAX, as the
    // per-column form says it).  Columns with a seed inside the chunk (one word in seven has any) go one by one.
    const bool far = lane < jn && dcur.word == 0ull;
    const int xj = x0 + lane;
    const int a1 = wave_max(far ? dcur.prev + xj : -(1 << 30)), a2 = wave_min(far ? dcur.next - xj : (1 << 30));
    const int b1 = wave_max(far ? dcur.prev - xj : -(1 << 30)), b2 = wave_min(far ? dcur.next + xj : (1 << 30));
    const int ai = min(y - a1, a2 - y), bi = min(y - b1, b2 - y);
    float a = ai >= (1 << 28) ? FLT_MAX : (float)ai, b = bi >= (1 << 28) ? FLT_MAX : (float)bi;
    for (unsigned long long m = nz; m; m &= m - 1ull) {
        const int j = __ffsll((long long)m) - 1;
        const float cq = l1_column_value(dcur, nz, j, lane, y), xf = (float)(x0 + j);
        a = std_min(a, cq - xf);
        b = std_min(b, cq + xf);
    }
    mins[(size_t)wid * 64 + lane] = make_float2(a, b);
}
// in place: .x <- min of the .x of the words before, .y <- min of the .y of the words behind (FLT_MAX where there is none)
__global__ void __launch_bounds__(256) k_l1_carries(float2* __restrict__ mins, int nwords, long nrows) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;  // (slice * HW64 + chunk) * 64 + row
    if (gid >= nrows) return;
    float2* m = mins + (size_t)(gid >> 6) * nwords * 64 + (gid & 63);
    float p = FLT_MAX;
    for (int w = 0; w < nwords; ++w) { const float t = m[(size_t)w * 64].x; m[(size_t)w * 64].x = p; p = std_min(p, t); }
    float q = FLT_MAX;
    for (int w = nwords - 1; w >= 0; --w) { const float t = m[(size_t)w * 64].y; m[(size_t)w * 64].y = q; q = std_min(q, t); }
}
template <bool FULL>  // FULL: the word has all 64 columns (every word but a row's last one)
__device__ __forceinline__ void l1_word(const ColDesc& dcur, unsigned long long nz, float2 carry, int x0, int jn, int lane, int y,
                                        __amdgpu_buffer_rsrc_t rs, unsigned vrow, int grpB) {
    float f[64];
    float run = carry.x + (float)(x0 - 1);  // F[x0 - 1]; FLT_MAX in front of the row's first column: min(c, FLT_MAX + 1) = c
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        f[j] = 0.f;  // (columns past W are padding and hold 0)
        if (FULL || j < jn) {
            run = std_min(l1_column_value(dcur, nz, j, lane, y), run + 1);
            f[j] = run;
        }
    }
    float r = carry.y - (float)(x0 + jn);  // B[x1 + 1]; FLT_MAX behind the row's last column: min(F, FLT_MAX + 1) = F
#pragma unroll
    for (int j = 63; j >= 0; --j) {
        if (FULL || j < jn) {
            r = std_min(f[j], r + 1);
            f[j] = r;
        }
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        if (FULL || 4 * g < jn) {
            u32x4 out;
            out.x = __float_as_uint(f[4 * g]); out.y = __float_as_uint(f[4 * g + 1]); out.z = __float_as_uint(f[4 * g + 2]); out.w = __float_as_uint(f[4 * g + 3]);
            __builtin_amdgcn_raw_buffer_store_b128(out, rs, vrow + (unsigned)(((x0 >> 2) + g) * grpB), 0, 0);  // (no scalar offset: see store_unit_note)
        }
    }
}
__global__ void __launch_bounds__(256) k_l1_word(const ColDesc* __restrict__ desc, const float2* __restrict__ carries, float* __restrict__ vol,
                                                 int W, int H, int HW64, int nwords, long nwaves) {
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (wid >= nwaves) return;
    const long kc = wid / nwords;
    const long k = kc / HW64;
    const int w = (int)(wid - kc * nwords), c = (int)(kc - k * HW64), y = c * 64 + lane;
    const int x0 = w * 64, jn = min(64, W - x0);
    const ColDesc dcur = desc[(size_t)kc * W + min(x0 + lane, W - 1)];
    const float2 carry = carries[(size_t)wid * 64 + lane];
    const unsigned long long nz = __builtin_amdgcn_ballot_w64(dcur.word != 0ull);
    const size_t sl = ivol_slice_floats(W, H);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(vol + (size_t)k * sl, 0, (unsigned)(sl * 4), 0x00020000);
    const unsigned vrow = y < H ? (unsigned)y * 16u : 0x80000000u;  // rows past the image: dropped stores
    if (jn == 64) l1_word<true>(dcur, nz, carry, x0, jn, lane, y, rs, vrow, H * 16);
    else l1_word<false>(dcur, nz, carry, x0, jn, lane, y, rs, vrow, H * 16);
}

__global__ void k_sqrt(float* __restrict__ vol, size_t n) {  // only for staged (test) builds
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) vol[i] = sqrtf(vol[i]);
}

// ------------------------------------------------------------------------------------------ K3
// propagateOrientation (dt3cpu.cpp:77-107): each pixel's m-vector is loaded once into LDS
// ([slice][thread], conflict free), the 4m steps S[c2] = min(S[c2], S[c1] + w) run there, and it
// is stored once.  L2's final sqrt (imgproc.h:191-192) is applied on load.
// Both variants read the y-fastest volume of the sweeps and write the interleaved volume (ivol_index) that the line
// integral and the search work on.  A thread is a position of the interleaved slice, q = (g*H + y)*4 + c for pixel
// (4g + c, y): 64 consecutive threads store 256 contiguous bytes and load 4 runs of 64 bytes (16 rows of 4 columns).
struct PropPixel {
    unsigned in_off, out_off;  // byte offsets inside a slice; 2^31: outside (columns of the last group past W)
};
__device__ __forceinline__ PropPixel prop_pixel(size_t q, int W, int H) {
    const unsigned g = (unsigned)(q / ((size_t)H * 4)), r = (unsigned)(q % ((size_t)H * 4));
    const unsigned y = r >> 2, x = 4 * g + (r & 3);
    PropPixel pp;
    pp.out_off = (unsigned)q * 4u;
    pp.in_off = x < (unsigned)W ? (x * (unsigned)H + y) * 4u : 0x80000000u;
    return pp;
}

__global__ void k_propagate(const float* __restrict__ vol, float* __restrict__ ivol, int W, int H, int m,
                            const PropStep* __restrict__ steps, int nsteps, int apply_sqrt) {
    extern __shared__ float S[];
    const int bd = blockDim.x, tid = threadIdx.x;
    const size_t q = (size_t)blockIdx.x * bd + tid, npix = (size_t)W * H, nq = ivol_slice_floats(W, H);
    const bool ok = q < nq;
    const PropPixel pp = prop_pixel(ok ? q : 0, W, H);
    const bool in_il = (apply_sqrt & 2) != 0;  // the transforms are in the interleaved layout already (segmented L2 sweep)
    const bool in = ok && (in_il || pp.in_off != 0x80000000u);
    for (int j = 0; j < m; ++j) {
        float v = in ? (in_il ? vol[(size_t)j * nq + q] : vol[(size_t)j * npix + pp.in_off / 4]) : 0.f;
        if (apply_sqrt & 1) v = sqrtf(v);
        S[j * bd + tid] = v;
    }
    for (int s = 0; s < nsteps; ++s) {
        const PropStep st = steps[s];
        const float a = S[st.c2 * bd + tid];
        const float b = S[st.c1 * bd + tid] + st.w;
        S[st.c2 * bd + tid] = std_min(a, b);
    }
    if (ok)
        for (int j = 0; j < m; ++j) ivol[(size_t)j * nq + q] = S[j * bd + tid];
}

// Register-resident variant for the common depths: the ring indices of propagateOrientation's
// 4M steps (dt3cpu.cpp:88-89) are compile-time constants, so the pixel's M-vector stays in VGPRs
// and the kernel is a pure stream (read V, write V) at full occupancy.
template <int M>
__global__ void __launch_bounds__(256) k_propagate_reg(const float* __restrict__ vol, float* __restrict__ ivol, int W, int H,
                                                       const PropStep* __restrict__ steps, int apply_sqrt) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x, npix = (size_t)W * H, nq = ivol_slice_floats(W, H);
    if (q >= nq) return;
    // One buffer descriptor per slice (scalar registers) + one 32-bit lane byte offset: addresses
    // cost no vector registers, so the M values are the kernel's whole register footprint.
    PropPixel pp = prop_pixel(q, W, H);  // slices are < 2^30 pixels
    const bool in_il = (apply_sqrt & 2) != 0;  // the transforms are in the interleaved layout already (segmented L2 sweep)
    if (in_il) pp.in_off = pp.out_off;
    apply_sqrt &= 1;
    const size_t in_slice = in_il ? nq : npix;
    const unsigned in_bytes = (unsigned)(in_slice * 4u), out_bytes = (unsigned)(nq * 4u);
    float S[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vol) + (size_t)j * in_slice, 0, in_bytes, 0x00020000);
        S[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, pp.in_off, 0, 0));
    }
    if (apply_sqrt) {
#pragma unroll
        for (int j = 0; j < M; ++j) S[j] = sqrtf(S[j]);
    }
    constexpr int FWD = (3 * M + 1) / 2;  // ceil(1.5 M)
    constexpr int BWD = (3 * M) / 2;      // floor(1.5 M)
    int s = 0;
#pragma unroll
    for (int c = 0; c < FWD; ++c, ++s) {  // propagate(0, ceil(1.5 m), +1)
        const int c1 = (M + ((c - 1) % M)) % M, c2 = (M + (c % M)) % M;
        S[c2] = std_min(S[c2], S[c1] + steps[s].w);
    }
#pragma unroll
    for (int i = 0; i < M + BWD; ++i, ++s) {  // propagate(m, -floor(1.5 m), -1): c = M - i
        const int c = M - i;
        const int c1 = (M + ((c + 1) % M)) % M, c2 = (M + (c % M)) % M;
        S[c2] = std_min(S[c2], S[c1] + steps[s].w);
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(ivol + (size_t)j * nq, 0, out_bytes, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(S[j]), rs, pp.out_off, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------ K4
// lineIntegral (imgproc.h:38-84).  The reference adds the previous (already integrated) line,
// shifted by dy_i = round(i r) - round((i-1) r), into the current one; the shifts telescope, so
// pixel (x_i, c + round(i r)) belongs to chain c and each chain is one sequential float32 sum, added in the
// reference's order.  Input and output are interleaved volumes (ivol_index: 16 bytes = 4 neighbouring columns of
// one row), the kernel reads one and writes the other, and every memory operation moves whole 16-byte units:
// a unit is stored by exactly one wave (shallow) or block (steep), which computes the up to three chains of a
// neighbour that cross its units itself (3 of 61 / 64 chains are such a halo) instead of sharing units.

// ---- shallow slices (mode 1: the sweep runs along x, a wave's chains are neighbouring rows)
// Per group of 4 columns (4 sweep steps) a wave issues one 16-byte load and one 16-byte store per lane: lane rho is
// the row where its chain sits at the group's first step.  During the group a chain moves on by 0 or 1 row per
// step (|r| <= 1), always in the direction of sign(r), so the running sums are kept in row coordinates: after a
// step that moves the chains the accumulator is shifted by one lane (DPP wave_shr), inputs and outputs need no
// shuffling, and at the end of the group the accumulator is shifted back by the group's total.
// Lane rho <-> chain a + sg*(rho - 3), sg = sign(r): lanes 0..2 are the halo (the chains that reach the wave's
// first rows during a group), lanes 3..60 the 58 chains whose rows the wave stores, lanes 61..63 would fall off
// the 64-row window after 3 moves and are not used.
#ifndef FDCM_SHP
#define FDCM_SHP 12
#endif
static constexpr int kShP = FDCM_SHP;                 // groups (loads of 1 KB) in flight per wave
static constexpr int kShOwn = 58, kShHalo = 3;
// Table per slice, one word per group in sweep order: 16 * (chain offset round(i r) at the group's first step) |
// bit j: the chains move between the group's steps j and j+1.  Steps outside the image (the padding columns of
// the last group) repeat the nearest offset.  Past the last group: 2^30, an out-of-range row for every lane.
__host__ __device__ inline int sh_tab_stride(int W) { return (((W + 3) / 4 + 2 * kShP) + 3) & ~3; }
__global__ void k_groups(const IntegralDesc* __restrict__ desc, int* __restrict__ tab, int W, int stride) {
    const int G = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (G >= stride) return;
    const IntegralDesc d = desc[k];
    const int W4 = (W + 3) / 4;
    int word = 0x40000000;
    if (d.mode == 1 && G < W4) {
        int o[4];
        for (int j = 0; j < 4; ++j) {
            const int x = d.s > 0 ? 4 * G + j : 4 * (W4 - 1 - G) + 3 - j;
            const int i = min(max(d.s > 0 ? x : W - 1 - x, 0), W - 1);      // sweep step of column x (imgproc.h:54-55)
            o[j] = (int)roundf((float)i * d.r);
        }
        word = o[0] * 16 | (o[1] != o[0] ? 1 : 0) | (o[2] != o[1] ? 2 : 0) | (o[3] != o[2] ? 4 : 0);
    }
    tab[(size_t)k * stride + G] = word;
}

__device__ __forceinline__ float lane_shr1(float v) {  // lane i <- lane i-1 (lane 0 <- 0)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_shl1(float v) {  // lane i <- lane i+1 (lane 63 <- 0)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

__device__ __forceinline__ void integral_shallow(const float* __restrict__ src, float* __restrict__ dst, int W, int H,
                                                 const IntegralDesc& d, int k, const int* __restrict__ tab, int shw) {
    constexpr int P = kShP;
    static_assert(P % 4 == 0, "the table is read four groups at a time");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W4 = (W + 3) >> 2;
    const int last_off = (int)roundf((float)(W - 1) * d.r);
    const int cmin = -max(0, last_off), cmax = H - 1 - min(0, last_off);
    if (wave >= shw) return;  // shw = 1 (one working wave per workgroup) or 4: see k_integral
    const int lo = cmin + ((int)blockIdx.x * shw + wave) * kShOwn;  // the wave stores the rows of chains lo .. lo + 57
    if (lo > cmax) return;
    const int sg = d.r < 0.f ? -1 : 1;
    const int a = sg > 0 ? lo : lo + kShOwn - 1;
    const int lanebase = (a + sg * (lane - kShHalo)) * 16;  // byte offset of the lane's row inside a group at chain offset 0
    const bool own = lane >= kShHalo && lane < kShHalo + kShOwn;
    const int* tb = tab + (size_t)k * sh_tab_stride(W);
    const size_t sl = ivol_slice_floats(W, H);
    const long gstride = (long)d.s * H * 4;  // floats from a group to the next one of the sweep
    const size_t g0 = d.s > 0 ? 0 : (size_t)(W4 - 1) * H * 4;
    const float* gf = src + (size_t)k * sl + g0;  // group of the next fetch
    float* gc = dst + (size_t)k * sl + g0;        // group of the next store
    const bool up = d.s > 0;                      // columns of a group in sweep order: x y z w, or w z y x
    constexpr unsigned OOB = 0x80000000u;
    u32x4 v[P];
    auto fetch = [&](int word, u32x4& r) {
        // one descriptor per group: base = the group, size = one group, so rows outside the image (and every row of
        // the groups past the end) read as 0 and their stores are dropped
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gf), 0, (unsigned)H * 16u, 0x00020000);
        r = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(lanebase + (word & ~15)), 0, 0);
        gf += gstride;
    };
#pragma unroll
    for (int p = 0; p < P; p += 4) {
        const int4 w4 = *reinterpret_cast<const int4*>(tb + p);
        fetch(w4.x, v[p]); fetch(w4.y, v[p + 1]); fetch(w4.z, v[p + 2]); fetch(w4.w, v[p + 3]);
    }
    // The first fetches are waited for here, once: the compiler orders them freely, and its wait at the loop header has
    // to cover the entry as well as the back edge -- with anything pending on entry it waits for (nearly) all memory
    // operations in every iteration.
#pragma unroll
    for (int p = 0; p < P; ++p) asm volatile("" : "+v"(v[p]));
    float acc = 0.f;
    for (int G0 = 0; G0 < W4; G0 += P) {
#pragma unroll
        for (int p = 0; p < P; p += 4) {
            const int4 wc = *reinterpret_cast<const int4*>(tb + G0 + p);
            const int4 wn = *reinterpret_cast<const int4*>(tb + G0 + p + P);
            const int wcur[4] = {wc.x, wc.y, wc.z, wc.w}, wnext[4] = {wn.x, wn.y, wn.z, wn.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int word = wcur[q];
                const u32x4 in = v[p + q];
                const float i0 = __uint_as_float(up ? in.x : in.w), i1 = __uint_as_float(up ? in.y : in.z),
                            i2 = __uint_as_float(up ? in.z : in.y), i3 = __uint_as_float(up ? in.w : in.x);
                // out-of-image elements are +0: 0 + acc == acc exactly (acc >= +0)
                acc = i0 + acc; const float o0 = acc; if (word & 1) acc = lane_shr1(acc);
                acc = i1 + acc; const float o1 = acc; if (word & 2) acc = lane_shr1(acc);
                acc = i2 + acc; const float o2 = acc; if (word & 4) acc = lane_shr1(acc);
                acc = i3 + acc; const float o3 = acc;
                const int moved = __builtin_popcount(word & 7);
                if (moved > 0) acc = lane_shl1(acc);
                if (moved > 1) acc = lane_shl1(acc);
                if (moved > 2) acc = lane_shl1(acc);
                u32x4 out;
                out.x = __float_as_uint(up ? o0 : o3); out.y = __float_as_uint(up ? o1 : o2);
                out.z = __float_as_uint(up ? o2 : o1); out.w = __float_as_uint(up ? o3 : o0);
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(gc, 0, (unsigned)H * 16u, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(out, rs, own ? (unsigned)(lanebase + (word & ~15)) : OOB, 0, 0);
                gc += gstride;
                fetch(wnext[q], v[p + q]);
            }
        }
    }
}

// ---- steep slices (mode 2: the sweep runs along y, chains run across x)
// They go through LDS tiles: a block computes 64 neighbouring chains, loads [64 + 32 (+4 to start on a group)
// columns] x [32 sweep steps] as 16-byte units (4 columns of one row), wave 0 runs the 64 sequential sums on the
// tile, and the units whose first column belongs to one of the block's first 60 chains are stored (their other
// three columns belong to the next three chains at most: the halo).  The next tile's loads are in flight while
// the current one is summed and stored.
// XC chains per block (XC / 64 waves run them), of which the first XC - 4 are the block's own: 64 while the launch is
// small (more blocks, shorter critical path), 256 when the slices are large -- a tile is XC + drift + 4 columns wide,
// so the columns read per column stored fall from (64 + 36) / 60 = 1.67 to (256 + 36) / 252 = 1.16, and the drift is
// sized per slice: the chains of a slice move by at most ceil(31 |r|) + 1 columns over a tile's 32 steps, not by 32.
template <int XC>
__device__ __forceinline__ void integral_steep(const float* __restrict__ src, float* __restrict__ dst, int W, int H,
                                               const IntegralDesc& d, int k, float* lds_tiles) {
    constexpr int TS = 32, TW = XC + TS + 4, NG = TW / 4, PASSES = (NG + 7) / 8, OWN = XC - 4;
    float (*tile)[TW][TS + 1] = reinterpret_cast<float (*)[TW][TS + 1]>(lds_tiles);  // [2][TW][TS + 1]
    const int steps = H, span = W, W4 = (W + 3) >> 2;
    const int last_off = (int)roundf((float)(steps - 1) * d.r);
    const int cmin = -max(0, last_off), cmax = span - 1 - min(0, last_off);
    const int c0 = cmin + (int)blockIdx.x * OWN;
    if (c0 > cmax) return;
    const int start = d.s < 0 ? steps - 1 : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int prow = tid & 31, pgrp = tid >> 5;  // load/store mapping: 8 groups x 32 rows per pass
    const int ntiles = (steps + TS - 1) / TS;
    // groups of a tile that this slice can touch: chains + the drift over TS - 1 steps (|round(a) - round(b)| <=
    // ceil(|a - b|) + 1) + up to 3 columns in front of the first chain (tiles start on a group)
    const int ng = min(NG, (XC + (int)ceilf((float)(TS - 1) * fabsf(d.r)) + 1 + 3 + 3) >> 2);
    // Units outside the image get an out-of-range offset: loads return 0, stores are dropped, and no memory
    // operation sits behind a branch, so the compiler counts them exactly and the loads of two tiles stay in
    // flight behind the stores of the previous ones.
    const size_t sl = ivol_slice_floats(W, H);
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src) + (size_t)k * sl, 0, (unsigned)(sl * 4), 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)k * sl, 0, (unsigned)(sl * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    auto off_at = [&](int i) { return (int)roundf((float)i * d.r); };  // chain offset at step i (imgproc.h:70-71)
    // first column of tile t: the leftmost column of the block's chains over the tile's steps, rounded down to a group
    auto xbase = [&](int t) { return (c0 + min(off_at(t * TS), off_at(min(t * TS + TS - 1, steps - 1)))) & ~3; };
    auto unit_off = [&](int xg, int i) {  // byte offset of the unit of group xg at sweep step i
        return (xg >= 0 && xg < W4 && i < steps) ? (unsigned)((xg * H + start + i * d.s) << 4) : OOB;
    };
    auto load_tile = [&](int t, u32x4 (&regs)[PASSES]) {
        const int i = t * TS + prow, xg0 = xbase(t) >> 2;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int g = p * 8 + pgrp;
            regs[p] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, g < ng ? unit_off(xg0 + g, i) : OOB, 0, 0);
        }
    };
    float acc = 0.f;
    auto process = [&](int t, u32x4 (&regs)[PASSES]) {  // regs hold tile t on entry, tile t + 2 on exit
        const int buf = t & 1, i0 = t * TS, xb = xbase(t);
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int g = p * 8 + pgrp;
            if (g < NG) {
                tile[buf][4 * g + 0][prow] = __uint_as_float(regs[p].x);
                tile[buf][4 * g + 1][prow] = __uint_as_float(regs[p].y);
                tile[buf][4 * g + 2][prow] = __uint_as_float(regs[p].z);
                tile[buf][4 * g + 3][prow] = __uint_as_float(regs[p].w);
            }
        }
        __syncthreads();
        load_tile(t + 2, regs);  // past the last tile every load is out of range
        if (wave < XC / 64) {
            // Chain c0 + ch, ch = 64 wave + lane.  At step ii it sits in tile column ch + off_ii - (xb - c0): always
            // inside the tile.  No validity test is needed here: elements outside the image or past the last
            // step were loaded as +0 (acc + 0 == acc exactly) and every tile element belongs to exactly one chain.
            const int ch = wave * 64 + lane;
            const int my_d = off_at(min(i0 + (lane & 31), steps - 1)) - (xb - c0);  // lane j < 32: step i0 + j
            // all 32 reads are issued before the dependent chain of adds (they never alias: one element per
            // step), so the chain costs 32 adds, not 32 LDS round trips
            float v[TS];
            float* cell[TS];
#pragma unroll
            for (int ii = 0; ii < TS; ++ii) {
                cell[ii] = &tile[buf][ch + __builtin_amdgcn_readlane(my_d, ii)][ii];
                v[ii] = *cell[ii];
            }
#pragma unroll
            for (int ii = 0; ii < TS; ++ii) {
                acc = v[ii] + acc;
                *cell[ii] = acc;
            }
        }
        __syncthreads();
        {  // a unit is stored by the block that owns the chain of its first column
            const int i = i0 + prow;
            const int o = off_at(min(i, steps - 1));
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int g = p * 8 + pgrp;
                const int fc = xb + 4 * g - o;  // chain of the unit's first column
                u32x4 out;
                const int gg = g < NG ? g : 0;
                out.x = __float_as_uint(tile[buf][4 * gg + 0][prow]); out.y = __float_as_uint(tile[buf][4 * gg + 1][prow]);
                out.z = __float_as_uint(tile[buf][4 * gg + 2][prow]); out.w = __float_as_uint(tile[buf][4 * gg + 3][prow]);
                const bool mine = g < ng && fc >= c0 && fc < c0 + OWN;
                __builtin_amdgcn_raw_buffer_store_b128(out, rs_out, mine ? unit_off((xb >> 2) + g, i) : OOB, 0, 0);
            }
        }
    };
    u32x4 ra[PASSES], rb[PASSES];
    load_tile(0, ra);
    load_tile(1, rb);
    for (int t = 0; t < ntiles; t += 2) {
        process(t, ra);
        if (t + 1 < ntiles) process(t + 1, rb);
    }
}
template <int XC>
constexpr size_t integral_lds_bytes() { return (size_t)2 * (XC + 32 + 4) * 33 * sizeof(float); }

// One launch for all slices: blockIdx.y = slice, and the slice's mode picks the sweep.  Shallow and
// steep slices are independent, so their (latency-bound) blocks overlap instead of running as two
// kernels back to back.  A workgroup takes 60 chains of a steep slice, or 58 chains of a shallow one on one wave
// (the other three exit at once) while the launch is small: a CU can only have so many cache misses outstanding, and
// four such waves on one CU (105 of 256 CUs busy at config 2) ran at 0.061 ms where one per workgroup, spread over
// all CUs, runs at 0.051.  Large launches (config 5: 25 000 workgroups) fill every CU anyway and put 4 x 58 chains on a
// workgroup (5.9 against 6.4 ms).
template <int XC>
__global__ void __launch_bounds__(256) k_integral(const float* __restrict__ src, float* __restrict__ dst, int W, int H,
                                                  const IntegralDesc* __restrict__ desc,
                                                  const int* __restrict__ tab, int shw, int kstride) {
    extern __shared__ float lds_tiles[];
    // Slices are taken in a strided order (kstride is coprime to the slice count and close to half of it): in index order
    // all steep slices of one angular range run before the shallow ones, and the two kinds stress different things (LDS
    // tiles against straight 1 KB streams), so mixing them over the launch overlaps them.
    const int k = (int)(((long)blockIdx.y * kstride) % (long)gridDim.y);
    const IntegralDesc d = desc[k];
#ifdef FDCM_LAB
    // lab builds, FDCM_INT_ONLY=shallow|steep: only that class of slices runs (PMC traffic per class: tools/int_split.sh);
    // the flag rides in kstride's upper bits
    const int only = kstride >> 24;
    kstride &= 0xffffff;
    const int k2 = (int)(((long)blockIdx.y * kstride) % (long)gridDim.y);
    const IntegralDesc d2 = desc[k2];
    if (only && d2.mode != only) return;
    if (only) { if (d2.mode == 1) integral_shallow(src, dst, W, H, d2, k2, tab, shw); else integral_steep<XC>(src, dst, W, H, d2, k2, lds_tiles); return; }
#endif
    if (d.mode == 1) integral_shallow(src, dst, W, H, d, k, tab, shw);
    else if (d.mode == 2) integral_steep<XC>(src, dst, W, H, d, k, lds_tiles);
    else {  // nothing to integrate (imgproc.h:43): the slice moves as it is
        const size_t sl = ivol_slice_floats(W, H);
        for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < sl; q += (size_t)gridDim.x * 256)
            dst[(size_t)k * sl + q] = src[(size_t)k * sl + q];
    }
}

// ------------------------------------------------------------------------------------------ driver
static void ensure_timing(fdcm_featuremap* fm) {
    if (fm->timing.created) return;
    for (auto& e : fm->timing.ev) FDCM_HIP(hipEventCreate(&e));
    fm->timing.created = true;
}

// builds of this process whose L2 sweep took its launch order from the handle's previous build / from the host's proxy
static std::atomic<int64_t> g_order_from_history{0}, g_order_from_proxy{0};
void sweep_order_counts(int64_t* from_history, int64_t* from_proxy) {
    *from_history = g_order_from_history.load();
    *from_proxy = g_order_from_proxy.load();
}

void run_build(fdcm_featuremap* fm, const BuildPlan& plan, int stop_after, bool reserve_only) {
    const auto t0 = std::chrono::steady_clock::now();
    FDCM_HIP(hipSetDevice(fm->device));
    if (!fm->stream) FDCM_HIP(hipStreamCreateWithFlags(&fm->stream, hipStreamNonBlocking));
    ensure_timing(fm);
    finish_build(fm);  // the previous build's staging and events are reused below
    hipStream_t st = fm->stream;
    // A reservation only grows buffers: the handle's geometry, the plan offsets of its last build and the sweep's cost
    // history stay as they are (restored where the function returns early for it); its content survives unless a volume
    // buffer had to grow, and the cost history unless the scratch that holds it did.
    const long kept_cost_chunks = fm->k2_cost_chunks;
    const int kept_cost_w = fm->k2_cost_w;
    const void* const kept_stack = fm->stack.p;
    const bool kept_v1 = fm->vol1_interleaved;
    const size_t kept_off[6] = {fm->off_raster, fm->off_prop, fm->off_integral, fm->off_keys, fm->off_slice, fm->off_cost};
    if (!reserve_only) {
        fm->W = plan.W; fm->H = plan.H; fm->m = plan.m; fm->tx = plan.tx; fm->ty = plan.ty;
        fm->keys = plan.keys;
        fm->last_build = fdcm_build_timing{};
    }
    if (plan.m == 0 || plan.W == 0) return;
    const int W = (int)plan.W, H = (int)plan.H, m = (int)plan.m;
    if (plan.W > 16384 || plan.H > 16384) throw std::string("feature size above 16384 is not supported");  // 32-bit byte offsets inside a slice
    const int HW64 = (H + 63) / 64;
    const size_t npix = (size_t)W * H, nvox = npix * m;
    const long ncols = (long)m * W;
    fm->vol.reserve(std::max(nvox, (size_t)m * ivol_slice_floats(W, H)) * sizeof(float));  // the integrated volume comes back here, interleaved
    if (HW64 > 64) fm->bitmap.reserve((size_t)ncols * HW64 * 8);  // (feature sizes above 4096 only: k_seeds + k_coldesc)
    // every buffer of the build is reserved here, before the first kernel is queued: an allocation between two stages
    // (a handle's first build) stalls the host for 0.5 - 1 ms while the GPU idles inside the stage events' span
    if (stop_after >= 2) fm->ivol.reserve((size_t)m * ivol_slice_floats(W, H) * sizeof(float));
    if (stop_after >= 3) fm->offtab.reserve((size_t)m * sh_tab_stride(W) * sizeof(int));
    const long nchunks = (long)m * HW64;  // (slice, 64-row chunk) pairs
    // Which L2 / L2^2 sweep: ranges of equal column count, merged (fdcm_sweep.hip) where every value of the pass is an exact
    // integer, the literal pass one wave per chunk (fdcm_sweep_literal.hip) otherwise.  FDCM_L2_SWEEP=literal is the tests'
    // switch for the latter at every size.
    static const bool env_literal = getenv("FDCM_L2_SWEEP") != nullptr && std::strcmp(getenv("FDCM_L2_SWEEP"), "literal") == 0;
    const bool l2 = fm->distance != FDCM_L1;
    const bool balanced = l2 && HW64 <= 64 && sweep_balanced_applies(W, H) && !env_literal;
    fm->vol1_interleaved = true;  // every sweep writes the transforms in the interleaved layout (ivol_index) the propagation reads
    fm->coldesc.reserve((size_t)ncols * HW64 * sizeof(ColDesc));
    fm->colmask.reserve((size_t)m * ((W + 63) / 64) * 8);
    SweepBuf sb{};
    bool proxy_order = false;
    std::vector<int32_t> proxy_cost;
    int* order_dst = nullptr;
    if (fm->distance == FDCM_L1) {
        fm->stack.reserve((size_t)m * HW64 * ((W + 63) / 64) * 64 * sizeof(float2));  // the L1 pass's minima / carries
    } else if (!balanced) {
        fm->stack.reserve(sweep_literal_scratch_bytes(W, nchunks));
        fm->k2_cost_chunks = 0;
    } else {
        // scratch of the balanced sweep, row-major: stack entries (W + 2 slots per row, 12 B), owner list (W + 2 entries per
        // row, 8 B), launch order and per-chunk cost
        const size_t NRr = (size_t)nchunks * 64, slots = (size_t)W + 2;
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
        const size_t o_ent = take(slots * NRr * sizeof(EnvEntry)), o_own = take(slots * NRr * sizeof(OwnEntry));
        const size_t o_ord = take((size_t)nchunks * 4), o_cost = take((size_t)nchunks * 4), o_steals = take(256);
        const void* stack_before = fm->stack.p;
        fm->stack.reserve(off);
        char* sp = (char*)fm->stack.p;
        // Launch order: workgroups are dispatched in index order, and when there are more of them than the GPU holds at once
        // (two per CU) the long ones must not start last.  Nothing cheap predicts a chunk's time well enough, the previous
        // build of the same shape does: scenes of a stream change little from frame to frame.  A handle's first build, and
        // every build after a change of size, takes the host's proxy per chunk (make_build_plan), which arrives with the plan.
        static const bool env_order = getenv("FDCM_SWEEP_ORDER") != nullptr;  // the tests' switch: the launch order at every size
        const bool want_order = env_order || nchunks > 2L * device_cus(fm->device);
        const bool have_cost = want_order && fm->k2_cost_chunks == nchunks && fm->k2_cost_w == W && stack_before == fm->stack.p;
        proxy_order = want_order && !have_cost && !reserve_only;
        if (proxy_order) sweep_cost_proxy(plan, proxy_cost);
        if (have_cost && !reserve_only) launch_sweep_order(st, (const int*)(sp + o_cost), (int)nchunks, (int*)(sp + o_ord));
        order_dst = (int*)(sp + o_ord);
        sb.ent = (EnvEntry*)(sp + o_ent); sb.own = (OwnEntry*)(sp + o_own);
        sb.order = (have_cost || proxy_order) ? (const int*)(sp + o_ord) : nullptr;
        sb.cost = (int*)(sp + o_cost);
        // Dynamic cuts (a wave out of columns begins a new range in what nobody has started): they shorten the heaviest
        // workgroup's chain and add junctions, i.e. work -- worth it where the kernel lasts as long as its slowest workgroup
        // (all workgroups resident at once, the GPU to this handle), not where workgroups queue for the CUs or frames of a
        // pipeline share them (config 2: one blocking build 0.354 -> 0.333 ms over four scenes; four frames in flight 69.7 ->
        // 68.4 M matches/s; config 3: 0.76 -> 0.81 ms).  FDCM_SWEEP_STEAL=<blocks> forces a threshold (0: never) for the tests.
        sb.steal_min = -1;  // the kernel's default threshold
        sb.steal_heavy_only = (!fm->shares_gpu && nchunks <= 2L * device_cus(fm->device)) ? 0 : 1;
        sb.steals = (int*)(sp + o_steals);
        if (!reserve_only) {
            if (fm->sweep_steals != sb.steals) FDCM_HIP(hipMemsetAsync(sp + o_steals, 0, 256, st));  // a new scratch (or shape): count from 0
            fm->sweep_steals = sb.steals;
        }
        sb.eslots = (int)slots; sb.lslots = (int)slots;
        sb.colmask = (const unsigned long long*)fm->colmask.p;
        fm->k2_cost_chunks = nchunks; fm->k2_cost_w = W;
    }

    // ---- plan upload: one pinned blob, one async copy
    auto align16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    fm->off_raster = 0;
    fm->off_prop = align16(plan.raster.size() * sizeof(RasterLine));
    fm->off_integral = fm->off_prop + align16(plan.prop.size() * sizeof(PropStep));
    fm->off_keys = fm->off_integral + align16(plan.integral.size() * sizeof(IntegralDesc));
    fm->off_slice = fm->off_keys + align16(plan.keys.size() * sizeof(float));
    fm->off_cost = fm->off_slice + align16(plan.slice_first.size() * sizeof(int32_t));
    const size_t blob = fm->off_cost + ((proxy_order || reserve_only) ? align16((size_t)nchunks * sizeof(int32_t)) : 0);
    fm->stage.reserve(blob);
    fm->plan.reserve(blob);
    if (reserve_only) {  // every buffer a build of this plan's shape takes is in place; nothing was queued
        fm->off_raster = kept_off[0]; fm->off_prop = kept_off[1]; fm->off_integral = kept_off[2];
        fm->off_keys = kept_off[3]; fm->off_slice = kept_off[4]; fm->off_cost = kept_off[5];
        fm->vol1_interleaved = kept_v1;
        if (fm->stack.p == kept_stack) { fm->k2_cost_chunks = kept_cost_chunks; fm->k2_cost_w = kept_cost_w; }
        else fm->k2_cost_chunks = 0;  // (a new scratch: its cost table holds nothing yet)
        return;
    }
    if (sb.order) (proxy_order ? g_order_from_proxy : g_order_from_history).fetch_add(1, std::memory_order_relaxed);
    char* hs = (char*)fm->stage.p;
    if (!plan.raster.empty()) std::memcpy(hs + fm->off_raster, plan.raster.data(), plan.raster.size() * sizeof(RasterLine));
    std::memcpy(hs + fm->off_prop, plan.prop.data(), plan.prop.size() * sizeof(PropStep));
    std::memcpy(hs + fm->off_integral, plan.integral.data(), plan.integral.size() * sizeof(IntegralDesc));
    std::memcpy(hs + fm->off_keys, plan.keys.data(), plan.keys.size() * sizeof(float));
    std::memcpy(hs + fm->off_slice, plan.slice_first.data(), plan.slice_first.size() * sizeof(int32_t));
    if (proxy_order) std::memcpy(hs + fm->off_cost, proxy_cost.data(), proxy_cost.size() * sizeof(int32_t));
    FDCM_HIP(hipMemcpyAsync(fm->plan.p, hs, blob, hipMemcpyHostToDevice, st));
    if (proxy_order) launch_sweep_order(st, (const int*)((const char*)fm->plan.p + fm->off_cost), (int)nchunks, order_dst);
    fm->n_raster = (int64_t)plan.raster.size();
    fm->n_prop = (int64_t)plan.prop.size();
    const char* dp = (const char*)fm->plan.p;
    const RasterLine* d_raster = (const RasterLine*)(dp + fm->off_raster);
    const PropStep* d_prop = (const PropStep*)(dp + fm->off_prop);
    const IntegralDesc* d_int = (const IntegralDesc*)(dp + fm->off_integral);
    float* vol = fm->vol.as<float>();
    hipEvent_t* ev = fm->timing.ev;

    fm->build_host_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    fm->stage_events = fm->want_stage_events == 1;  // (an event between two kernels costs a blocking frame 3 - 5 us: fdcm_featuremap_stage_timing)
    fm->total_events = fm->want_stage_events != 0;
    if (fm->total_events) FDCM_HIP(hipEventRecord(ev[0], st));
    ColDesc* d_desc = fm->coldesc.as<ColDesc>();
    if (HW64 <= 64) {
        // the tile kernel rasterises the seeds of its columns itself (LDS): no bitmap, no k_seeds, no stage of its own
        fm->seeds_fused = true;
        const int* d_first = (const int*)(dp + fm->off_slice);
        const int XT = HW64 > 32 ? 32 : 64;
        const dim3 grid((unsigned)((W + XT - 1) / XT), (unsigned)m);
        const size_t lds = (size_t)HW64 * (XT + 1) * sizeof(uint4) + (size_t)HW64 * XT * 8;
        unsigned* cm = (unsigned*)fm->colmask.p;
        if (HW64 <= 16) hipLaunchKernelGGL((k_coldesc_tile<16, 64>), grid, dim3(256), lds, st, d_raster, d_first, d_desc, W, H, HW64, cm);
        else if (HW64 <= 32) hipLaunchKernelGGL((k_coldesc_tile<32, 64>), grid, dim3(256), lds, st, d_raster, d_first, d_desc, W, H, HW64, cm);
        else hipLaunchKernelGGL((k_coldesc_tile<64, 32>), grid, dim3(256), lds, st, d_raster, d_first, d_desc, W, H, HW64, cm);
    } else {
        fm->seeds_fused = false;
        const long bitmap_words = ncols * HW64;
        FDCM_HIP(hipMemsetAsync(fm->bitmap.p, 0, (size_t)bitmap_words * 8, st));
        if (fm->n_raster > 0)
            hipLaunchKernelGGL(k_seeds, dim3((unsigned)fm->n_raster), dim3(256), 0, st, d_raster,
                               fm->bitmap.as<unsigned long long>(), W, H, HW64);
        if (fm->stage_events) FDCM_HIP(hipEventRecord(ev[1], st));
        hipLaunchKernelGGL(k_coldesc, dim3((unsigned)((ncols + 3) / 4)), dim3(256), 0, st,
                           fm->bitmap.as<unsigned long long>(), d_desc, W, HW64, ncols);
    }
    if (fm->stage_events) FDCM_HIP(hipEventRecord(ev[2], st));
    if (fm->distance == FDCM_L1) {
        // both L1 sweeps with one pass over the volume: minima per (row, word), their prefix / suffix over the row's words, then word by word
        const int nwords = (W + 63) / 64;
        const long wwaves = (long)m * HW64 * nwords;
        float2* mins = (float2*)fm->stack.p;  // reserved above
        hipLaunchKernelGGL(k_l1_word_mins, dim3((unsigned)((wwaves + 3) / 4)), dim3(256), 0, st, d_desc, mins, W, HW64, nwords, wwaves);
        hipLaunchKernelGGL(k_l1_carries, dim3((unsigned)(((long)m * HW64 * 64 + 255) / 256)), dim3(256), 0, st, mins, nwords, (long)m * HW64 * 64);
        hipLaunchKernelGGL(k_l1_word, dim3((unsigned)((wwaves + 3) / 4)), dim3(256), 0, st, d_desc, (const float2*)mins, vol, W, H, HW64, nwords, wwaves);
    } else if (balanced) {
#ifdef FDCM_LAB
        if (!lab_skip("sweep"))
#endif
        launch_sweep_balanced(st, d_desc, vol, W, H, HW64, nchunks, sb);
    } else {
        launch_sweep_literal(st, d_desc, vol, W, H, HW64, nchunks, fm->stack.p);
    }
    if (fm->stage_events) FDCM_HIP(hipEventRecord(ev[3], st));
    const bool want_sqrt = fm->distance == FDCM_L2;
    if (stop_after >= 2) {
        const size_t nq = ivol_slice_floats(W, H);
        fm->ivol.reserve((size_t)m * nq * sizeof(float));
        float* ivol = fm->ivol.as<float>();
        const unsigned pblocks = (unsigned)((nq + 255) / 256);
        const int sq = (want_sqrt ? 1 : 0) | (fm->vol1_interleaved ? 2 : 0);
#ifdef FDCM_LAB
        if (lab_skip("propagate")) {} else
#endif
        if (m == 30) hipLaunchKernelGGL(k_propagate_reg<30>, dim3(pblocks), dim3(256), 0, st, (const float*)vol, ivol, W, H, d_prop, sq);
        else if (m == 60) hipLaunchKernelGGL(k_propagate_reg<60>, dim3(pblocks), dim3(256), 0, st, (const float*)vol, ivol, W, H, d_prop, sq);
        else if (m == 90) hipLaunchKernelGGL(k_propagate_reg<90>, dim3(pblocks), dim3(256), 0, st, (const float*)vol, ivol, W, H, d_prop, sq);
        else if (m == 120) hipLaunchKernelGGL(k_propagate_reg<120>, dim3(pblocks), dim3(256), 0, st, (const float*)vol, ivol, W, H, d_prop, sq);
        else if (m == 180) hipLaunchKernelGGL(k_propagate_reg<180>, dim3(pblocks), dim3(256), 0, st, (const float*)vol, ivol, W, H, d_prop, sq);
        else {
            int bd = 256;
            while (bd > 64 && (size_t)m * bd * sizeof(float) > 64 * 1024) bd >>= 1;
            const size_t lds = (size_t)m * bd * sizeof(float);
            if (lds > 64 * 1024)
                FDCM_HIP(hipFuncSetAttribute((const void*)k_propagate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_propagate, dim3((unsigned)((nq + bd - 1) / bd)), dim3(bd), lds, st, (const float*)vol, ivol, W, H, m,
                               d_prop, (int)fm->n_prop, sq);
        }
    } else if (want_sqrt) {
        const size_t nel = fm->vol1_interleaved ? (size_t)m * ivol_slice_floats(W, H) : nvox;  // (padding elements: harmless)
        hipLaunchKernelGGL(k_sqrt, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, vol, nel);
    }
    if (fm->stage_events) FDCM_HIP(hipEventRecord(ev[4], st));
    if (stop_after >= 3) {
        const int chains = 2 * (W > H ? W : H);
        const int tab_stride = sh_tab_stride(W);
        fm->offtab.reserve((size_t)m * tab_stride * sizeof(int));
        int* d_tab = fm->offtab.as<int>();
        if (!(fm->off_m == m && fm->off_steps == W)) {  // the table only depends on the keys (fixed per handle) and the size
            hipLaunchKernelGGL(k_groups, dim3((unsigned)((tab_stride + 255) / 256), (unsigned)m), dim3(256), 0, st, d_int, d_tab, W, tab_stride);
            fm->off_m = m; fm->off_steps = W;
        }
        int shw = (long)m * ((chains + kShOwn - 1) / kShOwn) > 8192 ? 4 : 1;  // working waves per workgroup of a shallow slice
        // steep slices: 60 own chains per block while the launch is small, 124 / 252 once such blocks would outnumber
        // what the GPU holds several times over (fewer columns read twice; see integral_steep).  FDCM_INT_XC=64|128|256 is
        // the tests' switch (the wide forms are only selected by large volumes).
        static const int env_int_xc = [] { const char* e = getenv("FDCM_INT_XC"); const int v = e ? atoi(e) : 0; return (v == 64 || v == 128 || v == 256) ? v : 0; }();
        const long narrow_blocks = (long)m * ((chains + 59) / 60), cus = device_cus(fm->device);
        const int xc = env_int_xc ? env_int_xc : (narrow_blocks > 64 * cus ? 256 : (narrow_blocks > 12 * cus ? 128 : 64));
        const dim3 igrid((unsigned)((chains + kShOwn - 1) / kShOwn), (unsigned)m);
        int kstride = 1;
        // slices are visited in a strided order (coprime to the depth, near half of it) so that steep and shallow ones
        // overlap -- once the blocks queue for the CUs (config 3: 4 260 blocks, 0.45 - 0.49 against 0.50 - 0.51 ms); a small
        // launch (config 2: 1 080 blocks, four per CU) is faster in index order (0.070 against 0.077 ms)
        const bool small_launch = (long)m * igrid.x <= 6L * cus;
        if (m > 2 && !small_launch) {
            auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
            kstride = m / 2 + 1;
            while (gcd(kstride, m) != 1) ++kstride;
        }
#ifdef FDCM_LAB
        static const int env_int_only = [] { const char* e = getenv("FDCM_INT_ONLY"); return !e ? 0 : (!std::strcmp(e, "shallow") ? 1 : (!std::strcmp(e, "steep") ? 2 : 0)); }();
        kstride |= env_int_only << 24;
#endif
#define FDCM_INTEGRAL(XC)                                                                                                        \
        do {                                                                                                                     \
            constexpr size_t lds = integral_lds_bytes<XC>();                                                                     \
            static_assert(lds <= 160 * 1024, "tile pair must fit a CU's LDS");                                                   \
            if (lds > 64 * 1024)                                                                                                 \
                FDCM_HIP(hipFuncSetAttribute((const void*)k_integral<XC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL(k_integral<XC>, igrid, dim3(256), lds, st, (const float*)fm->ivol.as<float>(), vol, W, H, d_int,  \
                               d_tab, shw, kstride);                                                               \
        } while (0)
#ifdef FDCM_LAB
        if (lab_skip("integral")) {} else
#endif
        if (xc == 256) FDCM_INTEGRAL(256); else if (xc == 128) FDCM_INTEGRAL(128); else FDCM_INTEGRAL(64);
#undef FDCM_INTEGRAL
    }
    fm->vol_stage = stop_after >= 3 ? 3 : (stop_after == 2 ? 2 : 1);
    if (fm->total_events) FDCM_HIP(hipEventRecord(ev[5], st));
    FDCM_HIP(hipGetLastError());
    fm->build_pending = true;  // not waited for here: see finish_build
}

void finish_build(fdcm_featuremap* fm) {
    if (!fm->build_pending) return;
    fm->build_pending = false;
    FDCM_HIP(hipSetDevice(fm->device));
    FDCM_HIP(hipStreamSynchronize(fm->stream));
    hipEvent_t* ev = fm->timing.ev;
    fdcm_build_timing& bt = fm->last_build;
    if (fm->stage_events) {
        if (fm->seeds_fused) {
            FDCM_HIP(hipEventElapsedTime(&bt.pass1_ms, ev[0], ev[2]));  // (seeds_ms stays 0: k_coldesc_tile draws them)
        } else {
            FDCM_HIP(hipEventElapsedTime(&bt.seeds_ms, ev[0], ev[1]));
            FDCM_HIP(hipEventElapsedTime(&bt.pass1_ms, ev[1], ev[2]));
        }
        FDCM_HIP(hipEventElapsedTime(&bt.pass2_ms, ev[2], ev[3]));
        FDCM_HIP(hipEventElapsedTime(&bt.propagate_ms, ev[3], ev[4]));
        FDCM_HIP(hipEventElapsedTime(&bt.integral_ms, ev[4], ev[5]));
    }
    float span = 0.f;
    if (fm->total_events) FDCM_HIP(hipEventElapsedTime(&span, ev[0], ev[5]));
    bt.span_ms = span;
    bt.total_ms = fm->build_host_ms + span;  // host preparation + the kernels' span on the device
}

}  // namespace fdcm
