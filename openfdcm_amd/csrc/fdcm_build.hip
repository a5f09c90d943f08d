// fdcm_build.hip -- DT3 feature-map build on gfx950 (buildCpuFeaturemap<D>, dt3cpu.h:174-234).
//
// Volume layout in HBM: the sweeps write float vol[k][x][y] (y fastest: the reference's RawImage<float>(H, W)
// column-major, math.h:57); the propagation moves it into the interleaved layout [k][x/4][y][x%4] (ivol_index,
// fdcm_internal.h) that the line integral and the search work on.
//
// Kernels (W x H = feature size, m = slices, V = 4*m*W*H bytes):
//   K0 k_seeds          clipped scene lines -> seed bitmap (1 bit per pixel, bits along y)   ~V/32
//   K1 k_coldesc_tile   per column and 64-row chunk: seed bits + nearest seed before/after   ~V/16
//   K2 k_sweep          L2 / L2^2: pass 1 (from descriptors) fused into the literal in-place lower-envelope pass
//                       along x (imgproc.h:91-130), rows cut into verified segments; k_pass2_l2: the
//                       one-wave-per-chunk form (redo path, FDCM_K2_LEGACY)                   write V
//      k_l1_*           L1: forward sweep from descriptors (write V), backward sweep (read V, write V)
//   K3 k_propagate_reg<M> / k_propagate   orientation propagation, 4m steps per pixel in
//                       registers (generic depth: LDS) (+ sqrt for L2)                        read V, write V
//   K4 k_integral       directional prefix sum per slice, one sequential float chain per row / column of 16-byte
//                       units; shallow and steep sweeps (the latter through LDS tiles) in one launch  read V, write V
// Compiled with -ffp-contract=off; divide and sqrt are the correctly rounded forms.
#include <algorithm>
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
template <int SEG, int XT>  // XT columns per block: 64, or 32 when a column has 64 words (32 KB of LDS instead of 64)
__global__ void __launch_bounds__(256) k_coldesc_tile(unsigned long long* __restrict__ bitmap, ColDesc* __restrict__ desc,
                                                      int W, int HW64, unsigned* __restrict__ colmask) {
    extern __shared__ uint4 tile[];  // [word][XT columns], rows padded by one unit (bank spread)
    constexpr int STR = XT + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = lane & (SEG - 1), ci = lane / SEG;
    constexpr int CPP = 64 / SEG, CPW = XT / 4;  // columns per wave pass, columns per wave
    const long k = blockIdx.y;
    const int x0 = blockIdx.x * XT;
    for (int pass = 0; pass < CPW / CPP; ++pass) {
        const int xl = wave * CPW + pass * CPP + ci, x = x0 + xl;
        const bool valid = x < W && wi < HW64;
        unsigned long long* bw = bitmap + ((size_t)k * W + (size_t)min(x, W - 1)) * HW64;
        const unsigned long long word = valid ? bw[wi] : 0ull;
        if (word) bw[wi] = 0ull;
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

// ------------------------------------------------------------------------------------------ K2
// Both 1-D passes of distanceTransform<float, L2 / L2_SQUARED> (imgproc.h:178-183) in one sweep
// along x.  One wave per (slice k, 64-row chunk c[, sub-block of R rows]); lane = row.  The
// pass-1 value of column q is recomputed from the column descriptor (64 columns staged in LDS per
// 1 KiB load), so the sweep reads V/16 instead of V.  Pass 2 is followed literally: float
// intersections s = ((f[q] + q^2) - f[v] - v^2) / (2q - 2v), pop while s <= z[k], and the fill
// that reads the image being overwritten (imgproc.h:122-128); writes to column q are coalesced.
//
// The per-row (v, f[v], z) stack is a three-level structure: the two top entries live in
// registers (the push/pop/push pattern of seedless columns never leaves them), the next C entries
// in an LDS ring ([slot][row], conflict free), and only older entries spill to HBM scratch
// ([slot][row], coalesced).  Inside the column loop nothing depends on a vector-memory load, so
// stores (spills, results) are never waited for; refills from HBM are rare and self-contained.
//
// R = rows per wave (64, 32 or 16).  The chain per row is sequential, so a small volume has too
// few rows to occupy 1024 SIMDs with full waves; with R < 64 lanes l and l + R run the same row
// (same values, same addresses, identical control flow), which multiplies the number of waves
// and leaves a longer LDS ring per row.  C = ring entries per row, SG = staging entries per row for
// the fill; the launcher picks (R, C, SG) so that every wave of the grid is resident at once.
template <int R, int C, int SG, bool PF>
__global__ void __launch_bounds__(256) k_pass2_l2(const ColDesc* __restrict__ desc, float* __restrict__ vol, int W,
                                                  int H, int HW64, long nwaves, int* __restrict__ sv,
                                                  float* __restrict__ sf, float* __restrict__ sz,
                                                  const int* __restrict__ only_flagged, int il) {
    constexpr int NR = 4 * R;    // distinct rows per block
    __shared__ int r_v[C][NR];
    __shared__ float r_f[C][NR];
    __shared__ float r_z[C][NR];
    constexpr int G = PF ? 64 / R : 1;  // lane groups of a row that fill different parts of it
    __shared__ int g_v[SG][NR * G];
    __shared__ float g_f[SG][NR * G];
    __shared__ float g_z[SG][NR * G];
    __shared__ uint4 dsc[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long wid = (long)blockIdx.x * 4 + wave;
    if (wid >= nwaves) return;  // wave-uniform
    constexpr int SUB = 64 / R;  // waves per 64-row chunk
    const long chunk = wid / SUB;
    // redo mode (after the segmented kernels below): only the chunks they flagged
    if (only_flagged && __builtin_amdgcn_readfirstlane(only_flagged[chunk]) == 0) return;
    const int sub = (int)(wid - chunk * SUB);
    const long k = chunk / HW64;
    const int c = (int)(chunk - k * HW64);
    const int bit = sub * R + (lane & (R - 1));  // row inside the chunk = bit of the seed word
    const int urow = wave * R + (lane & (R - 1));  // row inside the block (LDS column)
    const int y = c * 64 + bit;
    const long gid = wid * R + (lane & (R - 1));  // scratch row
    const uint4* dp = reinterpret_cast<const uint4*>(desc + ((size_t)k * HW64 + c) * W);
    // element x of the row: y-fastest row[x * H], or (il: behind the segmented sweep, whose fill writes the interleaved
    // layout ivol_index that the propagation then reads with coalesced loads) row[((x / 4) * H) * 4 + x % 4]
    float* row = il ? vol + (size_t)k * ivol_slice_floats(W, H) + (size_t)(y < H ? y : 0) * 4 : vol + (size_t)k * W * H + (y < H ? y : 0);
    const size_t H_ = (size_t)H, NT = (size_t)nwaves * R;
    auto xoff = [&](int x) -> size_t { return il ? ((size_t)(x >> 2) * H_) * 4 + (size_t)(x & 3) : (size_t)x * H_; };
    const float inf = f_inf();
    // ---- envelope construction (imgproc.h:101-121)
    int tv = 0, uv = 0;
    float tf = 0.f, tz = -inf, uf = 0.f, uz = 0.f;
    bool has_u = false;
    int cnt = 0;     // entries below the register pair
    int base = 0;    // entries [base, cnt) are in the LDS ring, [0, base) only in HBM
    int gvalid = 0;  // entries [0, gvalid) have a valid HBM copy
    // second register entry moves down into the ring as entry `cnt`; the oldest ring entry spills
    auto push_down = [&](int v, float f, float z) {
        if (cnt - base == C) {
            if (base >= gvalid) {
                const size_t slot = (size_t)base * NT + gid;
                const int r = base & (C - 1);
                sv[slot] = r_v[r][urow]; sf[slot] = r_f[r][urow]; sz[slot] = r_z[r][urow];
                gvalid = base + 1;
            }
            ++base;
        }
        const int r = cnt & (C - 1);
        r_v[r][urow] = v; r_f[r][urow] = f; r_z[r][urow] = z;
        ++cnt;
    };
    // One column of the envelope construction for this lane's row (imgproc.h:108-120).
    auto process_column = [&](int q, float fq) {
        const float q2 = (float)((unsigned)q * (unsigned)q);
        while (true) {
            const float tvf = (float)tv;  // tvf * tvf rounds like the reference's float(long(v * v)): same integer
            const float s = (fq + q2 - tf - tvf * tvf) / (float)(2 * q - 2 * tv);
            // (!has_u && cnt == 0): the top is entry 0 whose z is -inf; only guards non-finite input
            if (s > tz || (!has_u && cnt == 0)) {
                if (has_u) push_down(uv, uf, uz);
                uv = tv; uf = tf; uz = tz; has_u = true;
                tv = q; tf = fq; tz = s;
                break;
            }
            if (has_u) {
                tv = uv; tf = uf; tz = uz; has_u = false;
            } else {
                if (cnt == base) {  // ring empty: bring one spilled entry back (rare)
                    --base;
                    const size_t slot = (size_t)base * NT + gid;
                    const int r = base & (C - 1);
                    r_v[r][urow] = sv[slot]; r_f[r][urow] = sf[slot]; r_z[r][urow] = sz[slot];
                }
                --cnt;
                const int r = cnt & (C - 1);
                tv = r_v[r][urow]; tf = r_f[r][urow]; tz = r_z[r][urow];
                if (gvalid > cnt) gvalid = cnt;
            }
        }
    };
    uint4 dreg = dp[min(lane, W - 1)];
    for (int q0 = 0; q0 < W; q0 += 64) {
        // lane j holds the descriptor of column q0 + j: one ballot tells which columns are seedless
        const bool sl = desc_seedless(dreg);
        const unsigned long long smask = __ballot(sl);
        dsc[wave][lane] = dreg;                    // the only wait on vector memory per 64 columns
        dreg = dp[min(q0 + 64 + lane, W - 1)];     // next 64 descriptors, in flight during this chunk
        const int jn = min(64, W - q0);
        // Only columns that hold a seed enter the envelope.  A seedless column q (f = FLT_MAX, which
        // absorbs every finite term: those are < 2^33 and ulp(FLT_MAX)/2 = 2^103) is pushed by the
        // reference with z = FLT_MAX / (2(q - v_top)) >= 2^110 over a finite top (or z = +0 over a
        // seedless entry 0) and is popped again by the very next column, seedless or not, because
        // that column's intersection with it is -v^2/(2(q'-q)) <= 0 or about -FLT_MAX
        // (imgproc.h:111-118); the entries below it are not touched in between.  If it is still on
        // top at the end of the row it owns no pixel (z >= 2^110 > q at imgproc.h:124), or, over a
        // seedless entry 0, it yields FLT_MAX like entry 0 itself.  So the fill's output does not
        // depend on seedless columns other than column 0, and they are skipped.
        unsigned long long todo = ~smask;
        if (jn < 64) todo &= (1ull << jn) - 1ull;
        if (q0 == 0) {  // v[0] = 0, z[0] = -inf (imgproc.h:103-105)
            const uint4 d0 = dsc[wave][0];
            tf = column_value<true>(((unsigned long long)d0.y << 32) | d0.x, (int)d0.z, (int)d0.w, bit, y);
            todo &= ~1ull;
        }
        // The 64 / R lane groups of a row would compute the same pass-1 value; instead group g takes
        // the (g+1)-th pending column, and the values are handed round with lane permutes (issued one
        // column ahead), so the bit-scan runs once per 64 / R columns.
        constexpr int GC = 64 / R;
        const int grp_c = lane / R, lane_r = lane & (R - 1);
        while (todo) {
            unsigned long long tm = todo;
#pragma unroll
            for (int i = 0; i + 1 < GC; ++i)
                if (i < grp_c) tm &= tm - 1ull;
            const int jm = tm ? __ffsll((long long)tm) - 1 : 0;
            const uint4 dj = dsc[wave][jm];
            const float fmine = column_value<true>(((unsigned long long)dj.y << 32) | dj.x, (int)dj.z, (int)dj.w, bit, y);
            float fq = GC > 1 ? __shfl(fmine, lane_r) : fmine;
#pragma unroll 1
            for (int cc = 0; cc < GC && todo; ++cc) {
                const int j = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const float fq_next = (GC > 1 && cc + 1 < GC) ? __shfl(fmine, lane_r + R * (cc + 1)) : 0.f;
                process_column(q0 + j, fq);
                fq = fq_next;
            }
        }
    }
    // ---- the register pair joins the ring: entries [base, n) are in LDS, [0, base) in HBM
    if (has_u) push_down(uv, uf, uz);
    push_down(tv, tf, tz);
    const int n_entries = cnt;
    // ---- fill (imgproc.h:122-128).  The reference walks the pixels q = 0..W-1 with a pointer k
    // into the stack (advance while z[k+1] < q) and writes (q - v[k])^2 + img(v[k]), reading
    // img(v[k]) from the image it is overwriting: the original f[v_k] while v_k >= q, the already
    // written g[v_k] afterwards.  z is strictly increasing along the stack, so entry k takes over at
    // the first pixel above z_k and its addend is one constant: f[v_k] if z_k < v_k, else
    // g[v_k] = (v_k - v_o)^2 + addend_o with o the owner of pixel v_k.
    //
    // Entries are consumed in order from LDS (ring, or an SG-entry staging window refilled from HBM
    // for the spilled part) with a two-entry look-ahead.  g[v_k] is re-evaluated from the last three
    // owners (same float expression), which removes almost every read-back of the image.
    //
    // PF: the 64 / R lane groups that ran the same row during the construction now fill different
    // parts of it.  A part starts inside the pixels of an entry b with z_b < v_b (it took over at or
    // before its own position, so its addend is f[v_b] and needs nothing from earlier pixels) and
    // ends where the next group's part starts.  Every later entry k has v_k > v_b >= the pixel b
    // took over at, so the g[v_k] it may need is a pixel owned by b or by a later entry of the
    // part: it comes from the owner history or, on a miss, from a pixel this same lane has already
    // written.  Groups never read each other's pixels.
    const int grp = PF ? lane / R : 0;
    const int srow = urow + NR * grp;  // staging column of this (row, lane group)
    int st0 = -SG;  // staging window holds entries [st0, st0 + SG)
    auto fetch = [&](int i, int& v, float& f, float& z) {
        if (i >= n_entries) { v = -1; f = 0.f; z = inf; return; }
        if (i >= base) {
            const int r = i & (C - 1);
            v = r_v[r][urow]; f = r_f[r][urow]; z = r_z[r][urow];
            return;
        }
        if (i >= st0 + SG) {  // refill the window with [i, i + SG) from HBM (rare, self-contained)
            st0 = i;
            int lv[SG];
            float lf[SG], lz[SG];
#pragma unroll
            for (int e = 0; e < SG; ++e) {
                const size_t slot = (size_t)min(i + e, base - 1) * NT + gid;
                lv[e] = sv[slot]; lf[e] = sf[slot]; lz[e] = sz[slot];
            }
#pragma unroll
            for (int e = 0; e < SG; ++e) { g_v[e][srow] = lv[e]; g_f[e][srow] = lf[e]; g_z[e][srow] = lz[e]; }
        }
        v = g_v[i - st0][srow]; f = g_f[i - st0][srow]; z = g_z[i - st0][srow];
    };
    // random access to entry i (rare paths only; spilled entries are loaded and consumed in place)
    auto entry_at = [&](int i, int& v, float& f, float& z) {
        if (i >= base) {
            const int r = i & (C - 1);
            v = r_v[r][urow]; f = r_f[r][urow]; z = r_z[r][urow];
        } else {
            const size_t slot = (size_t)i * NT + gid;
            const int a0 = sv[slot]; const float a1 = sf[slot], a2 = sz[slot];
            asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(a0));
            asm volatile("v_mov_b32 %0, %1" : "=v"(f) : "v"(a1));
            asm volatile("v_mov_b32 %0, %1" : "=v"(z) : "v"(a2));
        }
    };
    // owner of pixel x: the last entry with z < x (z_0 = -inf, z strictly increasing)
    auto owner_of = [&](float xf) {
        int lo = 0, hi = n_entries - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            int mv; float mf, mz;
            entry_at(mid, mv, mf, mz);
            if (mz < xf) lo = mid; else hi = mid - 1;
        }
        return lo;
    };
    // Start of lane group g's part: g = 0 starts like the reference (pointer 0 at pixel 0); g >= G
    // is the end of the row.  Else the part starts at pixel x = g * W / G if the owner of x takes
    // over at or before its own position, otherwise where the next such entry takes over.
    // Returns the entry, the first pixel, and the pixel at which the entry took over.
    auto takeover = [&](float z) { return z < 0.f ? 0 : (!(z < (float)W) ? W : (int)floorf(z) + 1); };
    auto part_start = [&](int g, int& k, int& q, int& since) {
        if (g >= G) { k = n_entries; q = W; since = W; return; }
        if (g == 0) { k = 0; q = 0; since = 0; return; }
        const int x = min(g * ((W + G - 1) / G), W - 1);
        k = owner_of((float)x);
        int ev; float ef, ez;
        entry_at(k, ev, ef, ez);
        if (ez < (float)ev) { q = x; since = takeover(ez); return; }
        for (++k; k < n_entries; ++k) {
            entry_at(k, ev, ef, ez);
            if (ez < (float)ev) { q = since = takeover(ez); return; }
        }
        q = since = W;
    };
    int kk = 0, q_begin = 0, q_end = W, k_next = n_entries, since0 = 0, since1 = 0;
    if (PF) {
        part_start(grp, kk, q_begin, since0);
        part_start(grp + 1, k_next, q_end, since1);
        q_begin = min(q_begin, q_end);
    }
    int cv, av, bv;
    float cf, cz, af, az, bf, bz;
    fetch(kk, cv, cf, cz);
    fetch(kk + 1, av, af, az);
    fetch(kk + 2, bv, bf, bz);
    // Owner history: the current owner (cv, base_val) has owned pixels since ca; the two owners
    // before it are (pv, pbase) since pa and (p2v, p2base) since p2a.  When entry k takes over at a
    // pixel beyond its own position (!(z_k < v_k)) the reference reads the already written g[v_k]:
    // it is re-evaluated from the owner of pixel v_k in the history (same expression as the pixel
    // loop), or read back from the image if that owner is older than the history.
    float base_val = cf;
    const int v_first = cv;
    const float f_first = cf;
    int ca = since0, pv = 0, pa = 0x7fffffff, p2v = 0, p2a = 0x7fffffff;
    float pbase = 0.f, p2base = 0.f;
    for (int it = 0; PF ? __any(q_begin + it < q_end) : it < W; ++it) {
        const int q = q_begin + it;
        const bool mine = !PF || q < q_end;
        const float qf = (float)q;
        while (mine && az < qf) {
            ++kk;
            const int nv_ = av;
            float nbase = af;
            if (!(az < (float)nv_)) {
                int ov = cv; float ob = base_val; bool found = nv_ >= ca;
                if (!found && nv_ >= pa) { ov = pv; ob = pbase; found = true; }
                if (!found && nv_ >= p2a) { ov = p2v; ob = p2base; found = true; }
                // a pixel before this part's first pixel belongs to the entry the part started in
                if (PF && !found && nv_ < q_begin) { ov = v_first; ob = f_first; found = true; }
                if (found) {
                    const float dv = (float)(nv_ - ov);  // dv * dv rounds like float(long(dv * dv)): same integer
                    nbase = ob + dv * dv;
                } else {
                    float t = 0.f;
                    if (y < H) t = row[xoff(nv_)];  // rare read-back, consumed inside the branch
                    asm volatile("v_mov_b32 %0, %1" : "=v"(nbase) : "v"(t));
                }
            }
            if (ca < q) {  // the outgoing owner really owned pixels: keep it in the history
                p2v = pv; p2base = pbase; p2a = pa;
                pv = cv; pbase = base_val; pa = ca;
            }
            cv = nv_; base_val = nbase; ca = q;
            av = bv; af = bf; az = bz;
            fetch(kk + 2, bv, bf, bz);
        }
        const float dq = (float)(q - cv);  // dq * dq rounds like the reference's float(long(dq * dq))
        if (mine && y < H && (PF || lane < R)) row[xoff(q)] = base_val + dq * dq;
    }
}

// ------------------------------------------------------------------------------------------ K2, segmented
// The same literal pass (imgproc.h:91-130), with the sequential chain of a row cut into S segments that run
// in different waves, and the fill split from the construction.
//
// Why a row can be cut.  Once a column c has been pushed it stays on the stack as long as no later column
// pops it, and while it stays, the algorithm never looks at the entries below it: every test of a later column
// q is against the top entry and its z, and the only test that involves what lies below c is q against c
// itself, s(q, c) > z_c with z_c = s(c, entry below c).  So the columns after c can be run on a stack whose
// bottom is c with z = -inf ("local run"): every test not against the bottom is the real test on the same
// operands, and a test against the bottom always pushes.  The local run equals the real one exactly when the
// real tests against c all push as well, i.e. when min over those tests of s(q, c) > z_c -- one float per
// segment (minF), compared afterwards with the real z_c, which is the z of the previous segment's top entry
// (c is the last column of the previous segment, hence its top).  If the comparison fails for a row, its chunk
// is flagged and redone by the one-wave-per-chunk kernel above; nothing is assumed about floating point.
//
// Which column.  The speculation holds when c is a vertex of the final envelope.  The owner of a pixel x
// (the column minimising f_u + (x - u)^2) is one, so phase A of k_env finds, per row, the owner of the
// junction pixel x_j by scanning the seeded columns outwards from x_j until the distance alone exceeds the
// best value; x_j is the same for all rows of a slice (the j/S quantile of its seeded columns).
//
//   k_env     block = one 64-row chunk, wave w = segment w: seeded mask of the slice, junction owners
//             (phase A), local literal run over the columns (c_w, c_{w+1}] (phase B); stack = top entry in
//             registers + a C-entry LDS ring + HBM scratch; writes each local stack and minF
//   k_addend  lane = row: checks the junctions, walks the concatenated stack once and writes the list of
//             entries that own pixels as (first pixel, column, addend); the addend is f[v] when the entry
//             takes over at or before its own column, else the already written g[v] (the in-place quirk,
//             imgproc.h:126-127), re-evaluated from the owner of pixel v found a few entries back
//   k_fill    lane = (row, quarter of the pixels): pure fill from the owner list, entries staged through
//             LDS in rounds of RE
// store_unit_note: every 16-byte store of this file carries its whole offset in the lane (vector) offset, never in the
// scalar offset operand.  A store of more than 8 bytes reads its data registers late, and a vector instruction that
// overwrites them right behind it needs a wait state; the compiler inserts it only when the store has NO scalar
// offset register.  With the group offset in an SGPR and a tight loop (k_l1_forward: store, then the shift of the
// group registers) lanes 12-15 of every 16 stored the NEXT column's value in a group's first column -- at config 5
// only (12 GB in flight: the store's data fetch is late), in 0.15 % of the pixels; found by the sampled-slice test.
static constexpr int kSegMax = 8;
static constexpr int kFillParts = 4;

// Row-major scratch: every lane streams through its own row's records (the positions differ from row to row, so a
// [slot][row] layout would scatter the lanes of one access over as many pages as rows).
struct K2Buf {
    EnvEntry* ent;                        // envelope entries [row][slot], eslots per row
    OwnEntry* own;                        // owner list [row][index], lslots per row
    int* tcnt; float* tminf; int* tslot;  // per (segment, row): entries, min s against the bottom, first slot
    int* lcount;                          // owner entries per row
    int* partidx;                         // [part - 1][row]: list index that owns the first pixel of fill part 1..3
    int* flags;                           // per 64-row chunk: 1 = redo with k_pass2_l2
    long long* dbg;                       // optional per-wave clock stamps and counters (FDCM_K2_DEBUG), else null
    const int* order;                     // launch position -> chunk (longest chunks of the previous build first), or null
    int* cost;                            // per chunk: 100 MHz ticks from the block's start to the end of its addend pass
    long NR;                              // rows of the scratch arrays (chunks * 64)
    int eslots, lslots;
    int addend_waves;                     // waves of a block that share the addend pass (a power of two; 0 = all of them)
    const unsigned long long* colmask;    // [slice][(W + 63) / 64]: the slice's seeded columns (k_coldesc_tile), or null
};

// Phase 1 of k_sweep.  smask: seeded columns of the slice, 64 per word (W <= 16384); cj: junction column of
// segment w per row; ring: stack entries below the top, (float(2 v), f[v], z, float(v)^2), C per thread.
template <int C, int NT, bool DBG>
__device__ __forceinline__ void env_phase(const ColDesc* __restrict__ desc, int W, int H, int HW64, int S, const K2Buf& B,
                                          int expm, unsigned long long* smask, int (*cj)[64], float4 (*ring)[NT]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long chunk = B.order ? B.order[blockIdx.x] : (long)blockIdx.x;
    const long k = chunk / HW64;
    const int c = (int)(chunk - k * HW64);
    const int y = c * 64 + lane;
    const size_t NR = (size_t)B.NR;
    const size_t r = (size_t)chunk * 64 + lane;
    const uint4* dp = reinterpret_cast<const uint4*>(desc + ((size_t)k * HW64 + c) * W);
    const int nwords = (W + 63) >> 6;
    const float inf = f_inf();
    long long* dbg = DBG ? B.dbg + ((size_t)chunk * kSegMax + wave) * 16 : nullptr;
    long long n_cols = 0, n_iter = 0, n_evict = 0, n_refill = 0, n_eval = 0;
    long long sc0 = 0;
    if (DBG && lane == 0) { dbg[0] = wall_clock64(); sc0 = clock64(); }
    if (tid == 0) B.flags[chunk] = 0;
    if (B.colmask) {  // (H <= 4096: the descriptor kernel left the mask)
        for (int b = tid; b < nwords; b += NT) smask[b] = B.colmask[(size_t)k * nwords + b];
    } else {
        for (int b = wave; b < nwords; b += S) {
            const int x = b * 64 + lane;
            const uint4 d = dp[min(x, W - 1)];
            const unsigned long long mk = __ballot(x < W && !desc_seedless(d));
            if (lane == 0) smask[b] = mk;
        }
    }
    __syncthreads();
    if (DBG && lane == 0) dbg[1] = wall_clock64();
    // ---- phase A: the owner of the junction pixel of this wave's segment start
    if (wave > 0) {
        int n = 0;
        for (int b = 0; b < nwords; ++b) n += __popcll(uni64(smask[b]));
        int x = (int)(((long)W * wave) / S);
        if (n >= 2 * S) {
            int t = (int)(((long)n * wave) / S);  // rank of the junction column among the seeded ones
            int b = 0;
            unsigned long long mk = uni64(smask[0]);
            while (__popcll(mk) <= t) { t -= __popcll(mk); ++b; mk = uni64(smask[b]); }
            for (; t > 0; --t) mk &= mk - 1ull;
            x = b * 64 + __ffsll((long long)mk) - 1;
        }
        float best = inf;
        int bestu = 0;
        // A column can only matter if even its smallest value over the chunk's 64 rows (squared distance from the
        // chunk to the column's nearest seed, from the descriptor) plus its distance to x is within the bound: one
        // vector test per word (lane = column) leaves the few columns worth a per-row evaluation.
        const int y0 = c * 64, y63 = y0 + 63;
        auto eval_word = [&](int wd, const uint4& dv, float bound) {
            const unsigned long long seeded = uni64(smask[wd]);
            if (!seeded) return;
            const float up = (float)(y0 - (int)dv.z), dn = (float)((int)dv.w - y63);  // 2^30 when there is none
            const float lb = (dv.x | dv.y) ? 0.f : fminf(up, dn);
            const float dx = (float)(wd * 64 + lane - x);
            unsigned long long mk = __ballot(lb * lb + dx * dx <= bound) & seeded;
            while (mk) {
                // two columns per round: their evaluations do not depend on each other, so the instructions of one fill
                // the issue gaps of the other (a lone wave issues a dependent instruction every ~7 cycles).  The last
                // column of an odd count is evaluated twice, which changes nothing (ties go to the smaller column).
                const int j1 = __ffsll((long long)mk) - 1;
                mk &= mk - 1ull;
                const int j2 = mk ? __ffsll((long long)mk) - 1 : j1;
                mk &= mk - 1ull;
                unsigned long long wc1, wc2;
                int pc1, nc1, pc2, nc2;
                desc_lane4(dv, j1, wc1, pc1, nc1);
                desc_lane4(dv, j2, wc2, pc2, nc2);
                const int u1 = wd * 64 + j1, u2 = wd * 64 + j2, du1 = u1 - x, du2 = u2 - x;
                const float val1 = column_value_sq_seeded(wc1, pc1, nc1, lane, y) + (float)(du1 * du1);
                const float val2 = column_value_sq_seeded(wc2, pc2, nc2, lane, y) + (float)(du2 * du2);
                const bool b1 = val1 < best || (val1 == best && u1 < bestu);
                best = b1 ? val1 : best;
                bestu = b1 ? u1 : bestu;
                const bool b2 = val2 < best || (val2 == best && u2 < bestu);
                best = b2 ? val2 : best;
                bestu = b2 ? u2 : bestu;
                if (DBG) n_eval += 2;
            }
        };
        const int wi = x >> 6;
        // descriptors of the words one step ahead are in flight while a step is evaluated
        auto load_word = [&](int wd) { return dp[min(min(max(wd, 0), nwords - 1) * 64 + lane, W - 1)]; };
        uint4 dR = load_word(wi), dL = dR, dRn = load_word(wi + 1), dLn = load_word(wi - 1);
        for (int s = 0;; ++s) {
            const int wr = wi + s, wl = wi - s;
            // values are >= +0, so their bit patterns order like the values
            const float mb = __int_as_float(__builtin_amdgcn_readfirstlane(wave_max(__float_as_int(best))));
            bool doR = wr < nwords, doL = s > 0 && wl >= 0;
            if (doR && s > 0) { const int d = wr * 64 - x; doR = (float)(d * d) <= mb; }
            if (doL) { const int d = x - (wl * 64 + 63); doL = (float)(d * d) <= mb; }
            if (!doR && !doL) break;
            const uint4 cR = dR, cL = dL;
            dR = dRn; dL = dLn;
            dRn = load_word(wr + 2); dLn = load_word(wl - 2);
            if (doR) eval_word(wr, cR, mb);
            if (doL) eval_word(wl, cL, mb);
        }
        cj[wave][lane] = bestu;
    }
    if (DBG && lane == 0) dbg[2] = wall_clock64();
    __syncthreads();
    if (DBG && lane == 0) dbg[3] = wall_clock64();
    // ---- phase B: the literal run over the columns (cs, ce] on a stack whose bottom is column cs
    const int cs = wave == 0 ? 0 : cj[wave][lane];
    int ce = wave == S - 1 ? W - 1 : cj[wave + 1][lane];
    ce = max(ce, cs);
    const uint4 db = dp[cs];
    // top entry t and the entry below it u (a register copy of ring entry cnt - 1, so that a single pop needs no
    // LDS round trip): twice the column (as float), f, z, column^2
    float tvf = (float)cs;  // (only for the squares below; the loop carries 2 v: what the test's denominator needs)
    float tf = column_value<true>(((unsigned long long)db.y << 32) | db.x, (int)db.z, (int)db.w, lane, y);
    float tz = -inf;
    float tv2 = tvf * tvf, tvx2 = tvf + tvf;
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0;   // entries below the top (indices 0..cnt-1); [base, cnt) in the LDS ring, [0, base) in HBM
    int base = 0;
    float minF = inf;
    const int slot0 = cs + wave;  // segments of a row use disjoint slot ranges
    EnvEntry* ent = B.ent + r * (size_t)B.eslots + slot0;
    auto evict = [&]() {
        const float4 e = ring[base & (C - 1)][tid];
        ent[base] = EnvEntry{(int)e.x >> 1, e.y, e.z};
        ++base;
        if (DBG) ++n_evict;
    };
    const int qlo = __builtin_amdgcn_readfirstlane(wave_min(cs)) + 1;
    const int qhi = __builtin_amdgcn_readfirstlane(wave_max(ce));
    if (qlo <= qhi) {
        const int wlo = qlo >> 6, whi = qhi >> 6;
        for (int wd = wlo; wd <= whi; ++wd) {
            {
                // Lane j holds the descriptor of column 64 wd + j; a column's fields are read with v_readlane.  Loaded and
                // waited for here, once per 64 columns, and not prefetched across words: a load still pending over the
                // column loop makes the compiler wait for (nearly) all memory operations at every column, i.e. for the
                // spill stores, and a descriptor staged in LDS couples every column to the ring traffic through lgkmcnt.
                const uint4 dcur = dp[min(wd * 64 + lane, W - 1)];
                asm volatile("; descriptors %0 %1 %2 %3 are complete here, before the column loop" ::"v"(dcur.x), "v"(dcur.y), "v"(dcur.z), "v"(dcur.w));
                unsigned long long mk = uni64(smask[wd]);  // seedless columns are skipped (see k_pass2_l2)
                if (wd == wlo) mk &= ~0ull << (qlo & 63);
                if (wd == whi) mk &= ~0ull >> (63 - (qhi & 63));
                while (mk) {
                    const int j = __ffsll((long long)mk) - 1;
                    mk &= mk - 1ull;
                    const int q = wd * 64 + j;
                    uint4 dq;
                    dq.x = (unsigned)__builtin_amdgcn_readlane((int)dcur.x, j);
                    dq.y = (unsigned)__builtin_amdgcn_readlane((int)dcur.y, j);
                    dq.z = (unsigned)__builtin_amdgcn_readlane((int)dcur.z, j);
                    dq.w = (unsigned)__builtin_amdgcn_readlane((int)dcur.w, j);
                    float fq = column_value_sq_seeded(((unsigned long long)dq.y << 32) | dq.x, (int)dq.z, (int)dq.w, lane, y);
                    if (DBG && (expm & 4)) fq = (float)((lane * 7 + q * 13) & 1023);  // timing experiment: no pass-1 value
                    const bool act = q > cs && q <= ce;
                    const float qf = (float)q;
                    const float q2 = qf * qf;  // rounds like the reference's float(long(q * q))
                    // (a lane outside its range carries NaN through the test: every comparison with it is false, it never pops)
                    const float hq = act ? fq + q2 : f_nan();
                    const float twoq = qf + qf;
                    float s;
                    bool pop;
                    unsigned long long any_pop;
                    if (DBG) ++n_cols;
                    // Test at the bottom: one taken branch per extra pass, none on the way out.  A lane that does not pop
                    // recomputes the same s in the passes other lanes still need.  (A fully predicated body -- selects
                    // instead of the branches below -- measured 17 % slower: a lone wave retires a dependent instruction
                    // every ~8 cycles, so the instruction count of the chain is what matters, not its branches.)
                    do {
                        if (DBG) ++n_iter;
                        // s = ((f[q] + q^2) - f[v] - v^2) / (2q - 2v), left to right in float (imgproc.h:111)
                        const float N = (hq - tf) - tv2;
                        s = envelope_quotient(N, twoq - tvx2);  // = N / (2q - 2v) bit for bit, in 4 instructions for 11 (fdcm_quotient.h)
                        if (DBG && (expm & 2)) s = N * __builtin_amdgcn_rcpf(twoq - tvx2);  // timing experiment: no division
                        // pop while s <= z[k].  One ordered comparison decides: an inactive lane's s is NaN; the bottom entry's
                        // z is -inf and s is finite (about -FLT_MAX / D over a seedless bottom column), so the bottom is never
                        // popped -- no test of the lane's range and of cnt > 0 on the chain (three instructions less)
                        pop = s <= tz;
                        if (DBG && (expm & 1)) pop = false;  // timing experiment: no pops
                        any_pop = __builtin_amdgcn_ballot_w64(pop);  // taken here, from the comparison's own mask
                        if (pop) {
                            tvx2 = u.x; tf = u.y; tz = u.z; tv2 = u.w;
                            --cnt;
                            if (cnt > 0) {
                                if (__builtin_expect(cnt == base, 0)) {  // ring empty: up to four spilled entries come back together
                                    EnvEntry en[4];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) en[e] = ent[max(base - 1 - e, 0)];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {  // all four are written (the ring is empty; entries below 0 land in free slots): no load stays pending
                                        const float vf = (float)en[e].v;
                                        ring[(base - 1 - e) & (C - 1)][tid] = make_float4(vf + vf, en[e].f, en[e].z, vf * vf);
                                    }
                                    base = max(base - 4, 0);
                                    if (DBG) ++n_refill;
                                }
                                u = ring[(cnt - 1) & (C - 1)][tid];  // needed at the next pop at the earliest
                            }
                        }
                    } while (any_pop != 0ull);
                    if (act) {
                        if (cnt == 0) minF = s < minF ? s : minF;  // a test against the bottom entry
                        if (__builtin_expect(cnt - base == C, 0)) evict();
                        u = make_float4(tvx2, tf, tz, tv2);
                        ring[cnt & (C - 1)][tid] = u;
                        ++cnt;
                        tf = fq; tz = s; tv2 = q2; tvx2 = twoq;
                    }
                }
            }
        }
    }
    if (DBG && lane == 0) dbg[4] = wall_clock64();
    // the top joins the entries; everything still in the ring goes to HBM
    if (cnt - base == C) evict();
    ring[cnt & (C - 1)][tid] = make_float4(tvx2, tf, tz, tv2);
    ++cnt;
#pragma unroll
    for (int e = 0; e < C; ++e) {
        const int i = base + e;
        if (i < cnt) {
            const float4 en = ring[i & (C - 1)][tid];
            ent[i] = EnvEntry{(int)en.x >> 1, en.y, en.z};
        }
    }
    const size_t to = (size_t)wave * NR + r;
    B.tcnt[to] = cnt; B.tminf[to] = minF; B.tslot[to] = slot0;
    if (DBG && lane == 0) {
        dbg[5] = wall_clock64();
        dbg[6] = n_cols; dbg[7] = n_iter; dbg[8] = n_evict; dbg[9] = n_refill;
        dbg[10] = qhi - qlo + 1; dbg[15] = n_eval;
        if (wave == 1) dbg[11] = clock64() - sc0;  // shader clocks of this wave's life (k_addend uses wave 0's slot 11)
    }
}

// Junction check + one walk over the concatenated stack of a row (lane = row, one wave per block).  The stack of a
// row is the stacks of its segments back to back, without the bottom entries of segments 1.. (they repeat the
// previous segment's top).  Every lane streams through its own row; the entries of a batch are staged in LDS so
// that one compact loop body serves every entry; the owner entries collect in an LDS ring and go to HBM in
// bursts, so that the loads of the walk do not queue behind stores.
template <int KL, int NB, bool DBG>
__device__ __forceinline__ void addend_phase(int W, int S, int part_w, const K2Buf& B, int force_mod_x, int rpw, unsigned (*l_pk)[64],
                                             float (*l_b)[64], int (*e_v)[64], float (*e_f)[64], float (*e_z)[64]) {
    // l_pk / l_b: the last KL owner entries of each row; e_*: the batch (NB entries) being walked
    const int force_mod = force_mod_x & 0xffff, xp = DBG ? force_mod_x >> 16 : 0;  // xp: timing experiments of the debug build
    // rpw rows per wave (a power of two): the block's waves share the chunk's 64 rows, wave w takes rows w * rpw .. and
    // every lane is one of them (lanes rpw .. 63 repeat lanes 0 .. rpw - 1, so wave-wide operations see consistent
    // values; only the first rpw lanes write to memory).  Fewer rows per wave: fewer of the rare paths (an owner
    // lookup, a read behind the LDS window) are taken by the wave for the sake of one row.
    const int wlane = threadIdx.x & 63;
    const int lane = (int)(threadIdx.x >> 6) * rpw + (wlane & (rpw - 1));  // the row inside the chunk, and the LDS column
    const bool writer = wlane < rpw;
    const long chunk = B.order ? B.order[blockIdx.x] : (long)blockIdx.x;
    const size_t NR = (size_t)B.NR;
    const size_t r = (size_t)chunk * 64 + lane;
    OwnEntry* own = B.own + r * (size_t)B.lslots;
    const EnvEntry* ent = B.ent + r * (size_t)B.eslots;
    long long* dbg = DBG ? B.dbg + ((size_t)chunk * kSegMax) * 16 : nullptr;
    long long n_batches = 0;
    int n_quirk = 0, n_hbm = 0, n_probe = 0, n_far = 0;
    if (DBG && threadIdx.x == 0) dbg[11] = wall_clock64();
    // Segment table of the row: stream index i lies in segment w for i in [o_w, o_{w+1}), at slot i + K_w.  The
    // junction checks: every test against a segment's bottom column must also push on the real stack, where that
    // column is the previous segment's top (min s over those tests > z of that top).
    int o[kSegMax + 1], K[kSegMax];
    bool ok = true;
    {
        int n_prev = B.tcnt[r], s_prev = B.tslot[r];
        o[0] = 0; o[1] = n_prev; K[0] = s_prev;
#pragma unroll
        for (int w = 1; w < kSegMax; ++w) {
            o[w + 1] = o[w]; K[w] = 0;
            if (w < S) {
                const int n = B.tcnt[(size_t)w * NR + r], s0 = B.tslot[(size_t)w * NR + r];
                const EnvEntry top = ent[s_prev + n_prev - 1];
                ok = ok && B.tminf[(size_t)w * NR + r] > top.z && top.v == s0 - w;
                o[w + 1] = o[w] + n - 1; K[w] = s0 - o[w] + 1;
                n_prev = n; s_prev = s0;
            }
        }
    }
    const int total = o[kSegMax];
    auto slot_of = [&](int i) {
        int k = K[0];
#pragma unroll
        for (int w = 1; w < kSegMax; ++w) k = i >= o[w] ? K[w] : k;
        return i + k;
    };
    int pend_v = 0, pend_st = -1, last_st = -1, lc = 0, flushed = 0;
    int pi[3] = {0, 0, 0}, part = 0, bnd = part_w;  // pi[j]: the last list entry whose first pixel is <= (j + 1) * part_w
    int optr = 0;  // owner list index whose first pixel is <= the column looked up last (columns only grow)
    float pend_f = 0.f;
    auto list_pk = [&](int i) -> unsigned {
        if (DBG) ++n_probe;
        if (i >= lc - KL) return l_pk[i & (KL - 1)][lane];
        if (DBG) ++n_hbm;
        const unsigned a0 = own[i].pk;  // older than the LDS window (rare): from the list in HBM, consumed in place
        unsigned v;
        asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(a0));
        return v;
    };
    auto list_b = [&](int i) -> float {
        if (i >= lc - KL) return l_b[i & (KL - 1)][lane];
        const float a0 = own[i].b;
        float v;
        asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(a0));
        return v;
    };
    // The pending entry owns the pixels [pend_st, st_next) if that range is not empty (owner of q = the last
    // entry with z < q, imgproc.h:124).  Addend: f[v] if it takes over at or before its own column, else the
    // value the reference reads back at v: g[v] = addend_o + (v - v_o)^2 with o the owner of pixel v.  The
    // columns of successive entries grow, so o is found with a pointer that only moves forward.
    auto finalize = [&](int st_next) {
        const bool owns = pend_st < st_next && pend_st > last_st;  // (last_st >= -1, so a pending entry exists)
        const bool quirk = owns && pend_st > pend_v;
        float b = pend_f;
        if (__builtin_amdgcn_ballot_w64(quirk && !(xp & 8)) != 0ull) {
            if (quirk) {
                if (DBG) ++n_quirk;
                // last list entry whose first pixel is <= pend_v, at or after optr: two steps forward (rows along a scene
                // line need one per entry), else gallop back from the tail (the owner is a few entries back) and bisect.
                // The entry at optr, the three after it and their addends are read together when they are all in the LDS
                // window (one LDS round trip; one read at a time made up to five, each waited for): the steps and the
                // owner's record are then selects.
                unsigned pk;
                float ob;
                bool far;  // a third step forward would still not be far enough
                if (__builtin_expect(optr >= lc - KL, 1)) {
                    const int i1 = min(optr + 1, lc - 1), i2 = min(optr + 2, lc - 1), i3 = min(optr + 3, lc - 1);
                    const unsigned pk0 = l_pk[optr & (KL - 1)][lane], pk1 = l_pk[i1 & (KL - 1)][lane], pk2 = l_pk[i2 & (KL - 1)][lane],
                                   pk3 = l_pk[i3 & (KL - 1)][lane];
                    const float b0 = l_b[optr & (KL - 1)][lane], b1 = l_b[i1 & (KL - 1)][lane], b2 = l_b[i2 & (KL - 1)][lane];
                    if (DBG) n_probe += 4;
                    const bool a1 = optr + 1 < lc && (int)(pk1 >> 16) <= pend_v;
                    const bool a2 = a1 && optr + 2 < lc && (int)(pk2 >> 16) <= pend_v;
                    far = a2 && optr + 3 < lc && (int)(pk3 >> 16) <= pend_v;
                    optr += (a1 ? 1 : 0) + (a2 ? 1 : 0);
                    pk = a2 ? pk2 : (a1 ? pk1 : pk0);
                    ob = a2 ? b2 : (a1 ? b1 : b0);
                } else {
                    // The pointer is more than the LDS window behind the list's end (rows whose envelope lags its columns by
                    // more than 64 pixels: the diagonals' longest blocks): the same four entries, from the list in HBM where
                    // they have left the window -- one trip to memory for all four, consumed in place, where one read at a
                    // time made three to five, each waiting for everything else in flight (the next batch of the walk).
                    const int i1 = min(optr + 1, lc - 1), i2 = min(optr + 2, lc - 1), i3 = min(optr + 3, lc - 1);
                    const OwnEntry h0 = own[optr], h1 = own[i1], h2 = own[i2], h3 = own[i3];  // (slots not flushed yet: read, not used)
                    unsigned hp0, hp1, hp2, hp3;
                    float hb0, hb1, hb2;
                    asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                                 : "=&v"(hp0), "=&v"(hp1), "=&v"(hp2), "=&v"(hp3) : "v"(h0.pk), "v"(h1.pk), "v"(h2.pk), "v"(h3.pk));
                    asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" : "=&v"(hb0), "=&v"(hb1), "=&v"(hb2) : "v"(h0.b), "v"(h1.b), "v"(h2.b));
                    if (DBG) { n_probe += 4; n_hbm += 4; }
                    const bool w1 = i1 >= lc - KL, w2 = i2 >= lc - KL, w3 = i3 >= lc - KL;  // (optr itself is behind the window here)
                    const unsigned pk0 = hp0, pk1 = w1 ? l_pk[i1 & (KL - 1)][lane] : hp1, pk2 = w2 ? l_pk[i2 & (KL - 1)][lane] : hp2,
                                   pk3 = w3 ? l_pk[i3 & (KL - 1)][lane] : hp3;
                    const float b0 = hb0, b1 = w1 ? l_b[i1 & (KL - 1)][lane] : hb1, b2 = w2 ? l_b[i2 & (KL - 1)][lane] : hb2;
                    const bool a1 = optr + 1 < lc && (int)(pk1 >> 16) <= pend_v;
                    const bool a2 = a1 && optr + 2 < lc && (int)(pk2 >> 16) <= pend_v;
                    far = a2 && optr + 3 < lc && (int)(pk3 >> 16) <= pend_v;
                    optr += (a1 ? 1 : 0) + (a2 ? 1 : 0);
                    pk = a2 ? pk2 : (a1 ? pk1 : pk0);
                    ob = a2 ? b2 : (a1 ? b1 : b0);
                }
                if (far) {
                    int hi = lc - 1, lo = hi, step = 1;
                    while (lo > optr && (int)(list_pk(lo) >> 16) > pend_v) { hi = lo - 1; lo = max(optr, lo - step); step <<= 1; }
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if ((int)(list_pk(mid) >> 16) <= pend_v) lo = mid; else hi = mid - 1;
                    }
                    optr = lo;
                    pk = list_pk(optr); ob = list_b(optr);
                }
                if (DBG && lc - optr > KL) ++n_far;
                const float dv = (float)(pend_v - (int)(pk & 0xffffu));  // dv * dv rounds like float(long(dv * dv))
                b = ob + dv * dv;
            }
        }
        if (owns) {
            l_pk[lc & (KL - 1)][lane] = ((unsigned)pend_st << 16) | (unsigned)pend_v;
            l_b[lc & (KL - 1)][lane] = b;
            if (__builtin_expect(pend_st > bnd, 0)) {  // first pixels grow: a part boundary is crossed three times per row
                while (part < 3 && pend_st > bnd) { pi[part++] = max(lc - 1, 0); bnd += part_w; }
                if (part == 3) bnd = 0x7fffffff;
            }
            last_st = pend_st;
            ++lc;
        }
    };
    auto flush = [&]() {  // ring -> HBM: all lanes together, in list order
        const int fmax = __builtin_amdgcn_readfirstlane(wave_max(lc - flushed));
        for (int e = 0; e < fmax; e += 4) {  // four entries per LDS round trip (slots past lc hold older entries: read, not stored)
            unsigned pk4[4];
            float b4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { pk4[j] = l_pk[(flushed + e + j) & (KL - 1)][lane]; b4[j] = l_b[(flushed + e + j) & (KL - 1)][lane]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = flushed + e + j;
                if (i < lc && writer && !(xp & 1)) own[i] = OwnEntry{pk4[j], b4[j]};
            }
        }
        flushed = lc;
    };
    const int tmax = __builtin_amdgcn_readfirstlane(wave_max(total));
    const float Wf = (float)W;
    EnvEntry nxt[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) nxt[u] = ent[slot_of(min(u, total - 1))];
    for (int i0 = 0; i0 < tmax; i0 += NB) {
        if (DBG) ++n_batches;
#pragma unroll
        for (int u = 0; u < NB; ++u) { e_v[u][lane] = nxt[u].v; e_f[u][lane] = nxt[u].f; e_z[u][lane] = nxt[u].z; }
        if (!(xp & 4))
#pragma unroll
        for (int u = 0; u < NB; ++u) nxt[u] = ent[slot_of(min(i0 + NB + u, total - 1))];  // in flight during this batch
        const int un = min(NB, tmax - i0);
        // four entries at a time: their LDS reads and first pixels do not depend on one another, only the list does
#pragma unroll 1
        for (int u0 = 0; u0 < un; u0 += 4) {
            int v4[4], st4[4];
            float f4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = e_z[u0 + j][lane];
                v4[j] = e_v[u0 + j][lane]; f4[j] = e_f[u0 + j][lane];
                // first pixel above z: the entry takes over there (while (z[k+1] < q) ++k); -1 past the row's last entry
                // (and past the batch: un need not be a multiple of 4, NB is): nothing is finalised (pend_st >= 0 > -1)
                // (0 below 0, W from W on; the clamps keep it to four instructions: max(z, -1) floors to -1 for every z
                // below 0, min(z, W - 1/2) to W - 1 for every z from W - 1 on)
                const int st = (int)floorf(__builtin_fminf(__builtin_fmaxf(z, -1.f), Wf - 0.5f)) + 1;
                st4[j] = (i0 + u0 + j < total && u0 + j < un) ? st : -1;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!(xp & 2)) finalize(st4[j]);
                if (st4[j] >= 0) { pend_v = v4[j]; pend_f = f4[j]; pend_st = st4[j]; }
            }
        }
        if (__builtin_amdgcn_ballot_w64(lc - flushed >= KL / 2) != 0ull) flush();  // a batch adds at most NB < KL / 2
    }
    finalize(W);
    flush();
    if (writer) B.lcount[r] = lc;
    for (; part < 3; ++part) pi[part] = max(lc - 1, 0);
    if (writer) { B.partidx[r] = pi[0]; B.partidx[NR + r] = pi[1]; B.partidx[2 * NR + r] = pi[2]; }
    const bool forced = force_mod > 0 && chunk % force_mod == 0;  // test hook: exercise the redo path
    if ((__builtin_amdgcn_ballot_w64(!ok) != 0ull || forced) && wlane == 0) B.flags[chunk] = 1;
    if (DBG) {
        const int q = wave_max(n_quirk), hb = wave_max(n_hbm), pr = wave_max(n_probe), fr = wave_max(n_far), tt = wave_max(total);
        if (threadIdx.x == 0) {
            dbg[12] = wall_clock64(); dbg[13] = n_batches; dbg[14] = __builtin_amdgcn_readfirstlane(wave_max(lc));
            long long* dx = B.dbg + ((size_t)chunk * kSegMax + (kSegMax - 1)) * 16;  // slots of a wave that does not exist (S <= 4 here)
            dx[0] = q; dx[1] = hb; dx[2] = pr; dx[3] = fr; dx[4] = tt;
        }
    }
}

// Pure fill (imgproc.h:122-128) from the owner list; wave p of a block fills the pixels
// [p * part_w, (p + 1) * part_w) of the block's 64 rows, writes to column q are coalesced.
template <int RE>
__device__ __forceinline__ void fill_phase(float* __restrict__ vol, int W, int H, int HW64, int part_w, const K2Buf& B, int p,
                                           unsigned (*f_pk)[256], float (*f_b)[256]) {
    const int tid = threadIdx.x & 255, lane = tid & 63;
    const long chunk = B.order ? B.order[blockIdx.x] : (long)blockIdx.x;
    const long k = chunk / HW64;
    const int c = (int)(chunk - k * HW64);
    const int y = c * 64 + lane;
    const size_t NR = (size_t)B.NR;
    const size_t r = (size_t)chunk * 64 + lane;
    int qcur = p * part_w;
    const int qend = min(qcur + part_w, W);
    if (qcur >= qend) return;  // (last phase of the kernel: nothing waits for this wave any more)
    int idx = p == 0 ? 0 : B.partidx[(size_t)(p - 1) * NR + r];
    const int lc = B.lcount[r];
    const OwnEntry* own = B.own + r * (size_t)B.lslots;
    // The fill writes the interleaved layout (ivol_index: 16 bytes = 4 neighbouring columns of one row) that the
    // propagation reads: a lane collects the values of a group of 4 columns and stores them as one unit, 64 rows = 1 KB
    // contiguous per wave (the y-fastest layout took a 4-byte store per pixel, and the propagation then loaded it in
    // runs of 64 bytes: config 3 0.43 -> 0.37 ms there).  Parts start on a group (part_w is a multiple of 4).
    const size_t sl = ivol_slice_floats(W, H);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(vol + (size_t)k * sl, 0, (unsigned)(sl * 4), 0x00020000);
    const unsigned vrow = y < H ? (unsigned)y * 16u : 0x80000000u;  // rows past the image: dropped stores
    const int grpB = H * 16;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;  // the group being collected (shift register: the newest value in g3)
    while (qcur < qend) {
        // entries [idx, idx + RE) of every row go to LDS; the round ends where the first row would need entry idx + RE
        unsigned pk[RE];
        float bb[RE];
#pragma unroll
        for (int e = 0; e < RE; ++e) {
            const OwnEntry oe = own[min(idx + e, lc - 1)];
            pk[e] = oe.pk; bb[e] = oe.b;
        }
#pragma unroll
        for (int e = 0; e < RE; ++e) {
            if (idx + e >= lc) pk[e] = 0x7fff0000u;  // past the list: never taken over
            f_pk[e][tid] = pk[e]; f_b[e][tid] = bb[e];
        }
        const int lim = (int)(pk[RE - 1] >> 16);
        // (max: the lists k_addend writes always allow progress; never spin on anything else)
        const int qstop = max(qcur + 1, min(qend, __builtin_amdgcn_readfirstlane(wave_min(lim))));
        unsigned cpk = pk[0], npk = pk[1];
        float cb = bb[0], nb = bb[1];
        int a = 0;
        for (int q = qcur; q < qstop; ++q) {
            const bool adv = q >= (int)(npk >> 16);
            if (adv) { cpk = npk; cb = nb; ++a; }
            npk = f_pk[a + 1][tid]; nb = f_b[a + 1][tid];  // a + 1 <= RE - 1 because q < lim
            const float dq = (float)(q - (int)(cpk & 0xffffu));  // dq * dq rounds like float(long(dq * dq))
            g0 = g1; g1 = g2; g2 = g3; g3 = cb + dq * dq;
            if ((q & 3) == 3) {  // wave-uniform
                u32x4 out;
                out.x = __float_as_uint(g0); out.y = __float_as_uint(g1); out.z = __float_as_uint(g2); out.w = __float_as_uint(g3);
                __builtin_amdgcn_raw_buffer_store_b128(out, rs, vrow + (unsigned)((q >> 2) * grpB), 0, 0);  // (no scalar offset: see store_unit_note)
            }
        }
        idx += a;
        qcur = qstop;
    }
    if (qend == W && (W & 3)) {  // the row's last group is partial: its columns past W are padding and hold 0
        for (int q = W; q & 3; ++q) { g0 = g1; g1 = g2; g2 = g3; g3 = 0.f; }
        u32x4 out;
        out.x = __float_as_uint(g0); out.y = __float_as_uint(g1); out.z = __float_as_uint(g2); out.w = __float_as_uint(g3);
        __builtin_amdgcn_raw_buffer_store_b128(out, rs, vrow + (unsigned)((W >> 2) * grpB), 0, 0);
    }
}

// The three phases in one launch, one workgroup per 64-row chunk: the chunks whose construction is long (rows far
// from every seed: long gaps without an envelope vertex) are not the ones whose owner lists are long (rows along a
// scene line), so running addend and fill of a chunk right behind its own construction lets the tails of the phases
// overlap between chunks instead of adding up between launches (config 2: 0.54 -> measured in DESIGN.md).  The LDS
// of the construction's ring is reused by the later phases.
template <int C, int NT, bool DBG>
__global__ void __launch_bounds__(NT) k_sweep(const ColDesc* __restrict__ desc, float* __restrict__ vol, int W, int H, int HW64,
                                              int S, int part_w, K2Buf B, int force_mod, int expm) {
    // KL: owner entries per row kept in LDS for the look-ups (32: 0.47 instead of 0.43 ms at config 2, the look-ups
    // behind the window go to HBM); NB: stack entries per row and batch of the walk (16: 140 instead of 98 VGPRs).
    // 42 KB of LDS: three blocks per CU.  (With the mask and the junction columns moved into the pool's tail a fourth
    // fits; config 3 then takes 1.11 instead of 1.03 ms: the waves are latency bound and slow one another down.)
    constexpr int KL = 64, NB = 8, RE = 16;
    // one LDS pool for the three phases: the construction's ring, the addend pass's lists, the fill's staging
    constexpr size_t kPoolBytes = std::max({(size_t)C * NT * sizeof(float4), (size_t)(2 * KL + 3 * NB) * 64 * 4, (size_t)2 * RE * 256 * 4});
    __shared__ unsigned long long smask[256];
    __shared__ int cj[NT / 64 + 1][64];
    __shared__ float4 pool[kPoolBytes / sizeof(float4)];
    const long long t_start = wall_clock64();
    env_phase<C, NT, DBG>(desc, W, H, HW64, S, B, expm, smask, cj, reinterpret_cast<float4(*)[NT]>(pool));
    __syncthreads();  // every segment's stack, count and minF are in memory
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // the addend pass on the largest power of two of the block's waves, each with its share of the 64 rows
    int aw = 1 << (31 - __builtin_clz((int)blockDim.x >> 6));
    if (B.addend_waves > 0 && B.addend_waves < aw) aw = B.addend_waves;
    if (wave < aw) {
        unsigned* w32 = reinterpret_cast<unsigned*>(pool);
        addend_phase<KL, NB, DBG>(W, S, part_w, B, force_mod, 64 / aw, reinterpret_cast<unsigned(*)[64]>(w32),
                                  reinterpret_cast<float(*)[64]>(w32 + KL * 64), reinterpret_cast<int(*)[64]>(w32 + 2 * KL * 64),
                                  reinterpret_cast<float(*)[64]>(w32 + (2 * KL + NB) * 64),
                                  reinterpret_cast<float(*)[64]>(w32 + (2 * KL + 2 * NB) * 64));
    }
    __syncthreads();  // the chunk's owner lists are in memory
    if (threadIdx.x == 0) B.cost[B.order ? B.order[blockIdx.x] : (long)blockIdx.x] = (int)(wall_clock64() - t_start);
    {
        unsigned* w32 = reinterpret_cast<unsigned*>(pool);
        const int nw = (int)blockDim.x >> 6;
        for (int p = wave; p < kFillParts; p += nw)
            fill_phase<RE>(vol, W, H, HW64, part_w, B, p, reinterpret_cast<unsigned(*)[256]>(w32),
                           reinterpret_cast<float(*)[256]>(w32 + RE * 256));
    }
}

// Launch order of the next build's chunks: by decreasing cost of this one (scenes of a stream change little from
// frame to frame).  Blocks are dispatched in index order; when there are more blocks than the GPU holds, the
// longest ones must not start last.  One workgroup: a counting sort over 256 cost classes.
__global__ void __launch_bounds__(1024) k_order(const int* __restrict__ cost, int n, int* __restrict__ order) {
    __shared__ int hist[256], cursor[256], smax;
    const int tid = threadIdx.x;
    if (tid < 256) hist[tid] = 0;
    if (tid == 0) smax = 1;
    __syncthreads();
    int mx = 1;
    for (int i = tid; i < n; i += 1024) mx = max(mx, cost[i]);
    atomicMax(&smax, mx);
    __syncthreads();
    const float scale = 255.f / (float)smax;
    for (int i = tid; i < n; i += 1024) atomicAdd(&hist[255 - min(255, max(0, (int)((float)cost[i] * scale)))], 1);
    __syncthreads();
    if (tid == 0) { int run = 0; for (int b = 0; b < 256; ++b) { cursor[b] = run; run += hist[b]; } }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) order[atomicAdd(&cursor[255 - min(255, max(0, (int)((float)cost[i] * scale)))], 1)] = i;
}

// The same phases as three launches: no wave waits at a barrier while one wave of its block runs the addend pass, so
// the GPU's other work (frames of a pipeline) gets those slots; alone, the tails of the three launches add up.
template <int C, int NT, bool DBG>
__global__ void __launch_bounds__(NT) k_env(const ColDesc* __restrict__ desc, int W, int H, int HW64, int S, K2Buf B, int expm) {
    __shared__ unsigned long long smask[256];
    __shared__ int cj[NT / 64 + 1][64];
    __shared__ float4 pool[C * NT];
    env_phase<C, NT, DBG>(desc, W, H, HW64, S, B, expm, smask, cj, reinterpret_cast<float4(*)[NT]>(pool));
}
template <bool DBG>
__global__ void __launch_bounds__(64) k_addend(int W, int S, int part_w, K2Buf B, int force_mod) {
    constexpr int KL = 64, NB = 16;
    __shared__ unsigned w32[(2 * KL + 3 * NB) * 64];
    addend_phase<KL, NB, DBG>(W, S, part_w, B, force_mod, 64, reinterpret_cast<unsigned(*)[64]>(w32),
                              reinterpret_cast<float(*)[64]>(w32 + KL * 64), reinterpret_cast<int(*)[64]>(w32 + 2 * KL * 64),
                              reinterpret_cast<float(*)[64]>(w32 + (2 * KL + NB) * 64),
                              reinterpret_cast<float(*)[64]>(w32 + (2 * KL + 2 * NB) * 64));
}
__global__ void __launch_bounds__(256) k_fill(float* __restrict__ vol, int W, int H, int HW64, int part_w, K2Buf B) {
    constexpr int RE = 16;
    __shared__ unsigned w32[2 * RE * 256];
    fill_phase<RE>(vol, W, H, HW64, part_w, B, __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6),
                   reinterpret_cast<unsigned(*)[256]>(w32), reinterpret_cast<float(*)[256]>(w32 + RE * 256));
}

// distanceTransform<float, L1> (imgproc.h:176-181): the sweeps along y are the descriptor
// distance (exact integers), the forward sweep along x runs on it directly (imgproc.h:138-140)
// and writes V; the backward sweep (imgproc.h:142-145) reads that and writes the result in place.
// Both work on the interleaved layout (ivol_index: 16 bytes = 4 neighbouring columns of one row) that the propagation
// reads: a lane moves one unit per group of 4 columns, 64 rows = 1 KB contiguous per wave operation (the y-fastest
// form moved 4 bytes per lane and column).  The forward sweep is bound by its vector instructions (11 520 waves x 4096
// columns at config 5), hence the short path for columns without a seed inside the chunk: 8.04 -> 7.27 ms for both sweeps.
__global__ void __launch_bounds__(256) k_l1_forward(const ColDesc* __restrict__ desc, float* __restrict__ vol, int W,
                                                    int H, int HW64, long nwaves) {
    const int tid = threadIdx.x, lane = tid & 63;
    const long wid = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wid >= nwaves) return;
    const long k = wid / HW64;
    const int c = (int)(wid - k * HW64);
    const int y = c * 64 + lane;
    const bool active = y < H;
    const ColDesc* dp = desc + ((size_t)k * HW64 + c) * W;
    const size_t sl = ivol_slice_floats(W, H);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(vol + (size_t)k * sl, 0, (unsigned)(sl * 4), 0x00020000);
    const unsigned vrow = active ? (unsigned)y * 16u : 0x80000000u;  // rows past the image: dropped stores
    const int grpB = H * 16;
    float run = 0.f;
    ColDesc dcur = dp[min(lane, W - 1)];
    // One column: the distance along y from the descriptor (most columns have no seed inside the chunk's 64 rows -- a
    // wave-uniform test on a ballot of the 64 descriptors -- and then the two neighbour rows decide: 2 lane reads and 6
    // vector instructions instead of 4 and ~22), then the forward recurrence.
    auto column = [&](unsigned long long nz, int j, bool first) -> float {
        float cq;
        if ((nz >> j) & 1ull) {
            unsigned long long wc;
            int pc, nc;
            desc_lane(dcur, j, wc, pc, nc);
            cq = column_value<false>(wc, pc, nc, lane, y);
        } else {
            const int pc = __builtin_amdgcn_readlane(dcur.prev, j), nc = __builtin_amdgcn_readlane(dcur.next, j);
            const int d = min(y - pc, nc - y);
            cq = d >= (1 << 29) ? FLT_MAX : (float)d;  // no seed in the whole column
        }
        run = first ? cq : std_min(cq, run + 1);
        return run;
    };
    for (int q0 = 0; q0 < W; q0 += 64) {
        const ColDesc dnext = dp[min(q0 + 64 + lane, W - 1)];
        const int jn = min(64, W - q0);
        const unsigned long long nz = __builtin_amdgcn_ballot_w64(dcur.word != 0ull);
        int j = 0;
        for (; j + 4 <= jn; j += 4) {  // whole groups of 4 columns: one 16-byte unit per lane
            u32x4 out;
            out.x = __float_as_uint(column(nz, j, q0 + j == 0));
            out.y = __float_as_uint(column(nz, j + 1, false));
            out.z = __float_as_uint(column(nz, j + 2, false));
            out.w = __float_as_uint(column(nz, j + 3, false));
            __builtin_amdgcn_raw_buffer_store_b128(out, rs, vrow + (unsigned)(((q0 + j) >> 2) * grpB), 0, 0);
        }
        if (j < jn) {  // the row's last group is partial: its columns past W are padding and hold 0
            float g[4] = {0.f, 0.f, 0.f, 0.f};
            for (int e = 0; j + e < jn; ++e) g[e] = column(nz, j + e, q0 + j + e == 0);
            u32x4 out;
            out.x = __float_as_uint(g[0]); out.y = __float_as_uint(g[1]); out.z = __float_as_uint(g[2]); out.w = __float_as_uint(g[3]);
            __builtin_amdgcn_raw_buffer_store_b128(out, rs, vrow + (unsigned)(((q0 + j) >> 2) * grpB), 0, 0);
        }
        dcur = dnext;
    }
}

__global__ void __launch_bounds__(256) k_l1_backward(float* __restrict__ vol, int W, int H, long nrows) {
    // one wave = 64 consecutive rows of one slice (H is padded per slice so that waves never straddle slices)
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wps = (H + 63) >> 6;  // waves per slice
    const long k = wid / wps;
    const int y = (int)(wid - k * wps) * 64 + lane;
    if (k * (long)H >= nrows) return;  // wave-uniform
    // The slice through a buffer descriptor: the group of 4 columns is the scalar offset, the row the lane offset; rows
    // past the image and groups before 0 get a lane offset of 2^31 (loads 0, drops stores), so no memory operation
    // sits behind a branch and two batches of loads stay in flight behind the stores.
    const size_t sl = ivol_slice_floats(W, H);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(vol + (size_t)k * sl, 0, (unsigned)(sl * 4), 0x00020000);
    const unsigned vrow = y < H ? (unsigned)y * 16u : 0x80000000u;
    const int grpB = H * 16;
    const int W4 = (W + 3) >> 2;
    // the last group by itself: column W - 1 is the sweep's start (img.col(W-1) stays), padding columns stay 0
    float run;
    {
        u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, vrow, (W4 - 1) * grpB, 0);
        float v[4] = {__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
        const int last = (W - 1) & 3;  // uniform
        run = v[0];
#pragma unroll
        for (int cc = 3; cc >= 0; --cc) {
            if (cc == last) run = v[cc];
            else if (cc < last) { run = std_min(v[cc], run + 1); v[cc] = run; }
        }
        u.x = __float_as_uint(v[0]); u.y = __float_as_uint(v[1]); u.z = __float_as_uint(v[2]); u.w = __float_as_uint(v[3]);
        __builtin_amdgcn_raw_buffer_store_b128(u, rs, vrow + (unsigned)((W4 - 1) * grpB), 0, 0);
    }
    constexpr int U = 4;  // groups per batch
    u32x4 va[U], vb[U];
    // groups run W4-2 .. 0; batch t covers G = W4-2-t*U-j.  Loads never alias earlier stores of the sweep.
    auto fetch = [&](int t, u32x4 (&v)[U]) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int G = W4 - 2 - t * U - j;  // uniform
            v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, G >= 0 ? vrow : 0x80000000u, max(G, 0) * grpB, 0);
        }
    };
    auto consume = [&](int t, const u32x4 (&v)[U], float& run) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int G = W4 - 2 - t * U - j;
            u32x4 o;  // past group 0 the values are unused and the store is dropped
            run = std_min(__uint_as_float(v[j].w), run + 1); o.w = __float_as_uint(run);
            run = std_min(__uint_as_float(v[j].z), run + 1); o.z = __float_as_uint(run);
            run = std_min(__uint_as_float(v[j].y), run + 1); o.y = __float_as_uint(run);
            run = std_min(__uint_as_float(v[j].x), run + 1); o.x = __float_as_uint(run);
            __builtin_amdgcn_raw_buffer_store_b128(o, rs, G >= 0 ? vrow + (unsigned)(G * grpB) : 0x80000000u, 0, 0);
        }
    };
    fetch(0, va);
    fetch(1, vb);
    for (int t = 0; t * U < W4 - 1; t += 2) {
        consume(t, va, run);
        fetch(t + 2, va);
        consume(t + 1, vb, run);  // past group 0: dropped stores
        fetch(t + 3, vb);
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
static constexpr int kShP = 12;                 // groups (loads of 1 KB) in flight per wave
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
                                                  const int* __restrict__ tab, int only_mode, int shw, int kstride) {
    extern __shared__ float lds_tiles[];
    // Slices are taken in a strided order (kstride is coprime to the slice count and close to half of it): in index order
    // all steep slices of one angular range run before the shallow ones, and the two kinds stress different things (LDS
    // tiles against straight 1 KB streams), so mixing them over the launch overlaps them.
    const int k = (int)(((long)blockIdx.y * kstride) % (long)gridDim.y);
    const IntegralDesc d = desc[k];
    if (only_mode && d.mode != only_mode) return;  // timing experiment (FDCM_INT_ONLY): one kind of slice only
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

void run_build(fdcm_featuremap* fm, const BuildPlan& plan, int stop_after) {
    const auto t0 = std::chrono::steady_clock::now();
    FDCM_HIP(hipSetDevice(fm->device));
    if (!fm->stream) FDCM_HIP(hipStreamCreateWithFlags(&fm->stream, hipStreamNonBlocking));
    ensure_timing(fm);
    finish_build(fm);  // the previous build's staging and events are reused below
    hipStream_t st = fm->stream;
    fm->W = plan.W; fm->H = plan.H; fm->m = plan.m; fm->tx = plan.tx; fm->ty = plan.ty;
    fm->keys = plan.keys;
    fm->last_build = fdcm_build_timing{};
    if (plan.m == 0 || plan.W == 0) return;
    const int W = (int)plan.W, H = (int)plan.H, m = (int)plan.m;
    if (plan.W > 16384 || plan.H > 16384) throw std::string("feature size above 16384 is not supported");  // 32-bit byte offsets inside a slice
    const int HW64 = (H + 63) / 64;
    const size_t npix = (size_t)W * H, nvox = npix * m;
    const long nrows = (long)m * H, ncols = (long)m * W;
    fm->vol.reserve(std::max(nvox, (size_t)m * ivol_slice_floats(W, H)) * sizeof(float));  // the integrated volume comes back here, interleaved
    fm->bitmap.reserve((size_t)ncols * HW64 * 8);
    // every buffer of the build is reserved here, before the first kernel is queued: an allocation between two stages
    // (a handle's first build) stalls the host for 0.5 - 1 ms while the GPU idles inside the stage events' span
    if (stop_after >= 2) fm->ivol.reserve((size_t)m * ivol_slice_floats(W, H) * sizeof(float));
    if (stop_after >= 3) fm->offtab.reserve((size_t)m * sh_tab_stride(W) * sizeof(int));
    const long nchunks = (long)m * HW64;  // (slice, 64-row chunk) pairs
    // Tuning overrides, read once (measurements and tests only).
    static const int env_rows = getenv("FDCM_K2_ROWS") ? atoi(getenv("FDCM_K2_ROWS")) : 0;
    static const int env_lean = getenv("FDCM_K2_LEAN") ? atoi(getenv("FDCM_K2_LEAN")) : -1;
    static const bool env_serial_fill = getenv("FDCM_K2_SERIAL_FILL") != nullptr;
    static const bool env_legacy = getenv("FDCM_K2_LEGACY") != nullptr;            // one-wave-per-chunk kernel only
    static const int env_segments = getenv("FDCM_K2_SEGMENTS") ? atoi(getenv("FDCM_K2_SEGMENTS")) : 0;
    static const int env_force_redo = getenv("FDCM_K2_FORCE_REDO") ? atoi(getenv("FDCM_K2_FORCE_REDO")) : 0;
    static const bool env_debug = getenv("FDCM_K2_DEBUG") != nullptr;  // per-wave clock stamps, printed after the sweep
    static const bool env_unfused = getenv("FDCM_K2_UNFUSED") != nullptr;  // three launches (k_env, k_addend, k_fill) instead of one
    static const int env_experiment = getenv("FDCM_K2_EXPERIMENT") ? atoi(getenv("FDCM_K2_EXPERIMENT")) : 0;  // debug kernels only
    // The segmented sweep (k_sweep) is the default: config 2 (480 chunks) 0.47 ms against 0.65 ms for the
    // one-wave-per-chunk kernel, config 3 (1920 chunks) 1.75 against 1.79 ms.  Its scratch is 20 B per pixel against
    // 12 B, so volumes above 2^32 pixels (48 GB of scratch) keep the fused kernel.
    const bool segmented = fm->distance != FDCM_L1 && !env_legacy && (nvox <= (1ull << 32) || env_segments > 0);
    // the L1 sweeps, the segmented L2 sweep's fill and its redo path write the transforms interleaved (ivol_index);
    // only the one-wave-per-chunk L2 kernel alone (volumes above 2^32 pixels, FDCM_K2_LEGACY) keeps the y-fastest form
    fm->vol1_interleaved = segmented || fm->distance == FDCM_L1;
    int R = 64;                            // rows per wave of the one-wave-per-chunk L2 sweep: keep >= 2048 waves in flight
    if (!segmented) {
        while (R > 16 && nchunks * (64 / R) < 2048) R >>= 1;
        if (env_rows == 16 || env_rows == 32 || env_rows == 64) R = env_rows;
    }
    const long nwaves = fm->distance == FDCM_L1 ? nchunks : nchunks * (64 / R);
    // segments per row of the segmented sweep: small images need the split most (few rows), but a segment
    // should still hold a few dozen columns
    int S = W >= 128 ? 4 : (W >= 64 ? 2 : 1);
    // A build that has the GPU to itself and fits it in one go (every block resident at once: the kernel then lasts as
    // long as its longest block) takes 5 segments, i.e. 320-thread blocks with shorter chains: config 2 0.42 -> 0.38 ms.
    // With other frames' kernels beside it (pipeline slots) or more blocks than slots (config 3) the four-wave block is
    // the better one (4 frames in flight: 67 against 59 M matches/s; config 3: 1.03 against 1.35 ms).
    if (S == 4 && W >= 512 && !fm->shares_gpu && nchunks <= 3L * device_cus(fm->device)) S = 5;
    if (env_segments >= 1 && env_segments <= kSegMax) S = env_segments;
    const int part_w = (((W + kFillParts - 1) / kFillParts) + 3) & ~3;  // fill parts start on a group of 4 columns
    fm->coldesc.reserve((size_t)ncols * HW64 * sizeof(ColDesc));
    if (fm->distance == FDCM_L1) fm->stack.reserve((size_t)m * HW64 * ((W + 63) / 64) * 64 * sizeof(float2));  // the L1 pass's minima / carries
    fm->colmask.reserve((size_t)m * ((W + 63) / 64) * 8);
    K2Buf kb{};
    bool proxy_order = false;
    int* order_dst = nullptr;
    if (fm->distance != FDCM_L1) {
        // scratch of the L2 sweeps: envelope entries (W + kSegMax + 2 slots per row, 12 B), owner list (W + 2
        // entries per row, 8 B), per-segment and per-row records, chunk flags
        // (the one-wave-per-chunk kernel alone -- volumes above 2^32 pixels -- only needs its (v, f, z) spill space,
        // 12 B per pixel, which is the entry area: no owner list and no per-segment records then)
        const size_t NRr = (size_t)nchunks * 64, slots = segmented ? (size_t)W + kSegMax + 2 : (size_t)W, lslots = (size_t)W + 2;
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
        const size_t segw = segmented ? 1 : 0;
        const size_t o_ent = take(slots * NRr * sizeof(EnvEntry)), o_own = take(segw * lslots * NRr * sizeof(OwnEntry));
        const size_t o_tc = take(segw * kSegMax * NRr * 4), o_tm = take(segw * kSegMax * NRr * 4), o_ts = take(segw * kSegMax * NRr * 4);
        const size_t o_lc = take(segw * NRr * 4), o_pi = take(segw * 3 * NRr * 4), o_fl = take((size_t)nchunks * 4);
        const size_t o_dbg = take(env_debug ? (size_t)nchunks * kSegMax * 16 * 8 : 0);
        const size_t o_ord = take((size_t)nchunks * 4), o_cost = take((size_t)nchunks * 4);
        const void* stack_before = fm->stack.p;
        fm->stack.reserve(off);
        char* sb = (char*)fm->stack.p;
        // Launch order: blocks are dispatched in index order, and when there are more of them than the GPU holds at once
        // (three per CU) the long ones must not start last.  Nothing cheap predicts a chunk's time well enough (the seeded
        // and the far columns of a chunk correlate 0.85 with it and buy 4 %), the previous build of the same shape does:
        // scenes of a stream change little from frame to frame (config 3: 1.50 -> 1.08 ms; the first build of a handle,
        // and every build after a change of size, runs in index order).  FDCM_K2_LPT=0 / 1 forces it off / on.
        static const int env_lpt = getenv("FDCM_K2_LPT") ? atoi(getenv("FDCM_K2_LPT")) : -1;
        const bool want_order = env_lpt >= 0 ? env_lpt != 0 : nchunks > 3L * device_cus(fm->device);
        const bool have_cost = want_order && segmented && fm->k2_cost_chunks == nchunks && fm->k2_cost_w == W && stack_before == fm->stack.p;
        // without history (a handle's first build, a change of size): the host's proxy per chunk (make_build_plan), which
        // arrives with the plan blob; k_order is queued behind that copy below
        proxy_order = want_order && segmented && !have_cost && plan.chunk_cost.size() == (size_t)nchunks;
        if (have_cost) hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, st, (const int*)(sb + o_cost), (int)nchunks, (int*)(sb + o_ord));
        kb.order = (have_cost || proxy_order) ? (const int*)(sb + o_ord) : nullptr;
        order_dst = (int*)(sb + o_ord);
        kb.cost = (int*)(sb + o_cost);
        fm->k2_cost_chunks = segmented ? nchunks : 0; fm->k2_cost_w = W;
        kb.ent = (EnvEntry*)(sb + o_ent); kb.own = (OwnEntry*)(sb + o_own);
        kb.tcnt = (int*)(sb + o_tc); kb.tminf = (float*)(sb + o_tm); kb.tslot = (int*)(sb + o_ts);
        kb.lcount = (int*)(sb + o_lc); kb.partidx = (int*)(sb + o_pi); kb.flags = (int*)(sb + o_fl);
        kb.dbg = env_debug ? (long long*)(sb + o_dbg) : nullptr;
        kb.NR = (long)NRr; kb.eslots = (int)slots; kb.lslots = (int)lslots;
        static const int env_aw = getenv("FDCM_K2_AW") ? atoi(getenv("FDCM_K2_AW")) : 0;  // measurement: waves sharing the addend pass
        kb.addend_waves = env_aw;
    }
    // ---- plan upload: one pinned blob, one async copy
    auto align16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    fm->off_raster = 0;
    fm->off_prop = align16(plan.raster.size() * sizeof(RasterLine));
    fm->off_integral = fm->off_prop + align16(plan.prop.size() * sizeof(PropStep));
    fm->off_keys = fm->off_integral + align16(plan.integral.size() * sizeof(IntegralDesc));
    fm->off_cost = fm->off_keys + align16(plan.keys.size() * sizeof(float));
    const size_t blob = fm->off_cost + (proxy_order ? align16(plan.chunk_cost.size() * sizeof(int32_t)) : 0);
    fm->stage.reserve(blob);
    fm->plan.reserve(blob);
    char* hs = (char*)fm->stage.p;
    if (!plan.raster.empty()) std::memcpy(hs + fm->off_raster, plan.raster.data(), plan.raster.size() * sizeof(RasterLine));
    std::memcpy(hs + fm->off_prop, plan.prop.data(), plan.prop.size() * sizeof(PropStep));
    std::memcpy(hs + fm->off_integral, plan.integral.data(), plan.integral.size() * sizeof(IntegralDesc));
    std::memcpy(hs + fm->off_keys, plan.keys.data(), plan.keys.size() * sizeof(float));
    if (proxy_order) std::memcpy(hs + fm->off_cost, plan.chunk_cost.data(), plan.chunk_cost.size() * sizeof(int32_t));
    FDCM_HIP(hipMemcpyAsync(fm->plan.p, hs, blob, hipMemcpyHostToDevice, st));
    if (proxy_order)
        hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, st, (const int*)((const char*)fm->plan.p + fm->off_cost), (int)nchunks, order_dst);
    fm->n_raster = (int64_t)plan.raster.size();
    fm->n_prop = (int64_t)plan.prop.size();
    const char* dp = (const char*)fm->plan.p;
    const RasterLine* d_raster = (const RasterLine*)(dp + fm->off_raster);
    const PropStep* d_prop = (const PropStep*)(dp + fm->off_prop);
    const IntegralDesc* d_int = (const IntegralDesc*)(dp + fm->off_integral);
    float* vol = fm->vol.as<float>();
    hipEvent_t* ev = fm->timing.ev;

    fm->build_host_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    FDCM_HIP(hipEventRecord(ev[0], st));
    const long bitmap_words = ncols * HW64;
    if (!(fm->bitmap_clean && fm->bitmap_words == bitmap_words))
        FDCM_HIP(hipMemsetAsync(fm->bitmap.p, 0, (size_t)bitmap_words * 8, st));
    fm->bitmap_clean = false;
    if (fm->n_raster > 0)
        hipLaunchKernelGGL(k_seeds, dim3((unsigned)fm->n_raster), dim3(256), 0, st, d_raster,
                           fm->bitmap.as<unsigned long long>(), W, H, HW64);
    FDCM_HIP(hipEventRecord(ev[1], st));
    ColDesc* d_desc = fm->coldesc.as<ColDesc>();
    if (HW64 <= 64) {
        unsigned long long* bm = fm->bitmap.as<unsigned long long>();
        const int XT = HW64 > 32 ? 32 : 64;
        const dim3 grid((unsigned)((W + XT - 1) / XT), (unsigned)m);
        const size_t lds = (size_t)HW64 * (XT + 1) * sizeof(uint4);
        unsigned* cm = (unsigned*)fm->colmask.p;
        kb.colmask = (const unsigned long long*)fm->colmask.p;
        if (HW64 <= 16) hipLaunchKernelGGL((k_coldesc_tile<16, 64>), grid, dim3(256), lds, st, bm, d_desc, W, HW64, cm);
        else if (HW64 <= 32) hipLaunchKernelGGL((k_coldesc_tile<32, 64>), grid, dim3(256), lds, st, bm, d_desc, W, HW64, cm);
        else hipLaunchKernelGGL((k_coldesc_tile<64, 32>), grid, dim3(256), lds, st, bm, d_desc, W, HW64, cm);
    } else {
        hipLaunchKernelGGL(k_coldesc, dim3((unsigned)((ncols + 3) / 4)), dim3(256), 0, st,
                           fm->bitmap.as<unsigned long long>(), d_desc, W, HW64, ncols);
    }
    if (HW64 <= 64) { fm->bitmap_clean = true; fm->bitmap_words = bitmap_words; }  // one group of words per column: cleared in place
    FDCM_HIP(hipEventRecord(ev[2], st));
    {
        const unsigned wblocks = (unsigned)((nwaves + 3) / 4);
        static const bool env_l1_two_sweeps = getenv("FDCM_L1_TWO_SWEEPS") != nullptr;  // measurement: the forward and the backward kernel
        if (fm->distance == FDCM_L1 && !env_l1_two_sweeps) {
            // one pass over the volume: minima per (row, word), their prefix / suffix over the row's words, then word by word
            const int nwords = (W + 63) / 64;
            const long wwaves = (long)m * HW64 * nwords;
            float2* mins = (float2*)fm->stack.p;  // reserved above
            hipLaunchKernelGGL(k_l1_word_mins, dim3((unsigned)((wwaves + 3) / 4)), dim3(256), 0, st, d_desc, mins, W, HW64, nwords, wwaves);
            hipLaunchKernelGGL(k_l1_carries, dim3((unsigned)(((long)m * HW64 * 64 + 255) / 256)), dim3(256), 0, st, mins, nwords, (long)m * HW64 * 64);
            hipLaunchKernelGGL(k_l1_word, dim3((unsigned)((wwaves + 3) / 4)), dim3(256), 0, st, d_desc, (const float2*)mins, vol, W, H, HW64, nwords, wwaves);
        } else if (fm->distance == FDCM_L1) {
            hipLaunchKernelGGL(k_l1_forward, dim3(wblocks), dim3(256), 0, st, d_desc, vol, W, H, HW64, nwaves);
            {
                const long bwaves = (long)m * ((H + 63) / 64);  // 64 rows of one slice per wave
                hipLaunchKernelGGL(k_l1_backward, dim3((unsigned)((bwaves + 3) / 4)), dim3(256), 0, st, vol, W, H, nrows);
            }
        } else {
            // the one-wave-per-chunk kernel uses the entry arrays as its (v, f, z) scratch: [W][nwaves * R] each
            int* sv = (int*)kb.ent;
            float* sf = (float*)(sv + (size_t)W * nwaves * R);
            float* sz = sf + (size_t)W * nwaves * R;
            const int* gate = nullptr;
#define FDCM_K2(RR, CC, SS, PP) hipLaunchKernelGGL((k_pass2_l2<RR, CC, SS, PP>), dim3(wblocks), dim3(256), 0, st, d_desc, vol, W, H, HW64, nwaves, sv, sf, sz, gate, segmented ? 1 : 0)
            static const bool env_junction = getenv("FDCM_K2_JUNCTION") != nullptr;  // measurement: the junction-verified sweep at every size
            if (segmented && kb.colmask && sweep_balanced_applies(W, H) && !env_junction) {
                // every value of the pass is an exact integer: ranges of equal column count, merged (fdcm_sweep.hip)
                SweepBuf sb{};
                sb.ent = kb.ent; sb.own = kb.own; sb.order = kb.order; sb.cost = kb.cost; sb.eslots = kb.eslots; sb.lslots = kb.lslots; sb.colmask = kb.colmask;
                launch_sweep_balanced(st, d_desc, vol, W, H, HW64, nchunks, sb);
            } else if (segmented) {
                // One launch: the phases' tails overlap between chunks (the three-launch form is kept for measurements:
                // FDCM_K2_UNFUSED).  More than 4 segments (FDCM_K2_SEGMENTS): 512-thread blocks with an 8-entry ring.
                const bool three = env_unfused;
#define FDCM_SWEEP(CC, NN, DD) hipLaunchKernelGGL((k_sweep<CC, NN, DD>), dim3((unsigned)nchunks), dim3(64 * S), 0, st, d_desc, vol, W, H, HW64, S, part_w, kb, (env_force_redo & 0xffff) | (DD ? (env_experiment >> 8) << 16 : 0), env_experiment & 0xff)
#define FDCM_ENV(CC, NN, DD) hipLaunchKernelGGL((k_env<CC, NN, DD>), dim3((unsigned)nchunks), dim3(64 * S), 0, st, d_desc, W, H, HW64, S, kb, env_experiment)
                if (!three) {
                    // Up to 4 segments: 256-thread blocks with an 8-entry ring (the pool is then the addend pass's 45 KB: three
                    // blocks per CU, which is also what 139 VGPRs allow).  A 16-entry ring (77 KB, two blocks per CU) is as
                    // fast alone at config 2 and slower wherever blocks queue for a CU: config 3 1.74 -> 1.50 ms, four frames
                    // in flight 61 -> 67 M matches/s.  (A 4-entry ring changes nothing more; capping the registers at 128
                    // for a fourth block spills in the column loop and loses.)
                    static const int env_ring = getenv("FDCM_K2_RING") ? atoi(getenv("FDCM_K2_RING")) : 0;  // measurement: ring entries per row in LDS
                    if (S <= 4 && env_ring == 16) FDCM_SWEEP(16, 256, false);
                    else if (S <= 4) { if (env_debug) FDCM_SWEEP(8, 256, true); else FDCM_SWEEP(8, 256, false); }
                    else { if (env_debug) FDCM_SWEEP(8, 512, true); else FDCM_SWEEP(8, 512, false); }
                } else {
                    if (S <= 4) { if (env_debug) FDCM_ENV(16, 256, true); else FDCM_ENV(16, 256, false); }
                    else { if (env_debug) FDCM_ENV(8, 512, true); else FDCM_ENV(8, 512, false); }
                    if (env_debug) hipLaunchKernelGGL(k_addend<true>, dim3((unsigned)nchunks), dim3(64), 0, st, W, S, part_w, kb, env_force_redo);
                    else hipLaunchKernelGGL(k_addend<false>, dim3((unsigned)nchunks), dim3(64), 0, st, W, S, part_w, kb, env_force_redo);
                    hipLaunchKernelGGL(k_fill, dim3((unsigned)nchunks), dim3(256), 0, st, vol, W, H, HW64, part_w, kb);
                }
#undef FDCM_SWEEP
#undef FDCM_ENV
                if (const char* dump = getenv("FDCM_K2_DUMP_COST")) {  // measurement: the chunks' times of this build, as int32
                    FDCM_HIP(hipStreamSynchronize(st));
                    std::vector<int> hc((size_t)nchunks);
                    FDCM_HIP(hipMemcpy(hc.data(), kb.cost, hc.size() * 4, hipMemcpyDeviceToHost));
                    if (FILE* f = fopen(dump, "wb")) { fwrite(hc.data(), 4, hc.size(), f); fclose(f); }
                }
                // chunks whose junction check failed are redone literally, one wave per chunk (all others exit at once)
                gate = kb.flags;
                FDCM_K2(64, 8, 4, false);
                if (env_debug) {  // diagnostic: per-wave phase times (100 MHz clock) and loop counters
                    FDCM_HIP(hipStreamSynchronize(st));
                    std::vector<long long> d((size_t)nchunks * kSegMax * 16);
                    FDCM_HIP(hipMemcpy(d.data(), kb.dbg, d.size() * 8, hipMemcpyDeviceToHost));
                    std::vector<int> fl((size_t)nchunks);
                    FDCM_HIP(hipMemcpy(fl.data(), kb.flags, fl.size() * 4, hipMemcpyDeviceToHost));
                    double sum[6] = {0, 0, 0, 0, 0, 0}, mx[6] = {0, 0, 0, 0, 0, 0}, cnt[5] = {0, 0, 0, 0, 0}, cmx[5] = {0, 0, 0, 0, 0};
                    double ad_sum = 0, ad_max = 0, ab_sum = 0, ab_max = 0, lc_max = 0, ev_sum = 0;
                    std::vector<double> v_loop, v_cols, v_tot, v_ad;
                    long nw = 0, flagged = 0;
                    for (long ch = 0; ch < nchunks; ++ch) {
                        flagged += fl[ch];
                        for (int w = 0; w < S; ++w, ++nw) {
                            const long long* e = &d[((size_t)ch * kSegMax + w) * 16];
                            const double ph[6] = {(e[1] - e[0]) / 100.0, (e[2] - e[1]) / 100.0, (e[3] - e[2]) / 100.0,
                                                  (e[4] - e[3]) / 100.0, (e[5] - e[4]) / 100.0, (e[5] - e[0]) / 100.0};
                            for (int i = 0; i < 6; ++i) { sum[i] += ph[i]; mx[i] = std::max(mx[i], ph[i]); }
                            for (int i = 0; i < 5; ++i) { cnt[i] += (double)e[6 + i]; cmx[i] = std::max(cmx[i], (double)e[6 + i]); }
                            v_loop.push_back(ph[3]); v_cols.push_back((double)e[6]); v_tot.push_back(ph[5]); ev_sum += (double)e[15];
                        }
                        const long long* e = &d[(size_t)ch * kSegMax * 16];
                        const double t = (e[12] - e[11]) / 100.0;
                        v_ad.push_back(t);
                        ad_sum += t; ad_max = std::max(ad_max, t); ab_sum += (double)e[13]; ab_max = std::max(ab_max, (double)e[13]);
                        lc_max = std::max(lc_max, (double)e[14]);
                    }
                    fprintf(stderr, "[k2 debug] S=%d waves=%ld flagged chunks=%ld | k_env us avg/max: mask %.1f/%.1f scanA %.1f/%.1f sync %.1f/%.1f "
                            "loopB %.1f/%.1f flush %.1f/%.1f total %.1f/%.1f | per wave avg/max: cols %.0f/%.0f iters %.0f/%.0f evict %.0f/%.0f "
                            "refill %.0f/%.0f range %.0f/%.0f | k_addend us %.1f/%.1f batches %.1f/%.0f max owners %.0f\n",
                            S, nw, flagged, sum[0] / nw, mx[0], sum[1] / nw, mx[1], sum[2] / nw, mx[2], sum[3] / nw, mx[3], sum[4] / nw, mx[4],
                            sum[5] / nw, mx[5], cnt[0] / nw, cmx[0], cnt[1] / nw, cmx[1], cnt[2] / nw, cmx[2], cnt[3] / nw, cmx[3],
                            cnt[4] / nw, cmx[4], ad_sum / nchunks, ad_max, ab_sum / nchunks, ab_max, lc_max);
                    {  // the blocks that finish their addend pass last: where their time went
                        long long t0 = 0x7fffffffffffffffll;
                        for (long ch = 0; ch < nchunks; ++ch) for (int w = 0; w < S; ++w) t0 = std::min(t0, d[((size_t)ch * kSegMax + w) * 16]);
                        std::vector<std::pair<double, long>> ends;
                        for (long ch = 0; ch < nchunks; ++ch) ends.push_back({(d[(size_t)ch * kSegMax * 16 + 12] - t0) / 100.0, ch});
                        std::sort(ends.begin(), ends.end());
                        for (size_t i = ends.size() > 6 ? ends.size() - 6 : 0; i < ends.size(); ++i) {
                            const long ch = ends[i].second;
                            double st = 1e30, scan = 0, loop = 0, envend = 0;
                            for (int w = 0; w < S; ++w) {
                                const long long* e = &d[((size_t)ch * kSegMax + w) * 16];
                                st = std::min(st, (e[0] - t0) / 100.0); scan = std::max(scan, (e[2] - e[1]) / 100.0);
                                loop = std::max(loop, (e[4] - e[3]) / 100.0); envend = std::max(envend, (e[5] - t0) / 100.0);
                            }
                            const long long* e = &d[(size_t)ch * kSegMax * 16];
                            const long long* dx = &d[((size_t)ch * kSegMax + (kSegMax - 1)) * 16];
                            fprintf(stderr, "[k2 debug] late block %ld: starts %.1f us, scan %.1f, loop %.1f, env ends %.1f, addend %.1f -> %.1f (%.1f us, %lld owners; "
                                    "max per row: %lld entries, %lld lookups, %lld list reads of which %lld from HBM, %lld owners older than the window)\n",
                                    ch, st, scan, loop, envend, (e[11] - t0) / 100.0, (e[12] - t0) / 100.0, (e[12] - e[11]) / 100.0, e[14],
                                    S <= 4 ? dx[4] : -1, S <= 4 ? dx[0] : -1, S <= 4 ? dx[2] : -1, S <= 4 ? dx[1] : -1, S <= 4 ? dx[3] : -1);
                        }
                    }
                    {
                        double mhz = 0; long nm = 0;
                        if (S > 1) for (long ch = 0; ch < nchunks; ++ch) {
                            const long long* e = &d[((size_t)ch * kSegMax + 1) * 16];
                            if (e[5] > e[0]) { mhz += (double)e[11] / ((e[5] - e[0]) / 100.0); ++nm; }
                        }
                        fprintf(stderr, "[k2 debug] shader clock during k_env: %.0f MHz (clock64 / wall_clock64 over %ld waves)\n", nm ? mhz / nm : 0.0, nm);
                    }
                    auto pct = [](std::vector<double>& v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
                    fprintf(stderr, "[k2 debug] p50/p90/p99: loopB us %.1f/%.1f/%.1f cols %.0f/%.0f/%.0f wave total us %.1f/%.1f/%.1f addend us %.1f/%.1f/%.1f; scan evals per junction %.0f\n",
                            pct(v_loop, .5), pct(v_loop, .9), pct(v_loop, .99), pct(v_cols, .5), pct(v_cols, .9), pct(v_cols, .99),
                            pct(v_tot, .5), pct(v_tot, .9), pct(v_tot, .99), pct(v_ad, .5), pct(v_ad, .9), pct(v_ad, .99),
                            ev_sum / std::max(1.0, (double)nchunks * (S - 1)));
                }
            } else {
                // LDS per block = (3 C + 3 SG) * 4R * 4 B + 4 KiB; a CU holds 160 KiB.  Small grids get the long
                // ring (fewer HBM round trips in the fill), large grids the short one (all waves resident).
                bool small_grid = nwaves <= 2048;
                if (env_lean >= 0) small_grid = env_lean == 0;
                const bool no_pf = env_serial_fill;
                if (R == 64) { if (small_grid) FDCM_K2(64, 16, 8, false); else FDCM_K2(64, 8, 4, false); }
                else if (R == 32) {
                    if (small_grid && !no_pf) FDCM_K2(32, 32, 8, true); else if (small_grid) FDCM_K2(32, 32, 16, false);
                    else if (!no_pf) FDCM_K2(32, 16, 4, true); else FDCM_K2(32, 16, 8, false);
                } else {
                    if (small_grid && !no_pf) FDCM_K2(16, 64, 4, true); else if (small_grid) FDCM_K2(16, 64, 16, false);
                    else if (!no_pf) FDCM_K2(16, 32, 4, true); else FDCM_K2(16, 32, 8, false);
                }
            }
#undef FDCM_K2
        }
    }
    FDCM_HIP(hipEventRecord(ev[3], st));
    const bool want_sqrt = fm->distance == FDCM_L2;
    if (stop_after >= 2) {
        const size_t nq = ivol_slice_floats(W, H);
        fm->ivol.reserve((size_t)m * nq * sizeof(float));
        float* ivol = fm->ivol.as<float>();
        const unsigned pblocks = (unsigned)((nq + 255) / 256);
        const int sq = (want_sqrt ? 1 : 0) | (fm->vol1_interleaved ? 2 : 0);
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
    FDCM_HIP(hipEventRecord(ev[4], st));
    if (stop_after >= 3) {
        const int chains = 2 * (W > H ? W : H);
        const int tab_stride = sh_tab_stride(W);
        fm->offtab.reserve((size_t)m * tab_stride * sizeof(int));
        int* d_tab = fm->offtab.as<int>();
        if (!(fm->off_m == m && fm->off_steps == W)) {  // the table only depends on the keys (fixed per handle) and the size
            hipLaunchKernelGGL(k_groups, dim3((unsigned)((tab_stride + 255) / 256), (unsigned)m), dim3(256), 0, st, d_int, d_tab, W, tab_stride);
            fm->off_m = m; fm->off_steps = W;
        }
        static const int env_int_only = getenv("FDCM_INT_ONLY") ? atoi(getenv("FDCM_INT_ONLY")) : 0;  // timing experiment
        static const int env_int_shw = getenv("FDCM_INT_SHW") ? atoi(getenv("FDCM_INT_SHW")) : 0;  // measurement: 1, 2 or 4
        int shw = (long)m * ((chains + kShOwn - 1) / kShOwn) > 8192 ? 4 : 1;  // working waves per workgroup of a shallow slice
        if (env_int_shw == 1 || env_int_shw == 2 || env_int_shw == 4) shw = env_int_shw;
        // steep slices: 60 own chains per block while the launch is small, 124 / 252 once such blocks would outnumber
        // what the GPU holds several times over (fewer columns read twice; see integral_steep)
        static const int env_int_xc = getenv("FDCM_INT_XC") ? atoi(getenv("FDCM_INT_XC")) : 0;  // measurement: 64 / 128 / 256
        const long narrow_blocks = (long)m * ((chains + 59) / 60), cus = device_cus(fm->device);
        const int xc = env_int_xc ? env_int_xc : (narrow_blocks > 64 * cus ? 256 : (narrow_blocks > 12 * cus ? 128 : 64));
        const dim3 igrid((unsigned)((chains + kShOwn - 1) / kShOwn), (unsigned)m);
        static const int env_int_stride = getenv("FDCM_INT_STRIDE") ? atoi(getenv("FDCM_INT_STRIDE")) : -1;  // measurement: 1 = index order
        int kstride = 1;
        // (a small launch -- config 2: 1 080 blocks, four per CU -- is faster in index order: 0.070 against 0.077 ms; the strided
        // order pays once the blocks queue for the CUs: config 3, 4 260 blocks, 0.45 - 0.49 against 0.50 - 0.51 ms)
        const bool small_launch = (long)m * igrid.x <= 6L * cus;
        if (env_int_stride != 1 && m > 2 && (!small_launch || env_int_stride > 1)) {
            auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
            kstride = m / 2 + 1;
            while (gcd(kstride, m) != 1) ++kstride;
        }
        if (env_int_stride > 1) kstride = env_int_stride;
#define FDCM_INTEGRAL(XC)                                                                                                        \
        do {                                                                                                                     \
            constexpr size_t lds = integral_lds_bytes<XC>();                                                                     \
            static_assert(lds <= 160 * 1024, "tile pair must fit a CU's LDS");                                                   \
            if (lds > 64 * 1024)                                                                                                 \
                FDCM_HIP(hipFuncSetAttribute((const void*)k_integral<XC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL(k_integral<XC>, igrid, dim3(256), lds, st, (const float*)fm->ivol.as<float>(), vol, W, H, d_int,  \
                               d_tab, env_int_only, shw, kstride);                                                               \
        } while (0)
        if (xc == 256) FDCM_INTEGRAL(256); else if (xc == 128) FDCM_INTEGRAL(128); else FDCM_INTEGRAL(64);
#undef FDCM_INTEGRAL
    }
    fm->vol_stage = stop_after >= 3 ? 3 : (stop_after == 2 ? 2 : 1);
    FDCM_HIP(hipEventRecord(ev[5], st));
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
    FDCM_HIP(hipEventElapsedTime(&bt.seeds_ms, ev[0], ev[1]));
    FDCM_HIP(hipEventElapsedTime(&bt.pass1_ms, ev[1], ev[2]));
    FDCM_HIP(hipEventElapsedTime(&bt.pass2_ms, ev[2], ev[3]));
    FDCM_HIP(hipEventElapsedTime(&bt.propagate_ms, ev[3], ev[4]));
    FDCM_HIP(hipEventElapsedTime(&bt.integral_ms, ev[4], ev[5]));
    float span = 0.f;
    FDCM_HIP(hipEventElapsedTime(&span, ev[0], ev[5]));
    bt.total_ms = fm->build_host_ms + span;  // host preparation + the kernels' span on the device
}

}  // namespace fdcm
