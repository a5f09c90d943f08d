// fdcm_sweep_literal.hip -- the reference's second 1-D pass of distanceTransform<float, L2 / L2_SQUARED> followed literally
// (imgproc.h:91-130), one wave per (slice, 64-row chunk): the sweep for feature sizes where values of the pass are not
// all exact integers in float (W^2 + H^2 > 2^24: fdcm_sweep.hip does the others) -- nothing is assumed about the
// arithmetic here, every test of the reference is made on the reference's operands in the reference's order.
#include <algorithm>

#include "fdcm_build_dev.h"
#include "fdcm_sweep.h"

namespace fdcm {

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
                                                  float* __restrict__ sf, float* __restrict__ sz) {
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
    const int sub = (int)(wid - chunk * SUB);
    const long k = chunk / HW64;
    const int c = (int)(chunk - k * HW64);
    const int bit = sub * R + (lane & (R - 1));  // row inside the chunk = bit of the seed word
    const int urow = wave * R + (lane & (R - 1));  // row inside the block (LDS column)
    const int y = c * 64 + bit;
    const long gid = wid * R + (lane & (R - 1));  // scratch row
    const uint4* dp = reinterpret_cast<const uint4*>(desc + ((size_t)k * HW64 + c) * W);
    // element x of the row in the interleaved layout (ivol_index) that the propagation reads: row[((x / 4) * H) * 4 + x % 4]
    float* row = vol + (size_t)k * ivol_slice_floats(W, H) + (size_t)(y < H ? y : 0) * 4;
    const size_t H_ = (size_t)H, NT = (size_t)nwaves * R;
    auto xoff = [&](int x) -> size_t { return ((size_t)(x >> 2) * H_) * 4 + (size_t)(x & 3); };
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

size_t sweep_literal_scratch_bytes(int W, long nchunks) { return (size_t)W * (size_t)nchunks * 64 * 12; }

void launch_sweep_literal(hipStream_t st, const void* desc_, float* vol, int W, int H, int HW64, long nchunks, void* scratch) {
    const ColDesc* desc = (const ColDesc*)desc_;
    // R = rows per wave: a small grid is cut into more waves (the chain per row is sequential); keep >= 2048 waves in flight
    int R = 64;
    while (R > 16 && nchunks * (64 / R) < 2048) R >>= 1;
    const long nwaves = nchunks * (64 / R);
    const unsigned wblocks = (unsigned)((nwaves + 3) / 4);
    // the (v, f, z) spill space: [W][nwaves * R] each
    int* sv = (int*)scratch;
    float* sf = (float*)(sv + (size_t)W * nwaves * R);
    float* sz = sf + (size_t)W * nwaves * R;
#define FDCM_K2(RR, CC, SS, PP) hipLaunchKernelGGL((k_pass2_l2<RR, CC, SS, PP>), dim3(wblocks), dim3(256), 0, st, desc, vol, W, H, HW64, nwaves, sv, sf, sz)
    // LDS per block = (3 C + 3 SG) * 4R * 4 B + 4 KiB; a CU holds 160 KiB.  Small grids get the long ring (fewer HBM round
    // trips in the fill), large grids the short one (all waves resident).
    const bool small_grid = nwaves <= 2048;
    if (R == 64) { if (small_grid) FDCM_K2(64, 16, 8, false); else FDCM_K2(64, 8, 4, false); }
    else if (R == 32) { if (small_grid) FDCM_K2(32, 32, 8, true); else FDCM_K2(32, 16, 4, true); }
    else { if (small_grid) FDCM_K2(16, 64, 4, true); else FDCM_K2(16, 32, 4, true); }
#undef FDCM_K2
}

}  // namespace fdcm
