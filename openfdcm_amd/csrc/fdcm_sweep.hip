// fdcm_sweep.hip -- the balanced L2 / L2^2 sweep: both 1-D passes of distanceTransform<float, L2 / L2_SQUARED>
// (imgproc.h:91-130, :178-193) for feature sizes with W^2 + H^2 <= 2^24.
//
// What the reference's second pass computes, when every number in it is exact.  Pass 1 leaves integers: the squared
// distance to the column's nearest seed (or FLT_MAX for a column without one, which never owns a pixel once any column
// has a seed).  With W^2 + H^2 <= 2^24 every numerator (f[q] + q^2) - f[v] - v^2 of imgproc.h:111 is an exact integer,
// every denominator 2 (q - v) <= 2 (W - 1) too, so the only rounded quantity is the quotient s = RN(N / D), and RN is
// monotone.  Two facts follow (DESIGN.md section 4 has the proof; tools/sim/exact_owner_sim.cpp checks it against the
// literal pass on the BASELINE scenes and on random / near-degenerate columns):
//   (1) for an integer pixel q <= W - 1 < 2^12:  q <= RN(N / D)  <=>  q <= N / D   (q - N / D is 0 or at least 1 / D, far
//       above half an ulp of q), so the fill's `while (z[k + 1] < q)` (:124) classifies pixels as exact arithmetic would;
//   (2) a stack the float construction keeps has strictly increasing z, hence strictly increasing exact intersections:
//       it is an exactly convex chain, and by (1) the pixels it hands its entries are those of exact arithmetic.
//   => the owner of every pixel in the reference's run is the EXACT owner: the seeded column u minimising
//      f[u] + (q - u)^2 over the integers, the smallest u on a tie.  The same holds for the reference's construction run on
//      ANY subset of the columns that contains those owners.
// So a row may be cut anywhere -- and the cuts need not be known in advance.  The waves start on S column ranges of equal
// seeded-column count; columns are handed out in blocks of ~8 through a bit map in LDS, a wave goes on through the blocks
// behind its own until it meets one that has an owner, and a wave that has run dry begins a NEW range in the middle of the
// longest stretch nobody has started (up to kMaxR = 16 ranges per row): the slowest wave of a workgroup -- the one whose
// columns make some row pop a lot, 3 x the mean on unlucky scenes -- is relieved of the far half of what it has left.
// Each range runs the literal construction on its own stack (bottom = its first column, z = -inf), the ranges are put in
// column order, and the stacks are then merged from left to right by
// landing the next range's entries on the accumulated stack (pop while s <= z, as the reference would) until one of
// them stays on its local predecessor -- which IS the reference's construction on the union of the local stacks, a set
// that contains every pixel owner.  No junction search, no speculation, no redo; the longest wave of a block holds
// n / S columns whatever the scene looks like.  The fill with the in-place read-back (:126-127) is unchanged:
// the owner list (first pixel, column, addend) of a row, addend = f[v] or the already written g[v].
//
// (Dynamic cuts shorten the heaviest workgroup's chain and add junctions: run_build turns them on where the kernel lasts as
// long as its slowest workgroup -- all workgroups resident, one blocking build on the GPU -- and leaves the ranges of equal
// count where workgroups queue for the CUs or the frames of a pipeline share them.)
//
// One workgroup per (slice, 64-row chunk), kSeg waves, lane = row:
//   local run   wave w: the literal construction over its columns; stack = top entry in registers + a ring of kRing
//               entries per row in LDS + HBM scratch behind it (row-major); everything also lands in HBM for the walk
//   merge       8 rows per wave, 8 lanes per row: the junctions from left to right, 8 entries per step (rings, HBM behind)
//   owner walk  same layout: 8 stack entries of a row per step -> owner list (addend chains by pointer doubling)
//   fill        all waves, W / kSeg pixels each, 16-byte units of the interleaved layout [k][x/4][y][x%4]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fdcm_build_dev.h"
#include "fdcm_quotient.h"
#include "fdcm_sweep.h"

namespace fdcm {

static constexpr int kSeg = kSweepSegments;  // waves per block = column ranges per row = fill parts
static constexpr int kNT = 64 * kSeg;
static constexpr int kRing = 8;      // stack entries per (row, range) below the top kept in LDS
static constexpr int kRE = 10;       // owner entries per row and round of the fill (three words each in LDS)
[[maybe_unused]] static constexpr int kLabN = 24;     // lab builds: 64-bit words per (chunk, wave) record
static constexpr int kMinCols = 16;  // a range holds at least this many seeded columns (fewer ranges on small slices); FDCM_SWEEP_MINCOLS

// The LDS ring of stack entries, three planes of consecutive dwords: entry i of workgroup lane c is
// (2 v, P = f + v^2, z) = plane[0..2][i & (kRing - 1)][c].  In the local run a lane reads and writes its own column c (bank = c mod 32
// whatever i is: conflict-free, which is what the layout is for).  The merge and the walk read 8 consecutive entries of ONE row
// with 8 lanes: strides of kNT = 512 dwords, all on one bank (8-way conflicts; 3 % of the kernel's wave cycles -- a stride of
// 516 would clear them and put lanes of the pop loop that stand at different depths on one bank instead: profiles/NOTES.md section 12).
struct Ring {
    float* p;
    static constexpr int kPlane = kRing * kNT;  // floats per plane
    __device__ __forceinline__ float4 get(int i, int c) const {
        const float* a = p + (i & (kRing - 1)) * kNT + c;
        return make_float4(a[0], a[kPlane], a[2 * kPlane], 0.f);
    }
    __device__ __forceinline__ void put(int i, int c, float v2, float P, float z) const {
        float* a = p + (i & (kRing - 1)) * kNT + c;
        a[0] = v2; a[kPlane] = P; a[2 * kPlane] = z;
    }
    __device__ __forceinline__ void put_z(int i, int c, float z) const { p[2 * kPlane + (i & (kRing - 1)) * kNT + c] = z; }
};

// Ranges of a row.  A workgroup starts with S <= kSeg ranges of equal column count, one per wave; a wave that runs out of
// columns takes over the far half of the longest stretch of columns nobody has started yet, with a fresh stack, as a new range
// (dynamic cuts: the exact-owner theorem allows any cut).  Columns are handed out in blocks: a wave claims the next block of its
// stretch when it gets there (an atomic OR on a bit map in LDS), and stops where somebody else's range begins.
static constexpr int kMaxR = 16;     // ranges per row at most (initial + taken over)
static constexpr int kMaxBlk = 128;  // claim blocks per slice at most (two 64-bit words)
static constexpr int kStealMin = 1;   // blocks an unclaimed stretch must have for a wave out of columns to begin a new range in its middle (FDCM_SWEEP_STEAL)
                                      // (mean of config 2's four scenes: 1 block 0.196 ms, 2 blocks 0.198 -- with an occasional 0.267 on seed 4 --, 3: 0.207, 6: 0.225)
static constexpr int kStealCols = 8;  // .. and columns  // blocks an unclaimed stretch must have for a wave out of columns to begin a new range in its middle (FDCM_SWEEP_STEAL)
struct SweepLds {
    unsigned long long smask[64];    // the slice's seeded columns, 64 per word (W <= 4096)
    unsigned long long claim[2];     // blocks of columns that have an owner
    unsigned short blk_q[kMaxBlk + 2];  // column (position) at which block b begins; blk_q[nblk] = W
    short t_cnt[kMaxR][64];          // entries of the local stack of (range, row)
    short t_base[kMaxR][64];         // of which [t_base, t_cnt) are in the LDS ring (all of them are in HBM too)
    short t_lo[kMaxR][64], t_hi[kMaxR][64];  // after the merge: the entries [t_lo, t_hi] of the local stack are on the row's stack
    signed char t_prev[kMaxR][64];   // the range below this one on the row's stack at the time it landed
    short t_K[kMaxR][64];            // the walk: stream index -> HBM slot offset of (range, row)
    int s_slot0[kMaxR];              // first HBM slot of a range's entries (its first column: ranges are disjoint); INT_MAX: no such range
    int r_wave[kMaxR];               // the wave whose ring columns hold the range's top entries
    int r_pos[kMaxR];                // rank of the range by first column (the merge takes them from left to right)
    int n_ranges;                    // range ids handed out so far
    int s_lcount[64];                // owner entries per row
    int s_pi[kSeg][64];              // [p - 1][row]: list index that owns the first pixel of fill part p
};

// position of the t-th set bit (t from 0) of the mask words; wave-uniform
__device__ __forceinline__ int select_column(const unsigned long long* smask, int t) {
    int b = 0;
    unsigned long long mk = uni64(smask[0]);
    while (__popcll(mk) <= t) { t -= __popcll(mk); ++b; mk = uni64(smask[b]); }
    for (; t > 0; --t) mk &= mk - 1ull;
    return b * 64 + __ffsll((long long)mk) - 1;
}

// the same for a rank that differs from lane to lane (no scalar broadcast of the mask words; the bit inside the word by halving)
__device__ __forceinline__ int select_column_lane(const unsigned long long* smask, int t) {
    int b = 0;
    unsigned long long mk = smask[0];
    while (__popcll(mk) <= t) { t -= __popcll(mk); ++b; mk = smask[b]; }
    int pos = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        const int c = __popcll(mk & ((1ull << w) - 1ull));
        if (t >= c) { t -= c; mk >>= w; pos += w; }
    }
    return b * 64 + pos;
}

// ---- the literal construction over the seeded columns [q0, ql] of the slice, bottom = q0 (imgproc.h:100-121)
// Stack entries in registers and in the ring are (float(2 v), P = f[v] + v^2, z): the numerator of imgproc.h:111,
// ((f[q] + q^2) - f[v]) - v^2, is the exact integer P_q - P_v whatever the order (every term is an integer below 2^24), so
// the test takes one subtraction; memory holds the same three numbers (f[v] = P - v^2 comes back exactly where the owner walk wants it).
__device__ __forceinline__ EnvEntry to_mem(const float4& e) { return EnvEntry{e.x, e.y, e.z}; }
__device__ __forceinline__ float4 from_mem(const EnvEntry& e) { return make_float4(e.v2, e.P, e.z, 0.f); }
// ---- end of a local run: the top joins the entries; everything in the ring also goes to HBM (the ring keeps its content
// for the merge)
__device__ __forceinline__ void local_finish(const Ring ring, EnvEntry* __restrict__ ent, int tid, float tvx2, float tP, float tz, int& cnt, int& base) {
    if (cnt - base == kRing) { ent[base] = to_mem(ring.get(base, tid)); ++base; }
    ring.put(cnt, tid, tvx2, tP, tz);
    ++cnt;
#pragma unroll
    for (int e = 0; e < kRing; ++e) {
        const int i = base + e;
        if (i < cnt) ent[i] = to_mem(ring.get(i, tid));
    }
    // The merge looks at the 8 entries below a range's top first.  After a run of pops the ring holds fewer than that (it is
    // only refilled when empty): the missing ones come back from HBM now, all lanes and entries in one trip, instead of one
    // trip per junction later.
    const int want = max(cnt - kRing, 0);
    if (__builtin_amdgcn_ballot_w64(base > want) != 0ull) {
        const EnvEntry t0 = ent[max(base - 1, 0)], t1 = ent[max(base - 2, 0)], t2 = ent[max(base - 3, 0)], t3 = ent[max(base - 4, 0)],
                       t4 = ent[max(base - 5, 0)], t5 = ent[max(base - 6, 0)], t6 = ent[max(base - 7, 0)];
        auto put = [&](const EnvEntry& e, int idx) { if (idx >= want) { const float4 m = from_mem(e); ring.put(idx, tid, m.x, m.y, m.z); } };
        put(t0, base - 1); put(t1, base - 2); put(t2, base - 3); put(t3, base - 4); put(t4, base - 5); put(t5, base - 6); put(t6, base - 7);
        base = min(base, want);
    }
}
// A block of columns gets its owner: true when this wave is the one (wave-uniform; one LDS atomic by lane 0)
__device__ __forceinline__ bool claim_block(SweepLds& L, int b, int lane) {
    unsigned long long old = 0ull;
    if (lane == 0) old = atomicOr(&L.claim[b >> 6], 1ull << (b & 63));
    old = uni64(old);
    return ((old >> (b & 63)) & 1ull) == 0ull;
}
// The run starts at the first column of block b0 (claimed by the caller) and goes on through the blocks behind it for as long
// as it is the first to claim them: it ends where another range begins, or at the slice's last column.
// (bend < 0: dynamic; bend >= 0: no claims at all, the run covers the blocks [b0, bend) -- ranges of equal count, FDCM_SWEEP_STEAL=0)
__device__ __forceinline__ void local_run(const uint4* __restrict__ dp, int W, SweepLds& L, int b0, int nblk, int bend, int lane,
                                          int y, int tid, const Ring ring, EnvEntry* __restrict__ ent_row, int& cnt_out, int& base_out, int& q0_out, long long* lab) {
    const unsigned long long* smask = L.smask;
    const int q0 = __builtin_amdgcn_readfirstlane((int)L.blk_q[b0]);
    EnvEntry* __restrict__ ent = ent_row + q0;
    q0_out = q0;
#ifdef FDCM_LAB
    long long lab_cols = 0, lab_pop = 0, lab_evict = 0;
    const long long lab_t0 = __builtin_amdgcn_s_memtime();
#endif
    const float inf = f_inf();
    const uint4 db = dp[q0];
    // top entry t and the entry below it u (a register copy of ring entry cnt - 1, so that a single pop needs no LDS round trip)
    float tvx2, tP, tz = -inf;
    {
        const float vf = (float)q0;
        tvx2 = vf + vf;
        tP = column_value_sq_seeded(((unsigned long long)db.y << 32) | db.x, (int)db.z, (int)db.w, lane, y) + vf * vf;
    }
    // u = (uv, up, uz): register copy of ring entry cnt - 1 (a single pop needs no LDS round trip)
    float uv = 0.f, up = 0.f, uz = 0.f;
    int cnt = 0;   // entries below the top (indices 0..cnt-1); [base, cnt) in the LDS ring, [0, base) in HBM
    int base = 0;
    auto evict = [&]() {
        ent[base] = to_mem(ring.get(base, tid));
        ++base;
    };
    const unsigned lanebase = (unsigned)(size_t)(ring.p + tid);  // LDS byte address of this lane's ring entry 0, plane 0 (entry i: + i * 2048; planes 16384 apart)
    static_assert(kNT * sizeof(float) == 2048 && Ring::kPlane * sizeof(float) == 16384, "the pop loop below shifts the ring index by 11 and has the plane offsets written out");
    const int qlo = q0 + 1;
    int qhi = __builtin_amdgcn_readfirstlane((int)L.blk_q[bend >= 0 ? bend : b0 + 1]) - 1;  // last position of the stretch claimed so far
    if (bend >= 0) nblk = 0;  // (nothing to ask for)
    {
        // The range's seeded columns in order (columns without a seed in the slice never own a pixel: skipped).  A column's
        // descriptor is the same 16 bytes for every lane: a scalar load into SGPRs (left to itself the compiler fetches the seed
        // word with a vector load and waits for every store in flight), issued one column ahead -- it lands behind the waits
        // of this column's pop loop instead of being waited for on the spot (~250 cycles per column).
        int bcur = b0;
        int whi = qhi >> 6;
        int wd = min(qlo >> 6, whi);
        unsigned long long mk = qlo <= qhi ? uni64(smask[wd]) & (~0ull << (qlo & 63)) : 0ull;
        if (wd == whi) mk &= ~0ull >> (63 - (qhi & 63));
        // The block behind the current one is asked for on entering the current one: the atomic's answer and the block's bounds
        // are in registers long before the stretch runs out (they land behind the waits of the pop loops in between), so the
        // hand-over from block to block costs no trip to LDS on the column chain.  (A wave so holds its block and the next.)
        unsigned long long pend_old = 0ull;  // lane 0: the claim word as it was before this wave's OR
        unsigned pend_q = 0u, pend_qn = 0u;  // first column of block bcur + 1, of block bcur + 2
        auto ask_next = [&]() {
            if (bcur + 1 < nblk) {
                if (lane == 0) pend_old = atomicOr(&L.claim[(bcur + 1) >> 6], 1ull << ((bcur + 1) & 63));
                pend_q = L.blk_q[bcur + 1]; pend_qn = L.blk_q[bcur + 2];
            }
        };
        ask_next();
        auto advance = [&]() {  // to the next word of the stretch with a seeded column; at its end, on into the next block if it is this wave's
            for (;;) {
                while (mk == 0ull && wd < whi) {
                    ++wd;
                    mk = uni64(smask[wd]);
                    if (wd == whi) mk &= ~0ull >> (63 - (qhi & 63));
                }
                if (mk != 0ull || bcur + 1 >= nblk) return;
                if (((uni64(pend_old) >> ((bcur + 1) & 63)) & 1ull) != 0ull) return;  // it had an owner already: the range ends here
                ++bcur;
                const int nq = __builtin_amdgcn_readfirstlane((int)pend_q);
                qhi = __builtin_amdgcn_readfirstlane((int)pend_qn) - 1;
                whi = qhi >> 6;
                wd = nq >> 6;
                mk = uni64(smask[wd]) & (~0ull << (nq & 63));
                if (wd == whi) mk &= ~0ull >> (63 - (qhi & 63));
                ask_next();
            }
        };
        advance();
        u32x4 dq;
        int q = -1;
        if (mk) {
            q = wd * 64 + __ffsll((long long)mk) - 1;
            mk &= mk - 1ull;
            advance();
            asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dq) : "s"(dp + q) : "memory");
        } else {
            dq = u32x4{0u, 0u, 0u, 0u};
        }
        while (q >= 0) {
            {
                const unsigned long long wc = ((unsigned long long)dq.y << 32) | dq.x;
                const int pc = (int)dq.z, nc = (int)dq.w;
                const float qf = (float)q;
                u32x4 dqn = dq;
                if (mk) {  // the next column's descriptor: in flight until the pop loop's waits.  The compiler does not know that:
                           // dqn is not named again before the `+s` statement behind the loop, and tools/check_sweep_prefetch.py
                           // (tests/test_capi_load.py) reads the built code to see that nothing touches its SGPRs before that wait
                    q = wd * 64 + __ffsll((long long)mk) - 1;
                    mk &= mk - 1ull;
                    advance();
                    asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(dqn) : "s"(dp + q) : "memory");
                } else {
                    q = -1;
                }
                const float fq = column_value_sq_seeded(wc, pc, nc, lane, y);
                const float hq = fq + qf * qf;  // P of this column (q * q rounds like the reference's float(long(q * q)))
                const float twoq = qf + qf;
                float s;
                // The pop loop, written out and without exec masks.  s = ((f[q] + q^2) - f[v] - v^2) / (2q - 2v) (imgproc.h:111)
                // by envelope_quotient's four instructions (fdcm_quotient.h); pop while s <= z[k] (the bottom entry's z is -inf
                // and s is finite, so the bottom is never popped); the wave repeats the pass while any lane pops (a lane that
                // does not recomputes the same s).  A pop is four selects from the register copy u; then EVERY lane reads the
                // entry below its top from the ring again (for a lane that kept its top that is the u it holds), and the read
                // is only waited for at the next pass's selects, behind its quotient.  One scalar round trip per pass (does any
                // lane pop?) plus the test for the rare refill, placed behind the reads.
                // The loop leaves with flag = 1 when a popping lane's ring ran empty above the stack's bottom (0 < cnt == base after
                // the pop): those lanes have pend = 1, took their pop, and get their refill and u below.
                int flag;
#ifdef FDCM_LAB
                ++lab_cols;
                const long long lab_p0 = __builtin_amdgcn_s_memtime();
#endif
                do {
                    int pend, c1;
                    unsigned long long sx, sy;
                    float qd, qn, qr, qe;
                    unsigned ua;
                    asm volatile(
                        "s_mov_b32 %[flag], 0\n\t"
                        "v_mov_b32 %[pend], 0\n"
                        "L_fdcm_pop_%=:\n\t"
                        "v_sub_f32 %[qd], %[twoq], %[tv]\n\t"
                        "v_sub_f32 %[qn], %[hq], %[tp]\n\t"
                        "v_rcp_f32 %[qr], %[qd]\n\t"
                        "v_add_u32 %[c1], -1, %[cnt]\n\t"
                        "v_mul_f32 %[s], %[qn], %[qr]\n\t"
                        "v_fma_f32 %[qe], -%[qd], %[s], %[qn]\n\t"
                        "v_fmac_f32 %[s], %[qe], %[qr]\n\t"
                        "v_cmp_le_f32 vcc, %[s], %[tz]\n\t"
                        "s_cbranch_vccz L_fdcm_done_%=\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "v_cndmask_b32 %[tv], %[tv], %[uv], vcc\n\t"
                        "v_cndmask_b32 %[tp], %[tp], %[up], vcc\n\t"
                        "v_cndmask_b32 %[tz], %[tz], %[uz], vcc\n\t"
                        "v_cndmask_b32 %[cnt], %[cnt], %[c1], vcc\n\t"
                        "v_add_u32 %[ua], -1, %[cnt]\n\t"
                        "v_cmp_eq_u32 %[sx], %[cnt], %[base]\n\t"
                        "v_cmp_lt_i32 %[sy], 0, %[cnt]\n\t"
                        "v_and_b32 %[ua], 7, %[ua]\n\t"
                        "v_lshl_add_u32 %[ua], %[ua], 11, %[lb]\n\t"
                        "ds_read_b32 %[uv], %[ua]\n\t"
                        "ds_read_b32 %[up], %[ua] offset:16384\n\t"
                        "ds_read_b32 %[uz], %[ua] offset:32768\n\t"
                        "s_and_b64 %[sx], %[sx], %[sy]\n\t"
                        "s_and_b64 %[sx], %[sx], vcc\n\t"
                        "s_cbranch_scc0 L_fdcm_pop_%=\n\t"
                        "v_cndmask_b32 %[pend], 0, 1, %[sx]\n\t"
                        "s_mov_b32 %[flag], 1\n"
                        "L_fdcm_done_%=:\n\t"
                        "s_waitcnt lgkmcnt(0)"
                        : [s] "=&v"(s), [tv] "+v"(tvx2), [tp] "+v"(tP), [tz] "+v"(tz), [uv] "+v"(uv), [up] "+v"(up), [uz] "+v"(uz), [cnt] "+v"(cnt),
                          [pend] "=&v"(pend), [flag] "=&s"(flag), [sx] "=&s"(sx), [sy] "=&s"(sy), [c1] "=&v"(c1), [qd] "=&v"(qd), [qn] "=&v"(qn), [qr] "=&v"(qr), [qe] "=&v"(qe),
                          [ua] "=&v"(ua)
                        : [twoq] "v"(twoq), [hq] "v"(hq), [base] "v"(base), [lb] "v"(lanebase)
                        : "vcc", "scc", "memory");
                    if (flag) {  // wave-uniform
                        if (pend && cnt > 0) {
                            if (cnt == base) {  // ring empty: up to four spilled entries come back together
                                // all four are written (the ring is empty; entries below 0 land in free slots): no load stays pending
                                const EnvEntry e0 = ent[max(base - 1, 0)], e1 = ent[max(base - 2, 0)], e2 = ent[max(base - 3, 0)], e3 = ent[max(base - 4, 0)];
                                const float4 m0 = from_mem(e0), m1 = from_mem(e1), m2 = from_mem(e2), m3 = from_mem(e3);
                                ring.put(base - 1, tid, m0.x, m0.y, m0.z);
                                ring.put(base - 2, tid, m1.x, m1.y, m1.z);
                                ring.put(base - 3, tid, m2.x, m2.y, m2.z);
                                ring.put(base - 4, tid, m3.x, m3.y, m3.z);
                                base = max(base - 4, 0);
                            }
                            const float4 e = ring.get(cnt - 1, tid);
                            uv = e.x; up = e.y; uz = e.z;
                        }
                    }
                } while (flag);
#ifdef FDCM_LAB
                lab_pop += __builtin_amdgcn_s_memtime() - lab_p0;
                if (__builtin_amdgcn_ballot_w64(cnt - base == kRing) != 0ull) ++lab_evict;
#endif
                if (__builtin_expect(cnt - base == kRing, 0)) evict();
                uv = tvx2; up = tP; uz = tz;
                ring.put(cnt, tid, tvx2, tP, tz);
                ++cnt;
                tP = hq; tz = s; tvx2 = twoq;
                asm volatile("; the next column's descriptor has landed (the pop loop ends with s_waitcnt lgkmcnt(0))" : "+s"(dqn));
                dq = dqn;
            }
        }
    }
#ifdef FDCM_LAB
    if (lab && lane == 0) { lab[16] = 0; lab[17] = lab_cols; lab[18] = lab_evict; lab[19] = lab_pop; lab[20] = 0; lab[21] = __builtin_amdgcn_s_memtime() - lab_t0; }
#endif
    local_finish(ring, ent, tid, tvx2, tP, tz, cnt, base);
    cnt_out = cnt; base_out = base;
}

// (Round 5 built a local run with a column cursor per lane for lab builds -- FDCM_SWEEP_LOCAL=cursors: the longest wave makes 201
// passes instead of 482 and the kernel is slower, 0.22 against 0.195 ms, because a pass with its own cursor is ~45 instructions
// against 20; profiles/NOTES.md section 11.  It was removed in round 6 when the ranges became dynamic; git history has it.)

// ---- Phase 2 lane layout: wave j works on the rows 8 j .. 8 j + 7 of the chunk, lane = 8 g + t with g the row inside
// the wave and t = 0..7.  The 8 lanes of a row hold the row's state in copies and spend their width on 8 stack entries
// at a time: the rows of a chunk do the same thing at the same place (so splitting rows over lanes buys nothing), the
// cost of a junction is the LONGEST landing among a wave's rows, and those have a heavy tail (entries popped on either
// side per row and junction: p50 1, p90 10, p99 25 - 40: tools/sim/balanced_sim.cpp).
static_assert(kSeg == 8, "phase 2 lays a wave out as 8 rows x 8 lanes");

// first pixel above z: the entry takes over there (while (z[k+1] < q) ++k, imgproc.h:124): 0 below 0, W from W - 1 on
__device__ __forceinline__ int first_pixel(float z, float Wf) { return (int)floorf(__builtin_fminf(__builtin_fmaxf(z, -1.f), Wf - 0.5f)) + 1; }

// ---- merge of the ranges' stacks of a row, left to right: the reference's construction continued with the next range's
// entries as the incoming columns.  An incoming entry is tested against the 8 entries at the top of the row's stack at
// once (it pops the leading ones whose test says so: the reference stops at the first that does not), and the 8 entries
// behind it against the entry it landed on at once (entry c + 1 pops entry c if its local z -- its quotient on c, the very
// test the reference makes -- is <= c's quotient on the landing entry; landing deeper only raises that quotient, so what
// this decides the reference decides too, and what it leaves open the next round settles).
__device__ __forceinline__ void merge_bulk(SweepLds& L, int S, int row, int t, int sh, const Ring ring, EnvEntry* __restrict__ entr, long long* lab) {
#ifdef FDCM_LAB
    long long n_iter = 0, n_hbm = 0, n_refill = 0;
#endif
    // The first 8 incoming entries of every junction are fetched together before the first one is needed (one trip to
    // memory for all of them; the sets rotate through named registers): lane t holds entry t of the range as
    // (2 v, f + v^2, local z).  (Entries 8 .. 15 only come with a refill: rows scatter over memory, and the fetches of a
    // whole chip's junctions at once cost their bytes -- the 16-entry form took 8 us here, p50.)
    struct Cand { float a2v, ahq, az, b2v, bhq, bz; };
    auto load_cand = [&](int w, int cb, bool both) -> Cand {
        Cand c{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (w < S) {
            const int nw = L.t_cnt[w][row], ws = L.s_slot0[w];
            const EnvEntry ea = entr[ws + min(cb + t, nw - 1)];
            c.a2v = ea.v2; c.ahq = ea.P; c.az = ea.z;
            if (both) {
                const EnvEntry eb = entr[ws + min(cb + t + 8, nw - 1)];
                c.b2v = eb.v2; c.bhq = eb.P; c.bz = eb.z;
            }
        }
        return c;
    };
#ifdef FDCM_LAB
    const long long lt0 = wall_clock64();
    long long lt_loop = 0, lt_first = 0;
#endif
    Cand c0 = load_cand(1, 0, false), c1 = load_cand(2, 0, false), c2 = load_cand(3, 0, false), c3 = load_cand(4, 0, false), c4 = load_cand(5, 0, false), c5 = load_cand(6, 0, false), c6 = load_cand(7, 0, false);
#ifdef FDCM_LAB
    asm volatile("; lab: the incoming entries have arrived" :: "v"(c0.a2v), "v"(c1.a2v), "v"(c2.a2v), "v"(c3.a2v), "v"(c4.a2v), "v"(c5.a2v), "v"(c6.a2v), "v"(c6.bz));
    const long long lt1 = wall_clock64();
#endif
    int ms = 0, mi = L.t_cnt[0][row] - 1, ms_lo = 0, ms_base = L.t_base[0][row], ms_slot = L.s_slot0[0];  // the top of the row's stack
    int ms_rcol = L.r_wave[0] * 64 + row;  // .. and the ring column its top entries are in
    L.t_lo[0][row] = 0;
#pragma unroll 1
    for (int w = 1; w < S; ++w) {
        const int nw = L.t_cnt[w][row], wbase = L.t_base[w][row], wslot = L.s_slot0[w];
        // window of incoming entries starting at entry cb: lane t holds cb + t (A) and, after a refill, cb + t + 8 (B)
        float A2v = c0.a2v, Ahq = c0.ahq, Az = c0.az, B2v = 0.f, Bhq = 0.f, Bz = 0.f;
        int wend = 7;  // last window index that is held
        c0 = c1; c1 = c2; c2 = c3; c3 = c4; c4 = c5; c5 = c6;
        if (S > kSeg) c6 = load_cand(w + 7, 0, false);  // (more than 8 ranges: the junctions past the seventh get theirs on the way)
        int cur = 0, cb = 0;
        bool done = false;
        float zc = 0.f;
        // the incoming entry and the local z of the entry behind it, in every lane of the row
        float c2v = __shfl(A2v, sh), chq = __shfl(Ahq, sh), nz = __shfl(Az, sh + 1);
        if (nw < 2) nz = f_inf();
#ifdef FDCM_LAB
        const long long lj0 = wall_clock64();
        bool firstit = true;
#endif
        for (;;) {
#ifdef FDCM_LAB
            ++n_iter;
            if (!firstit && lt_first == 0) lt_first = wall_clock64() - lj0;
            firstit = false;
#endif
            // ---- the incoming entry cur against the 8 entries at the top of the row's stack
            // (lane t looks at entry mi - t: (2 v, P = f + v^2, z) from the LDS ring while it is still there.  The entries below
            // the ring -- the deeper lanes, after earlier pops -- are in memory: they are only fetched when the run of pops gets
            // as far as the first of them, since the reference stops at the first entry that does not pop.  Round 5 fetched them
            // whenever a lane looked below the ring: 40 of a deep-stack workgroup's 50 merge steps made a trip to memory.)
            const int idx = mi - t;
            const bool valid = idx >= ms_lo;
            const int ci = max(idx, ms_lo);
            float4 e = ring.get(ci, ms_rcol);
            const bool deep = valid && !done && ci < ms_base;
            // s = ((f[q] + q^2) - f[v] - v^2) / (2q - 2v), left to right in float (imgproc.h:111); pop while s <= z[k]
            float s = envelope_quotient(chq - e.y, c2v - e.x);
            bool pop = valid && !deep && s <= e.z;
            unsigned m8 = (unsigned)(__builtin_amdgcn_ballot_w64(pop) >> sh) & 0xffu;
            int npop = __builtin_ctz(~m8);  // leading pops, 0..8
            if (__builtin_amdgcn_ballot_w64(deep && npop == t) != 0ull) {  // some row's pops reach below its ring: one trip for all deep lanes
#ifdef FDCM_LAB
                ++n_hbm;
#endif
                const EnvEntry h = entr[ms_slot + ci];
                float hv, hf, hz;
                asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" : "=&v"(hv), "=&v"(hf), "=&v"(hz) : "v"(h.v2), "v"(h.P), "v"(h.z));
                if (deep) e = make_float4(hv, hf, hz, 0.f);
                s = envelope_quotient(chq - e.y, c2v - e.x);
                pop = valid && s <= e.z;
                m8 = (unsigned)(__builtin_amdgcn_ballot_w64(pop) >> sh) & 0xffu;
                npop = __builtin_ctz(~m8);
            }
            const int nvalid = min(8, mi - ms_lo + 1);
            bool landed = false;
            if (!done) {
                if (npop >= nvalid) {  // everything in sight is gone (the row's first entry has z = -inf: it never pops)
                    if (mi - 8 >= ms_lo) mi -= 8;
                    else {  // the whole range: on to the range below it
                        L.t_hi[ms][row] = ms_lo - 1;
                        ms = L.t_prev[ms][row];
                        ms_lo = L.t_lo[ms][row]; ms_base = L.t_base[ms][row]; ms_slot = L.s_slot0[ms]; ms_rcol = L.r_wave[ms] * 64 + row;
                        mi = L.t_hi[ms][row];
                    }
                } else { mi -= npop; landed = true; }
            }
            // the quotient of the test on the entry landed on (lane npop looked at it)
            const int lsrc = sh + min(npop, 7);
            const float ls = __shfl(s, lsrc);
            bool moves = false;  // the entry behind pops the one that just landed (the reference's test: same operands as in the local run)
            if (landed) { zc = ls; moves = nz <= zc; done = !moves; }
            if (__builtin_amdgcn_ballot_w64(moves) != 0ull) {
                // ---- the entries behind it, 8 at once: entry cur + t + 1 pops entry cur + t if its local z <= the latter's quotient on the landing entry
                const float l2v = __shfl(e.x, lsrc), lP = __shfl(e.y, lsrc);
                const int ci = cur - cb;             // window index of the incoming entry, 0..7
                const int wi = ci + t, wn = wi + 1;  // .. of entry cur + t and of its successor (<= 15)
                const int ksrc = sh + (wi & 7), nsrc = sh + (wn & 7);
                const float kA2v = __shfl(A2v, ksrc), kAhq = __shfl(Ahq, ksrc), kB2v = __shfl(B2v, ksrc), kBhq = __shfl(Bhq, ksrc);
                const float nAz = __shfl(Az, nsrc), nBz = __shfl(Bz, nsrc);
                const float k2v = wi < 8 ? kA2v : kB2v, khq = wi < 8 ? kAhq : kBhq, nzk = wn < 8 ? nAz : nBz;
                const float sk = t == 0 ? ls : envelope_quotient(khq - lP, k2v - l2v);
                const bool adv = moves && cur + t + 1 < nw && wn <= wend && nzk <= sk;
                const unsigned a8 = (unsigned)(__builtin_amdgcn_ballot_w64(adv) >> sh) & 0xffu;
                if (moves) cur += __builtin_ctz(~a8);  // (at least one: lane 0's test is the one that said so)
                // the window ran out (the incoming entry's successor is not held any more): the next 16 from memory
                if (__builtin_amdgcn_ballot_w64(moves && cur - cb >= wend) != 0ull) {
#ifdef FDCM_LAB
                    ++n_refill;
#endif
                    if (moves && cur - cb >= wend) cb = cur;
                    const Cand c = load_cand(w, cb, true);
                    A2v = c.a2v; Ahq = c.ahq; Az = c.az; B2v = c.b2v; Bhq = c.bhq; Bz = c.bz;
                    wend = 15;
                }
                // the new incoming entry and its successor's z, to every lane of the row
                const int ni = cur - cb, nn = ni + 1;  // window indices of the incoming entry and of its successor (<= wend)
                const int s1 = sh + (ni & 7), s2 = sh + (nn & 7);
                const float rA2v = __shfl(A2v, s1), rAhq = __shfl(Ahq, s1), rB2v = __shfl(B2v, s1), rBhq = __shfl(Bhq, s1), rAz = __shfl(Az, s2), rBz = __shfl(Bz, s2);
                c2v = ni < 8 ? rA2v : rB2v; chq = ni < 8 ? rAhq : rBhq;
                nz = cur + 1 < nw ? (nn < 8 ? rAz : rBz) : f_inf();
            }
            if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
        }
#ifdef FDCM_LAB
        lt_loop += wall_clock64() - lj0;
#endif
        L.t_hi[ms][row] = mi;
        L.t_prev[w][row] = ms;
        L.t_lo[w][row] = cur;
        const int wrcol = L.r_wave[w] * 64 + row;
        if (cur >= wbase) ring.put_z(cur, wrcol, zc);
        entr[wslot + cur].z = zc;
        ms = w; mi = nw - 1; ms_lo = cur; ms_base = wbase; ms_slot = wslot; ms_rcol = wrcol;
    }
    L.t_hi[ms][row] = mi;
#ifdef FDCM_LAB
    if (lab && (threadIdx.x & 63) == 0) { lab[12] = n_iter; lab[13] = n_hbm; lab[14] = n_refill; lab[15] = ((lt1 - lt0) << 40) | (lt_loop << 20) | lt_first; }
#endif
}

// ---- the walk over the merged stack of a row -> owner list (first pixel, column, addend).  The stack of a row is the valid
// parts of its ranges' stacks back to back.  8 entries of a row per step, one per lane: first pixels, ownership and list
// positions are independent per entry; the addend of an entry that takes over behind its own column is the value the
// reference reads back at its column, g[v] = addend_o + (v - v_o)^2 with o the owner of pixel v (imgproc.h:126-127) -- a
// chain through earlier entries.  Every term is an integer below 2^24, so the chain inside a step is summed by pointer
// doubling (3 rounds for 8 entries) instead of entry by entry; the regrouping changes no bit.
constexpr int kWin = 64;        // owner entries per row kept in LDS for the look-ups
constexpr int kWinStride = 65;  // (odd: the lanes of a row read neighbouring list positions)
template <int NR>  // ranges the row's table is laid out for: 8 (no range was taken over) or kMaxR
__device__ __forceinline__ void walk_batched(SweepLds& L, int W, int S, int part_w, const SweepBuf& B, long chunk, int row, int t, int sh,
                                             unsigned (*l_pk)[kWinStride], float (*l_b)[kWinStride], int (*l_pt)[kWinStride]) {
    const size_t r = (size_t)chunk * 64 + row;
    OwnEntry* own = B.own + r * (size_t)B.lslots;
    const EnvEntry* ent = B.ent + r * (size_t)B.eslots;
    // Range table of the row: stream index i lies in range w for i in [o_w, o_{w+1}), at slot i + K_w.
    // (K lives in LDS: the compiler turns a select chain over a register array into an indexed load from scratch memory)
    int o[NR + 1];
    o[0] = 0;
#pragma unroll
    for (int w = 0; w < NR; ++w) {
        const int lo = w < S ? L.t_lo[w][row] : 0, n = w < S ? max(L.t_hi[w][row] - lo + 1, 0) : 0;
        o[w + 1] = o[w] + n;
        L.t_K[w][row] = (short)((w < S ? L.s_slot0[w] : 0) + lo - o[w]);  // (every lane of a row writes the same value)
    }
    const int total = o[NR];
    auto slot_of = [&](int i) {
        int w = 0;
#pragma unroll
        for (int j = 1; j < NR; ++j) w += i >= o[j] ? 1 : 0;  // the last range that starts at or before i (empty ranges in between start there too)
        return i + L.t_K[w][row];
    };
    struct WalkEntry { int v; float f; float z; };  // column, f[v], z (the entries in memory hold 2 v and f[v] + v^2)
    auto load = [&](int i) -> WalkEntry {
        const EnvEntry m = ent[slot_of(min(i, total - 1))];
        const float vf = 0.5f * m.v2;
        return WalkEntry{(int)vf, m.P - vf * vf, m.z};
    };
    const float Wf = (float)W;
    int lc = 0;    // owner entries of the row so far
    int optr = 0;  // list index of the owner of the column looked up last (columns only grow, so do the owners)
    const int tmax = __builtin_amdgcn_readfirstlane(wave_max(total));
    WalkEntry en = load(t), nx = load(8 + t);
    for (int i0 = 0; i0 < tmax; i0 += 8) {
        const WalkEntry e = en;
        en = nx;
        nx = load(i0 + 16 + t);  // in flight during the next step
        const int i = i0 + t;
        const bool valid = i < total;
        // z of the entry behind: the next lane's, the next step's first for the last lane
        const float zn_in = __shfl(e.z, sh + min(t + 1, 7)), zn_nx = __shfl(en.z, sh);
        const float zn = t < 7 ? zn_in : zn_nx;
        const int st = first_pixel(e.z, Wf), stn = i + 1 < total ? first_pixel(zn, Wf) : W;
        const bool owns = valid && st < stn;  // owner of q = the last entry with z < q (imgproc.h:124): the pixels [st, stn)
        const unsigned m8 = (unsigned)(__builtin_amdgcn_ballot_w64(owns) >> sh) & 0xffu;
        const int pos = lc + __popc(m8 & ((1u << t) - 1u));
        const int lc_new = lc + __popc(m8);
        const int win_lo = lc_new - kWin;  // list positions from here on are in the LDS window
        const bool quirk = owns && st > e.v;  // takes over behind its own column: the addend is the value already written at e.v
        const unsigned pk = ((unsigned)st << 16) | (unsigned)e.v;
        if (owns) l_pk[row][pos & (kWin - 1)] = pk;
        auto list_pk = [&](int j, bool need) -> unsigned {
            unsigned v = l_pk[row][j & (kWin - 1)];
            if (__builtin_amdgcn_ballot_w64(need && j < win_lo) != 0ull) {  // older than the window (rare): from the list in HBM, consumed in place
                const unsigned a0 = own[max(j, 0)].pk;
                unsigned hv;
                asm volatile("v_mov_b32 %0, %1" : "=v"(hv) : "v"(a0));
                if (j < win_lo) v = hv;
            }
            return v;
        };
        // the owner of pixel e.v: the last list entry whose first pixel is <= e.v, between the previous look-up's answer and pos - 1
        int plo = optr, phi = pos - 1;
        while (__builtin_amdgcn_ballot_w64(quirk && plo < phi) != 0ull) {
            const bool act = quirk && plo < phi;
            const int mid = (plo + phi + 1) >> 1;
            const unsigned mpk = list_pk(mid, act);
            if (act) { if ((int)(mpk >> 16) <= e.v) plo = mid; else phi = mid - 1; }
        }
        float bval = e.f;
        int ptr = -1;
        if (__builtin_amdgcn_ballot_w64(quirk) != 0ull) {
            const unsigned ppk = list_pk(plo, quirk);
            float pb = l_b[row][plo & (kWin - 1)];
            if (__builtin_amdgcn_ballot_w64(quirk && plo < win_lo) != 0ull) {
                const float a0 = own[max(plo, 0)].b;
                float hb;
                asm volatile("v_mov_b32 %0, %1" : "=v"(hb) : "v"(a0));
                if (plo < win_lo) pb = hb;
            }
            if (quirk) {
                const float dv = (float)(e.v - (int)(ppk & 0xffffu));  // dv * dv rounds like float(long(dv * dv))
                bval = dv * dv;
                if (plo < lc) bval = pb + bval;  // an entry of an earlier step: final
                else ptr = plo;                  // an entry of this step: summed below
            }
        }
        if (owns) { l_b[row][pos & (kWin - 1)] = bval; l_pt[row][pos & (kWin - 1)] = ptr; }
        // pointer doubling inside the step: b_i = val_i + b_{ptr_i}; all lanes read before any lane writes
#pragma unroll 1
        for (int rd = 0; rd < 3; ++rd) {
            if (__builtin_amdgcn_ballot_w64(ptr >= 0) == 0ull) break;
            const bool act = ptr >= 0;
            const float pb = l_b[row][ptr & (kWin - 1)];
            const int pp = l_pt[row][ptr & (kWin - 1)];
            asm volatile("; the reads of round %0 are complete before its writes" ::"v"(pb), "v"(pp));
            if (act) { bval += pb; ptr = pp; l_b[row][pos & (kWin - 1)] = bval; l_pt[row][pos & (kWin - 1)] = ptr; }
        }
        if (owns) own[pos] = OwnEntry{pk, bval};
        // the owner of the first pixel of a fill part
        {
            const int lo_st = __builtin_amdgcn_readfirstlane(wave_min(owns ? st : 0x7fffffff)), hi_st = __builtin_amdgcn_readfirstlane(wave_max(owns ? stn : -1));
            for (int p = max(1, (lo_st + part_w - 1) / part_w); p < kSeg && p * part_w < hi_st; ++p) {
                const int x = p * part_w;
                if (owns && st <= x && x < stn) L.s_pi[p - 1][row] = pos;
            }
        }
        // the next step's look-ups start at the owner found for the last entry of this one that looked
        const unsigned q8 = (unsigned)(__builtin_amdgcn_ballot_w64(quirk) >> sh) & 0xffu;
        const int qtop = q8 ? 31 - __builtin_clz(q8) : 0;
        const int po = __shfl(plo, sh + qtop);
        if (q8) optr = po;
        lc = lc_new;
    }
    L.s_lcount[row] = lc;
}

// ---- pure fill (imgproc.h:122-128) from the owner list; wave p of a block fills the pixels
// [p * part_w, (p + 1) * part_w) of the block's 64 rows
__device__ __forceinline__ void fill_part(SweepLds& L, float* __restrict__ vol, int W, int H, long k, int c, long chunk, int part_w, const SweepBuf& B, int p,
                                          int (*f_st)[kNT], float (*f_vf)[kNT], float (*f_b)[kNT]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int y = c * 64 + lane;
    const size_t r = (size_t)chunk * 64 + lane;
    int qcur = p * part_w;
    const int qend = min(qcur + part_w, W);
    if (qcur >= qend) return;  // (last phase of the kernel: nothing waits for this wave any more)
    int idx = p == 0 ? 0 : L.s_pi[p - 1][lane];
    const int lc = L.s_lcount[lane];
    const OwnEntry* own = B.own + r * (size_t)B.lslots;
    // The fill writes the interleaved layout (ivol_index: 16 bytes = 4 neighbouring columns of one row) that the
    // propagation reads: a lane computes the values of a group of 4 columns and stores them as one unit, 64 rows = 1 KB
    // contiguous per wave.  Parts start on a group (part_w is a multiple of 4) and rounds are whole groups.
    const size_t sl = ivol_slice_floats(W, H);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(vol + (size_t)k * sl, 0, (unsigned)(sl * 4), 0x00020000);
    const unsigned vrow = y < H ? (unsigned)y * 16u : 0x80000000u;  // rows past the image: dropped stores
    const int grpB = H * 16;
    const int qend4 = (qend + 3) & ~3;  // the row's last group may be partial: its columns past W are padding and hold 0
    while (qcur < qend4) {
        // entries [idx, idx + kRE) of every row go to LDS as (first pixel, float(column), addend); the round ends where the
        // first row would need entry idx + kRE -- and takes 4 pixels at least: an entry takes over at most once per pixel,
        // so 4 pixels need 5 staged entries at most
        unsigned pk[kRE];
        float bb[kRE];
#pragma unroll
        for (int e = 0; e < kRE; ++e) {
            const OwnEntry oe = own[min(idx + e, lc - 1)];
            pk[e] = oe.pk; bb[e] = oe.b;
        }
#pragma unroll
        for (int e = 0; e < kRE; ++e) {
            if (idx + e >= lc) pk[e] = 0x7fff0000u;  // past the list: never taken over
            f_st[e][tid] = (int)(pk[e] >> 16); f_vf[e][tid] = (float)(pk[e] & 0xffffu); f_b[e][tid] = bb[e];
        }
        const int lim = (int)(pk[kRE - 1] >> 16);
        const int qstop = max(qcur + 4, min(qend4, __builtin_amdgcn_readfirstlane(wave_min(lim))) & ~3);
        float cvf = (float)(pk[0] & 0xffffu), cb = bb[0];
        int a = 0;
        for (int q0 = qcur; q0 < qstop; q0 += 4) {
            float g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // the next entry takes over at its first pixel (a + 1 <= kRE - 1: see above)
                const int nst = f_st[a + 1][tid];
                const float nvf = f_vf[a + 1][tid], nb = f_b[a + 1][tid];
                const bool adv = q0 + j >= nst;
                cvf = adv ? nvf : cvf; cb = adv ? nb : cb; a += adv ? 1 : 0;
                const float dq = (float)(q0 + j) - cvf;
                // addend + (q - v)^2: integers below 2^24, so the fused form rounds nothing either (imgproc.h:127)
                g[j] = q0 + j < W ? __builtin_fmaf(dq, dq, cb) : 0.f;
            }
            u32x4 out;
            out.x = __float_as_uint(g[0]); out.y = __float_as_uint(g[1]); out.z = __float_as_uint(g[2]); out.w = __float_as_uint(g[3]);
            // (the whole offset in the lane offset, none in the scalar operand: a 16-byte store reads its data late, and the
            // compiler only inserts the wait state before the registers are overwritten when there is no scalar offset)
            __builtin_amdgcn_raw_buffer_store_b128(out, rs, vrow + (unsigned)((q0 >> 2) * grpB), 0, 0);
        }
        idx += a;
        qcur = qstop;
    }
}

__global__ void __launch_bounds__(kNT) k_sweep_balanced(const ColDesc* __restrict__ desc, float* __restrict__ vol, int W, int H, int HW64, int part_w,
                                                        SweepBuf B) {
    // one LDS pool for the phases: the construction's rings (kept through the merge), the walk's lists, the fill's staging
    constexpr size_t kPoolBytes = std::max({(size_t)3 * kRing * kNT * sizeof(float), (size_t)3 * 64 * kWinStride * 4, (size_t)3 * kRE * kNT * 4});
    __shared__ SweepLds L;
    __shared__ float4 pool[kPoolBytes / sizeof(float4)];
    const long long t_start = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long chunk = B.order ? B.order[blockIdx.x] : (long)blockIdx.x;
#ifdef FDCM_LAB
    long long* lab = B.lab ? B.lab + ((size_t)chunk * kSeg + wave) * kLabN : nullptr;
#define LAB_STAMP(i) do { if (lab && lane == 0) lab[i] = wall_clock64(); } while (0)
#else
#define LAB_STAMP(i) do { } while (0)
#endif
    LAB_STAMP(0);
#ifdef FDCM_LAB
    if (lab && lane == 0) {  // where the workgroup runs: HW_ID (CU, SH, SE) and the XCC
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        lab[22] = hw; lab[23] = xcc;
    }
#endif
    const long k = chunk / HW64;
    const int c = (int)(chunk - k * HW64);
    const int y = c * 64 + lane;
    const size_t r = (size_t)chunk * 64 + lane;
    const uint4* dp = reinterpret_cast<const uint4*>(desc + ((size_t)k * HW64 + c) * W);
    const int nwords = (W + 63) >> 6;
    for (int b = tid; b < nwords; b += kNT) L.smask[b] = B.colmask[(size_t)k * nwords + b];
    __syncthreads();
    int n = 0;
    for (int b = 0; b < nwords; ++b) n += __popcll(uni64(L.smask[b]));
    if (n == 0) {
        // no seed in the slice: column 0 owns every pixel with f = FLT_MAX, and FLT_MAX + d^2 == FLT_MAX (imgproc.h:127)
        const size_t sl = ivol_slice_floats(W, H);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(vol + (size_t)k * sl, 0, (unsigned)(sl * 4), 0x00020000);
        const unsigned vrow = y < H ? (unsigned)y * 16u : 0x80000000u;
        const int ngroups = (W + 3) >> 2;
        for (int g = wave; g < ngroups; g += kSeg) {
            u32x4 out;
            out.x = __float_as_uint(FLT_MAX);
            out.y = 4 * g + 1 < W ? __float_as_uint(FLT_MAX) : 0u;
            out.z = 4 * g + 2 < W ? __float_as_uint(FLT_MAX) : 0u;
            out.w = 4 * g + 3 < W ? __float_as_uint(FLT_MAX) : 0u;
            __builtin_amdgcn_raw_buffer_store_b128(out, rs, vrow + (unsigned)(g * (H * 16)), 0, 0);
        }
        if (tid == 0) B.cost[chunk] = (int)(wall_clock64() - t_start);
        return;
    }
    // The kernel lasts as long as its heaviest workgroups (their chains, not the chip's throughput: DESIGN.md section 4), and a
    // workgroup shares its CU's SIMDs with a lighter one: the heavy ones get the issue slots first.  Rank in the launch order
    // when there is one (longest chunks of the previous build first), the slice's seeded columns otherwise.  (Config 2: 0.190 -
    // 0.193 ms against 0.196 - 0.198 without, three alternating pairs on one box; config 3: no difference.)
    int heavy;
    {
        heavy = B.order ? (blockIdx.x < 96u ? 3 : blockIdx.x < 224u ? 2 : blockIdx.x < 512u ? 1 : 0) : min(3, 4 * n / max(W, 1));
        if (heavy == 3) __builtin_amdgcn_s_setprio(3);
        else if (heavy == 2) __builtin_amdgcn_s_setprio(2);
        else if (heavy == 1) __builtin_amdgcn_s_setprio(1);
    }
    const int S0 = sweep_ranges(n, B.min_cols);  // ranges this slice starts with (one wave each)
    const Ring ring{reinterpret_cast<float*>(pool)};
    // ---- claim blocks
    // kpr blocks per initial range, of n / nblk columns each (+- 1): ~8 columns where the slice has them, single columns on small
    // slices, kMaxBlk blocks in all at most; block b begins at the column of rank floor(n b / nblk), so that the
    // initial ranges hold equal column counts
    const int per = n / S0;
    const int kpr = max(1, min(kMaxBlk / kSeg, per >= 16 ? (per + 4) / 8 : per));
    const int nblk = S0 * kpr;
    for (int bq = tid; bq <= nblk; bq += kNT) L.blk_q[bq] = (unsigned short)(bq < nblk ? select_column_lane(L.smask, (int)(((long)n * bq) / nblk)) : W);
    if (tid < kMaxR) { L.s_slot0[tid] = 0x7fffffff; L.r_pos[tid] = tid; }
    if (tid == 0) {
        unsigned long long c0 = 0ull, c1 = 0ull;
        for (int w = 0; w < S0; ++w) { const int bw = w * kpr; if (bw < 64) c0 |= 1ull << bw; else c1 |= 1ull << (bw - 64); }
        L.claim[0] = c0; L.claim[1] = c1;
        L.n_ranges = S0;
    }
    __syncthreads();
    {
        // Wave w < S0 begins with range w at block nblk w / S0 (equal column counts, as far as blocks allow).  A wave without
        // columns (its stretch ran into the next range, or it never had one) looks for the longest stretch of blocks nobody
        // has started and begins a new range in its middle, as long as that stretch is worth a stack of its own.
        // (where workgroups queue for the CUs or share them with other frames, only the heaviest ones cut dynamically: theirs are
        // the chains the kernel ends with, and the junctions a new range adds are work the others would only pay for)
        const bool dyn = B.steal_min > 0 && (B.steal_heavy_only == 0 || heavy >= 3);
        EnvEntry* ent_row = B.ent + r * (size_t)B.eslots;
        int rid = wave, bstart = wave * kpr;
        bool have = false, run_now = wave < S0;  // have: this wave's ring columns hold a range's top entries
        int cnt = 0;
        for (;;) {
            if (run_now) {
                run_now = false;
                have = true;
                int base, q0;
#ifdef FDCM_LAB
                local_run(dp, W, L, bstart, nblk, dyn ? -1 : (wave + 1) * kpr, lane, y, tid, ring, ent_row, cnt, base, q0, lab);
#else
                local_run(dp, W, L, bstart, nblk, dyn ? -1 : (wave + 1) * kpr, lane, y, tid, ring, ent_row, cnt, base, q0, nullptr);
#endif
                L.t_cnt[rid][lane] = (short)cnt; L.t_base[rid][lane] = (short)base;
                if (lane == 0) { L.s_slot0[rid] = q0; L.r_wave[rid] = wave; }
#ifdef FDCM_LAB
                if (lab) { const int mc = wave_max(cnt); if (lane == 0) { lab[8] += 1; lab[10] = max((int)lab[10], mc); } }
#endif
            }
            if (!dyn) break;
            // the longest stretch of unclaimed blocks (wave-uniform scalar scan; the map may change under it: the claim decides)
            const unsigned long long f0 = ~uni64(L.claim[0]), f1 = ~uni64(L.claim[1]);
            int best = 0, best_at = 0, run = 0;
            for (int bb = 0; bb < nblk; ++bb) {
                const bool fr = (((bb < 64 ? f0 : f1) >> (bb & 63)) & 1ull) != 0ull;
                run = fr ? run + 1 : 0;
                if (run > best) { best = run; best_at = bb - run + 1; }
            }
            if (best < B.steal_min || best * (n / nblk) < B.steal_cols) break;
            int id = 0;
            if (lane == 0) id = atomicAdd(&L.n_ranges, 1);
            id = __builtin_amdgcn_readfirstlane(id);
            if (id >= kMaxR) break;                       // (the id is lost: n_ranges only says how many were handed out)
            const int bs = best_at + best / 2;            // the far half: whoever runs towards it from the left keeps the near one
            if (!claim_block(L, bs, lane)) continue;      // somebody else got there first (the id stays without a range): look again
            if (lane == 0 && B.steals) atomicAdd(B.steals, 1);
            if (have) L.t_base[rid][lane] = (short)cnt;   // this wave's ring columns get a new tenant: the old range is read from memory from now on
            rid = id; bstart = bs; run_now = true;
        }
    }
    LAB_STAMP(1);
    __syncthreads();  // every range's stack is in memory, its top entries in the rings
    LAB_STAMP(2);
    // ---- the ranges in column order: ids beyond the first S0 were handed out as waves ran dry.  The tables are rewritten by
    // rank (the merge and the walk take the ranges from left to right) -- only when a range was taken over at all.
    int S = S0;
    if (__builtin_amdgcn_readfirstlane(L.n_ranges) > S0) {
        const int nid = min(__builtin_amdgcn_readfirstlane(L.n_ranges), kMaxR);
        if (tid < kMaxR) {
            const int mine = tid < nid ? L.s_slot0[tid] : 0x7fffffff;
            int pos = 0;
            for (int j = 0; j < nid; ++j) pos += L.s_slot0[j] < mine ? 1 : 0;
            L.r_pos[tid] = mine == 0x7fffffff ? -1 : pos;
        }
        __syncthreads();
        S = 0;
        for (int j = 0; j < nid; ++j) S += L.r_pos[j] >= 0 ? 1 : 0;
        S = __builtin_amdgcn_readfirstlane(S);
        short pc[2], pb[2];
        int pp[2], ps[2], pw[2];
        for (int e = 0; e < 2; ++e) {   // kMaxR * 64 = 2 kNT elements
            const int i = tid + e * kNT, id = i >> 6, row = i & 63;
            pp[e] = L.r_pos[id]; pc[e] = L.t_cnt[id][row]; pb[e] = L.t_base[id][row]; ps[e] = L.s_slot0[id]; pw[e] = L.r_wave[id];
        }
        __syncthreads();
        for (int e = 0; e < 2; ++e) {
            const int i = tid + e * kNT, row = i & 63;
            if (pp[e] >= 0) {
                L.t_cnt[pp[e]][row] = pc[e]; L.t_base[pp[e]][row] = pb[e];
                if (row == 0) { L.s_slot0[pp[e]] = ps[e]; L.r_wave[pp[e]] = pw[e]; }
            }
        }
        __syncthreads();
    }
    const int g8 = lane >> 3, t8 = lane & 7, sh8 = lane & 56, row8 = wave * 8 + g8;
    {
        EnvEntry* entr = B.ent + ((size_t)chunk * 64 + row8) * (size_t)B.eslots;
#ifdef FDCM_LAB
        merge_bulk(L, S, row8, t8, sh8, ring, entr, lab);
#else
        merge_bulk(L, S, row8, t8, sh8, ring, entr, nullptr);
#endif
    }
    LAB_STAMP(3);
    __syncthreads();  // every wave is through with the rings: their LDS becomes the walk's lists
    LAB_STAMP(4);
    {
        unsigned* w32 = reinterpret_cast<unsigned*>(pool);
        if (S <= kSeg)
            walk_batched<kSeg>(L, W, S, part_w, B, chunk, row8, t8, sh8, reinterpret_cast<unsigned(*)[kWinStride]>(w32),
                               reinterpret_cast<float(*)[kWinStride]>(w32 + 64 * kWinStride), reinterpret_cast<int(*)[kWinStride]>(w32 + 2 * 64 * kWinStride));
        else
            walk_batched<kMaxR>(L, W, S, part_w, B, chunk, row8, t8, sh8, reinterpret_cast<unsigned(*)[kWinStride]>(w32),
                                reinterpret_cast<float(*)[kWinStride]>(w32 + 64 * kWinStride), reinterpret_cast<int(*)[kWinStride]>(w32 + 2 * 64 * kWinStride));
    }
    LAB_STAMP(5);
    __syncthreads();  // the chunk's owner lists are in memory
    LAB_STAMP(6);
    if (tid == 0) B.cost[chunk] = (int)(wall_clock64() - t_start);
    {
        unsigned* w32 = reinterpret_cast<unsigned*>(pool);
        fill_part(L, vol, W, H, k, c, chunk, part_w, B, wave, reinterpret_cast<int(*)[kNT]>(w32), reinterpret_cast<float(*)[kNT]>(w32 + kRE * kNT),
                  reinterpret_cast<float(*)[kNT]>(w32 + 2 * kRE * kNT));
    }
    LAB_STAMP(7);
#ifdef FDCM_LAB
    if (lab && lane == 0) { lab[11] = L.s_lcount[wave * (64 / kSeg)]; }
#endif
#undef LAB_STAMP
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

int sweep_min_cols() {
    static const int v = [] { const char* e = getenv("FDCM_SWEEP_MINCOLS"); const int x = e ? atoi(e) : 0; return (x >= 1 && x <= 64) ? x : kMinCols; }();
    return v;
}

void launch_sweep_order(hipStream_t st, const int* cost, int n, int* order) { hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, st, cost, n, order); }

void launch_sweep_balanced(hipStream_t st, const void* desc, float* vol, int W, int H, int HW64, long nchunks, const SweepBuf& B_) {
    const int part_w = (((W + kSeg - 1) / kSeg) + 3) & ~3;  // fill parts start on a group of 4 columns
    SweepBuf B = B_;
    B.min_cols = sweep_min_cols();
    // FDCM_SWEEP_STEAL=<blocks> (the tests' switch): a wave out of columns begins a new range in an unclaimed stretch of at least
    // that many blocks (of 8 columns, fewer on small slices); 0 = never (ranges of equal count only); default kStealMin
    static const int env_steal = [] { const char* e = getenv("FDCM_SWEEP_STEAL"); const int x = (e && *e) ? atoi(e) : -1; return (x >= 0 && x <= kMaxBlk) ? x : -1; }();
    B.steal_min = env_steal >= 0 ? env_steal : (B.steal_min < 0 ? kStealMin : B.steal_min);
    static const int env_heavy = [] { const char* e = getenv("FDCM_SWEEP_STEAL_HEAVY"); return (e && *e) ? atoi(e) : -1; }();  // measurement: 0 all workgroups, 1 the heaviest only
    if (env_heavy >= 0) B.steal_heavy_only = env_heavy;
    else if (env_steal >= 0) B.steal_heavy_only = 0;
    B.steal_cols = env_steal >= 0 ? 0 : kStealCols;  // (the forced threshold counts blocks only: small test images have blocks of one column)
#ifdef FDCM_LAB
    if (getenv("FDCM_SWEEP_LAB")) {  // per-wave phase times (100 MHz clock) and counters of this launch, on stderr
        static DevBuf labbuf;
        const size_t nl = (size_t)nchunks * kSeg * kLabN;
        labbuf.reserve(nl * 8);
        FDCM_HIP(hipMemsetAsync(labbuf.p, 0, nl * 8, st));
        SweepBuf B2 = B;
        B2.lab = labbuf.as<long long>();
        // FDCM_SWEEP_LAB=4: unused dynamic LDS on top, so that one workgroup has a CU to itself (what do co-resident waves cost?)
        const size_t pad = atoi(getenv("FDCM_SWEEP_LAB")) == 4 ? 70000 : 0;
        if (pad) FDCM_HIP(hipFuncSetAttribute((const void*)k_sweep_balanced, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad));
        hipLaunchKernelGGL(k_sweep_balanced, dim3((unsigned)nchunks), dim3(kNT), pad, st, (const ColDesc*)desc, vol, W, H, HW64, part_w, B2);
        FDCM_HIP(hipStreamSynchronize(st));
        std::vector<long long> d(nl);
        FDCM_HIP(hipMemcpy(d.data(), labbuf.p, nl * 8, hipMemcpyDeviceToHost));
        long long t0 = 0x7fffffffffffffffll, t1 = 0;
        for (long ch = 0; ch < nchunks; ++ch) for (int w = 0; w < kSeg; ++w) { const long long* e = &d[((size_t)ch * kSeg + w) * kLabN]; if (e[0]) { t0 = std::min(t0, e[0]); t1 = std::max(t1, e[7]); } }
        auto pct = [](std::vector<double>& v, double q) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
        const char* names[7] = {"local run", "wait 1", "merge", "wait 2", "owner walk", "wait 3", "fill"};
        fprintf(stderr, "[sweep lab] %ld chunks, kernel span %.1f us (first stamp to last)\n", nchunks, (t1 - t0) / 100.0);
        for (int ph = 0; ph < 7; ++ph) {
            std::vector<double> v;
            for (long ch = 0; ch < nchunks; ++ch) for (int w = 0; w < kSeg; ++w) {
                const long long* e = &d[((size_t)ch * kSeg + w) * kLabN];
                if (!e[0]) continue;
                v.push_back((e[ph + 1] - e[ph]) / 100.0);
            }
            double sum = 0; for (double x : v) sum += x;
            fprintf(stderr, "[sweep lab] %-16s us per wave: mean %7.1f  p50 %7.1f  p90 %7.1f  p99 %7.1f  max %7.1f\n", names[ph], v.empty() ? 0.0 : sum / v.size(), pct(v, .5), pct(v, .9), pct(v, .99), pct(v, 1.0));
        }
        {
            std::vector<double> life, start, cols, it, hb, lst;
            for (long ch = 0; ch < nchunks; ++ch) {
                const long long* e = &d[(size_t)ch * kSeg * kLabN];
                if (!e[0]) continue;
                long long end = 0;
                for (int w = 0; w < kSeg; ++w) end = std::max(end, d[((size_t)ch * kSeg + w) * kLabN + 7]);
                life.push_back((end - e[0]) / 100.0); start.push_back((e[0] - t0) / 100.0);
                it.push_back((double)e[12]); hb.push_back((double)e[13]);
                for (int w = 0; w < kSeg; ++w) { cols.push_back((double)d[((size_t)ch * kSeg + w) * kLabN + 9]); lst.push_back((double)d[((size_t)ch * kSeg + w) * kLabN + 10]); }
            }
            double ls = 0; for (double x : life) ls += x;
            fprintf(stderr, "[sweep lab] block life us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f (sum %.0f); block start us: p50 %.1f p90 %.1f max %.1f\n", ls / life.size(), pct(life, .5), pct(life, .9),
                    pct(life, .99), pct(life, 1.0), ls, pct(start, .5), pct(start, .9), pct(start, 1.0));
            {  // the blocks that end last: where their time went (per phase: the longest wave)
                std::vector<std::pair<double, long>> ends;
                for (long ch = 0; ch < nchunks; ++ch) {
                    long long end = 0;
                    for (int w = 0; w < kSeg; ++w) end = std::max(end, d[((size_t)ch * kSeg + w) * kLabN + 7]);
                    if (d[(size_t)ch * kSeg * kLabN]) ends.push_back({(end - t0) / 100.0, ch});
                }
                std::sort(ends.begin(), ends.end());
                // .. and the blocks that live longest (second pass of the loop below)
                std::vector<std::pair<double, long>> lives;
                for (auto& en : ends) lives.push_back({en.first - (d[(size_t)en.second * kSeg * kLabN] - t0) / 100.0, en.second});
                std::sort(lives.begin(), lives.end());
                for (int pass = 0; pass < 2; ++pass)
                for (size_t i = ends.size() > 6 ? ends.size() - 6 : 0; i < ends.size(); ++i) {
                    const long ch = pass == 0 ? ends[i].second : lives[i].second;
                    const char* what = pass == 0 ? "late" : "long";
                    double ph[7] = {0, 0, 0, 0, 0, 0, 0}, st0 = 1e30;
                    long long cols = 0, depth = 0, iters = 0, hbm = 0, refill = 0, owners = 0;
                    for (int w = 0; w < kSeg; ++w) {
                        const long long* e = &d[((size_t)ch * kSeg + w) * kLabN];
                        st0 = std::min(st0, (e[0] - t0) / 100.0);
                        for (int q = 0; q < 7; ++q) ph[q] = std::max(ph[q], (e[q + 1] - e[q]) / 100.0);
                        cols = std::max(cols, e[9]); depth = std::max(depth, e[10]); iters = std::max(iters, e[12]); hbm = std::max(hbm, e[13]); refill = std::max(refill, e[14]);
                        owners = std::max(owners, e[11]);
                    }
                    double endt = 0;
                    for (auto& en : ends) if (en.second == ch) endt = en.first;
                    fprintf(stderr, "[sweep lab] %s block %ld (slice %ld chunk %ld): %.1f -> %.1f us | local %.1f merge %.1f walk %.1f fill %.1f | columns %lld deepest %lld merge steps %lld (hbm %lld, refills %lld) owners %lld\n",
                            what, ch, ch / HW64, ch % HW64, st0, endt, ph[0], ph[2], ph[4], ph[6], cols, depth, iters, hbm, refill, owners);
                }
            }
            {
                std::vector<double> a, b, c;
                for (long ch = 0; ch < nchunks; ++ch) for (int w = 0; w < kSeg; ++w) { const long long v = d[((size_t)ch * kSeg + w) * kLabN + 15]; if (!d[((size_t)ch * kSeg + w) * kLabN]) continue; a.push_back((double)(v >> 40) / 100.0); b.push_back((double)((v >> 20) & 0xfffff) / 100.0); c.push_back((double)(v & 0xfffff) / 100.0); }
                fprintf(stderr, "[sweep lab] merge: incoming entries arrive after p50 %.1f p90 %.1f us; junction loops p50 %.1f p90 %.1f us; first step of the first junction p50 %.2f p90 %.2f us\n", pct(a, .5), pct(a, .9), pct(b, .5), pct(b, .9), pct(c, .5), pct(c, .9));
            }
            {
                std::vector<double> np, ns, nf, cp, cs, ct;
                double snp = 0, scp = 0, sns = 0, scs = 0, sct = 0;
                long long hp = 0, hcp = 0, hs = 0, hcs = 0, hct = 0, hf = 0;
                for (long ch = 0; ch < nchunks; ++ch) for (int w = 0; w < kSeg; ++w) {
                    const long long* e = &d[((size_t)ch * kSeg + w) * kLabN];
                    if (!e[0] || !e[16]) continue;
                    np.push_back((double)e[16]); ns.push_back((double)e[17]); nf.push_back((double)e[18]);
                    cp.push_back((double)e[19] / (double)e[16]); cs.push_back(e[17] ? (double)e[20] / (double)e[17] : 0.0); ct.push_back((double)e[21]);
                    snp += (double)e[16]; scp += (double)e[19]; sns += (double)e[17]; scs += (double)e[20]; sct += (double)e[21];
                    if (e[21] > hct) { hct = e[21]; hp = e[16]; hcp = e[19]; hs = e[17]; hcs = e[20]; hf = e[18]; }
                }
                {  // the shared-cursor run: columns, ticks inside the pop loop statements, ticks of the whole local run
                    std::vector<double> pc, oc, cols;
                    double sp = 0, st = 0, sc = 0;
                    for (long ch = 0; ch < nchunks; ++ch) for (int w = 0; w < kSeg; ++w) {
                        const long long* e = &d[((size_t)ch * kSeg + w) * kLabN];
                        if (!e[0] || e[16] || !e[17]) continue;
                        pc.push_back((double)e[19] / (double)e[17]); oc.push_back((double)(e[21] - e[19]) / (double)e[17]); cols.push_back((double)e[17]);
                        sp += (double)e[19]; st += (double)e[21]; sc += (double)e[17];
                    }
                    if (!pc.empty())
                        fprintf(stderr, "[sweep lab] shared cursor: s_memtime ticks per column inside the pop loop: mean %.0f p50 %.0f p90 %.0f; outside it (descriptor, column value, push, eviction, the scan): mean %.0f p50 %.0f p90 %.0f; pop loop = %.0f %% of the local runs' ticks; columns per wave p50 %.0f\n",
                                sp / sc, pct(pc, .5), pct(pc, .9), (st - sp) / sc, pct(oc, .5), pct(oc, .9), 100.0 * sp / st, pct(cols, .5));
                }
                if (!np.empty())
                    fprintf(stderr, "[sweep lab] lane cursors: passes per wave p50 %.0f p90 %.0f max %.0f; stagings p50 %.0f max %.0f; slow passes p50 %.0f max %.0f; s_memtime ticks per pass: mean %.0f p50 %.0f p90 %.0f; per staging: mean %.0f; share of the local run: passes %.0f %%, stagings %.0f %%; longest wave: %lld ticks = %lld passes (%lld ticks) + %lld stagings (%lld ticks), %lld slow\n",
                            pct(np, .5), pct(np, .9), pct(np, 1.0), pct(ns, .5), pct(ns, 1.0), pct(nf, .5), pct(nf, 1.0), scp / snp, pct(cp, .5), pct(cp, .9), scs / std::max(1.0, sns), 100.0 * scp / sct, 100.0 * scs / sct, hct, hp, hcp, hs, hcs, hf);
            }
            if (atoi(getenv("FDCM_SWEEP_LAB")) == 2) {  // which workgroups share a CU
                std::vector<std::pair<unsigned long long, long>> where;
                for (long ch = 0; ch < nchunks; ++ch) {
                    const long long* e = &d[(size_t)ch * kSeg * kLabN];
                    if (!e[0]) continue;
                    const unsigned hw = (unsigned)e[22], xcc = (unsigned)e[23] & 0xf;
                    const unsigned cu = (hw >> 8) & 0xf, shid = (hw >> 12) & 1, se = (hw >> 13) & 7;
                    where.push_back({((unsigned long long)xcc << 24) | (se << 16) | (shid << 8) | cu, ch});
                }
                std::sort(where.begin(), where.end());
                fprintf(stderr, "[sweep lab] placement (xcc.se.sh.cu: chunks in launch position order):");
                unsigned long long last = ~0ull;
                int shown = 0;
                for (auto& w : where) {
                    if (w.first != last) { if (++shown > 40) break; fprintf(stderr, "\n[sweep lab]   %llu.%llu.%llu.%llu:", w.first >> 24, (w.first >> 16) & 0xff, (w.first >> 8) & 0xff, w.first & 0xff); last = w.first; }
                    long long end = 0;
                    for (int w2 = 0; w2 < kSeg; ++w2) end = std::max(end, d[((size_t)w.second * kSeg + w2) * kLabN + 7]);
                    fprintf(stderr, " %ld(%.0f-%.0f)", w.second, (d[(size_t)w.second * kSeg * kLabN] - t0) / 100.0, (end - t0) / 100.0);
                }
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "[sweep lab] columns per wave p50 %.0f max %.0f; deepest local stack p50 %.0f p99 %.0f max %.0f; merge iterations per block p50 %.0f p90 %.0f max %.0f, with an HBM fetch p50 %.0f p90 %.0f max %.0f\n",
                    pct(cols, .5), pct(cols, 1.0), pct(lst, .5), pct(lst, .99), pct(lst, 1.0), pct(it, .5), pct(it, .9), pct(it, 1.0), pct(hb, .5), pct(hb, .9), pct(hb, 1.0));
        }
        return;
    }
    // FDCM_SWEEP_PAD=<bytes>: unused dynamic LDS on every workgroup, no other change (how does a pipeline of frames react to
    // fewer sweeps per CU?)
    static const int env_pad = getenv("FDCM_SWEEP_PAD") ? atoi(getenv("FDCM_SWEEP_PAD")) : 0;
    if (env_pad > 0) {
        FDCM_HIP(hipFuncSetAttribute((const void*)k_sweep_balanced, hipFuncAttributeMaxDynamicSharedMemorySize, env_pad));
        hipLaunchKernelGGL(k_sweep_balanced, dim3((unsigned)nchunks), dim3(kNT), (size_t)env_pad, st, (const ColDesc*)desc, vol, W, H, HW64, part_w, B);
        return;
    }
#endif
    hipLaunchKernelGGL(k_sweep_balanced, dim3((unsigned)nchunks), dim3(kNT), 0, st, (const ColDesc*)desc, vol, W, H, HW64, part_w, B);
}

}  // namespace fdcm
