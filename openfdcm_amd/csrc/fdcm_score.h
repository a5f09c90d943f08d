// fdcm_score.h -- evaluate<Dt3Cpu> (dt3cpu.cpp:126-179) on the interleaved volume: the device helpers shared by the
// search kernel (fdcm_search.hip) and the feature-map seam (fdcm_seam.hip).  Device code only.
#pragma once
#include "fdcm_internal.h"

namespace fdcm {

__device__ __forceinline__ float wave_min_f(float v) {
    for (int d = 32; d >= 1; d >>= 1) v = std_min(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
    for (int d = 32; d >= 1; d >>= 1) v = std_max(v, __shfl_xor(v, d));
    return v;
}

// evaluate<Dt3Cpu> for one line and one translation, dt3cpu.cpp:153-173.  L = per-wave LDS lines
// (x1,y1,x2,y2,bin), off = sceneTranslation + translation.  The two reads are split from the subtraction so that a
// caller can have the reads of many lines in flight (SL = floats per slice, ivol_slice_floats).
struct LineReads {
    float a, b;
};
__device__ __forceinline__ LineReads line_reads(const float* __restrict__ vol, const float* L, int i, float offx,
                                                float offy, size_t SL, unsigned H) {
    const float* l = L + 5 * i;
    const int x1 = (int)(l[0] + offx), y1 = (int)(l[1] + offy);  // translate then cast<int>()
    const int x2 = (int)(l[2] + offx), y2 = (int)(l[3] + offy);
    const float* slice = vol + (size_t)__float_as_int(l[4]) * SL;  // the two halves of a wave work on different lines
    // the integrated volume is interleaved (ivol_index)
    LineReads r;
    r.a = slice[((unsigned)(x1 >> 2) * H + (unsigned)y1) * 4u + (unsigned)(x1 & 3)];
    r.b = slice[((unsigned)(x2 >> 2) * H + (unsigned)y2) * 4u + (unsigned)(x2 & 3)];
    return r;
}
__device__ __forceinline__ float line_value(const float* __restrict__ vol, const float* L, int i, float offx,
                                            float offy, size_t SL, size_t H) {
    const LineReads r = line_reads(vol, L, i, offx, offy, SL, (unsigned)H);
    return f_abs(r.a - r.b);
}

// score_per_line.sum() (dt3cpu.cpp:175): Eigen 3.4.0 redux (Redux.h, LinearVectorizedTraversal,
// Packet4f): p0 = packet(0), p1 = packet(4); blocks of 8: p0 += packet(i), p1 += packet(i+4);
// p0 += p1; optional trailing packet; predux (p0+p2)+(p1+p3); scalar tail in order.
// Lane h = 0 owns p0, lane h = 1 owns p1 of the same translation; the result is valid in h = 0.
__device__ __forceinline__ float pair_score(const float* __restrict__ vol, const float* L, int n, float offx,
                                            float offy, size_t W, size_t H, int h, bool active) {
    const int aligned2 = (n / 8) * 8, aligned = (n / 4) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};  // 0 + v == v exactly (v >= +0)
    if (active) {
        if (aligned >= 8) {
            // Four blocks (16 lines per lane, 32 reads) are fetched before the first of them is added: one memory
            // round trip where a block at a time made four.  A short last group repeats its last block's reads and
            // adds 0 for them (acc + 0 == acc exactly).
            const int nblk = aligned2 / 8;
            for (int i0 = 0; i0 < nblk; i0 += 4) {
                LineReads r[4][4];
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int b = 8 * min(i0 + ii, nblk - 1) + 4 * h;
#pragma unroll
                    for (int l = 0; l < 4; ++l) r[ii][l] = line_reads(vol, L, b + l, offx, offy, W, (unsigned)H);
                }
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const bool real = i0 + ii < nblk;
#pragma unroll
                    for (int l = 0; l < 4; ++l) acc[l] = acc[l] + (real ? f_abs(r[ii][l].a - r[ii][l].b) : 0.f);
                }
            }
        } else if (aligned == 4 && h == 0) {
#pragma unroll
            for (int l = 0; l < 4; ++l) acc[l] = line_value(vol, L, l, offx, offy, W, H);
        }
    }
    float res = 0.f;
    if (aligned >= 8) {
#pragma unroll
        for (int l = 0; l < 4; ++l) acc[l] = acc[l] + __shfl_xor(acc[l], 32);  // h = 0: p0 + p1
    }
    if (active && h == 0) {
        if (aligned) {
            if (aligned >= 8 && aligned > aligned2) {
#pragma unroll
                for (int l = 0; l < 4; ++l) acc[l] = acc[l] + line_value(vol, L, aligned2 + l, offx, offy, W, H);
            }
            res = (acc[0] + acc[2]) + (acc[1] + acc[3]);
            for (int idx = aligned; idx < n; ++idx) res = res + line_value(vol, L, idx, offx, offy, W, H);
        } else if (n > 0) {
            res = line_value(vol, L, 0, offx, offy, W, H);
            for (int idx = 1; idx < n; ++idx) res = res + line_value(vol, L, idx, offx, offy, W, H);
        }
    }
    return res;
}

}  // namespace fdcm
