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
// (x1,y1,x2,y2,slice), off = sceneTranslation + translation.  The two reads are split from the subtraction so that a
// caller can have the reads of many lines in flight.
// Two addressing forms.  VolRef::buf32 (volumes below 4 GB, i.e. every BASELINE config except 5): one buffer descriptor
// for the whole volume, the line's slice as a 32-bit element offset (bin * floats per slice, written into L[5 i + 4] by
// the caller) and 24-bit multiplies -- 18 vector instructions per line where 64-bit flat addresses with a 64-bit
// bin * slice product and full 32-bit multiplies (quarter rate) took the time of 43; k_search alone 0.145 -> measured in
// DESIGN.md.  Otherwise: L[5 i + 4] holds the bin and addresses are 64-bit.
struct VolRef {
    const float* vol;
    size_t SL;                    // floats per slice (ivol_slice_floats)
    __amdgpu_buffer_rsrc_t rs;    // the whole volume (buf32 only)
    bool buf32;
};
__device__ __forceinline__ VolRef make_volref(const float* vol, size_t SL, int m, bool buf32) {
    VolRef v;
    v.vol = vol; v.SL = SL; v.buf32 = buf32;
    v.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vol), 0, buf32 ? (unsigned)((size_t)m * SL * 4) : 0u, 0x00020000);
    return v;
}
// what the caller stores in L[5 i + 4] for a line of orientation bin `bin`
__device__ __forceinline__ float line_slice_word(const VolRef& V, int bin) {
    return __int_as_float(V.buf32 ? (int)((unsigned)bin * (unsigned)V.SL) : bin);
}
struct LineReads {
    float a, b;
};
template <bool BUF32>
__device__ __forceinline__ LineReads line_reads(const VolRef& V, const float* L, int i, float offx, float offy, unsigned H) {
    const float* l = L + 5 * i;
    const int x1 = (int)(l[0] + offx), y1 = (int)(l[1] + offy);  // translate then cast<int>()
    const int x2 = (int)(l[2] + offx), y2 = (int)(l[3] + offy);
    LineReads r;
    // the integrated volume is interleaved (ivol_index): element ((x / 4) * H + y) * 4 + x % 4 of the slice
    if (BUF32) {
        const unsigned se = (unsigned)__float_as_int(l[4]);  // the two halves of a wave work on different lines
        // x / 4 < 2^12 and H <= 2^14: the 24-bit multiply is exact (and full rate)
        const unsigned i1 = (((__umul24((unsigned)x1 >> 2, H) + (unsigned)y1) << 2) | ((unsigned)x1 & 3u)) + se;
        const unsigned i2 = (((__umul24((unsigned)x2 >> 2, H) + (unsigned)y2) << 2) | ((unsigned)x2 & 3u)) + se;
        r.a = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(V.rs, i1 << 2, 0, 0));
        r.b = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(V.rs, i2 << 2, 0, 0));
    } else {
        const float* slice = V.vol + (size_t)__float_as_int(l[4]) * V.SL;
        r.a = slice[((unsigned)(x1 >> 2) * H + (unsigned)y1) * 4u + (unsigned)(x1 & 3)];
        r.b = slice[((unsigned)(x2 >> 2) * H + (unsigned)y2) * 4u + (unsigned)(x2 & 3)];
    }
    return r;
}
template <bool BUF32>
__device__ __forceinline__ float line_value(const VolRef& V, const float* L, int i, float offx, float offy, unsigned H) {
    const LineReads r = line_reads<BUF32>(V, L, i, offx, offy, H);
    return f_abs(r.a - r.b);
}

// score_per_line.sum() (dt3cpu.cpp:175): Eigen 3.4.0 redux (Redux.h, LinearVectorizedTraversal,
// Packet4f): p0 = packet(0), p1 = packet(4); blocks of 8: p0 += packet(i), p1 += packet(i+4);
// p0 += p1; optional trailing packet; predux (p0+p2)+(p1+p3); scalar tail in order.
// Lane h = 0 owns p0, lane h = 1 owns p1 of the same translation; the result is valid in h = 0.
template <bool BUF32>
__device__ __forceinline__ float pair_score(const VolRef& V, const float* L, int n, float offx, float offy, unsigned H, int h,
                                            bool active) {
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
                    for (int l = 0; l < 4; ++l) r[ii][l] = line_reads<BUF32>(V, L, b + l, offx, offy, H);
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
            for (int l = 0; l < 4; ++l) acc[l] = line_value<BUF32>(V, L, l, offx, offy, H);
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
                for (int l = 0; l < 4; ++l) acc[l] = acc[l] + line_value<BUF32>(V, L, aligned2 + l, offx, offy, H);
            }
            res = (acc[0] + acc[2]) + (acc[1] + acc[3]);
            for (int idx = aligned; idx < n; ++idx) res = res + line_value<BUF32>(V, L, idx, offx, offy, H);
        } else if (n > 0) {
            res = line_value<BUF32>(V, L, 0, offx, offy, H);
            for (int idx = 1; idx < n; ++idx) res = res + line_value<BUF32>(V, L, idx, offx, offy, H);
        }
    }
    return res;
}

}  // namespace fdcm
