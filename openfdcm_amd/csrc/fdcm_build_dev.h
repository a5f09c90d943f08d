// fdcm_build_dev.h -- device helpers shared by the build kernels (fdcm_build.hip, fdcm_sweep.hip).  Device code only.
#pragma once
#include <hip/hip_runtime.h>

#include "fdcm_internal.h"

namespace fdcm {

static constexpr int kWave = 64;

__device__ __forceinline__ int wave_scan_max_excl(int v, int lane) {  // exclusive prefix max
    int incl = v;
    for (int d = 1; d < kWave; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl = max(incl, o);
    }
    const int prev = __shfl_up(incl, 1);
    return lane == 0 ? INT_MIN : prev;
}
__device__ __forceinline__ int wave_scan_min_excl_rev(int v, int lane) {  // exclusive suffix min
    int incl = v;
    for (int d = 1; d < kWave; d <<= 1) {
        const int o = __shfl_down(incl, d);
        if (lane + d < kWave) incl = min(incl, o);
    }
    const int nxt = __shfl_down(incl, 1);
    return lane == kWave - 1 ? INT_MAX : nxt;
}
__device__ __forceinline__ int wave_max(int v) {
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ int wave_min(int v) {
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d));
    return v;
}

// Column-chunk descriptor: for column (k, x) and the 64 rows [64c, 64c+64): the seed bits of the
// chunk, the last seed row before it and the first seed row after it.  16 bytes per 64 pixels,
// stored [k][c][x] so that a wave sweeping along x prefetches 64 columns with one coalesced load.
static constexpr int kFar = 1 << 30;  // "no seed on that side": a row 2^30 away (rows are < 2^14)
struct __attribute__((aligned(16))) ColDesc {
    unsigned long long word;
    int prev;  // -kFar: none
    int next;  // +kFar: none
};
// a column without any seed in the slice (all its chunks say the same)
__device__ __forceinline__ bool desc_seedless(const uint4& d) {
    return (d.x | d.y) == 0u && (int)d.z == -kFar && (int)d.w == kFar;
}

// Pass 1 of distanceTransform (imgproc.h:178 / :186, along y) evaluated on the fly.  On a
// 0 / FLT_MAX image the lower-envelope pass yields exactly the squared distance to the nearest
// seed of the column (every envelope owner is a seed and owns itself), or FLT_MAX for a seedless
// column; the L1 sweeps yield the plain distance.  Both are integers < 2^24, so this bit-scan
// gives the reference's bits.  y = 64c + lane.
template <bool SQUARED>
__device__ __forceinline__ float column_value(unsigned long long wc, int pc, int nc, int lane, int y) {
    // branch-free: a missing neighbour chunk seed is a position 2^30 away (the descriptor says so), so "no seed
    // in the column" is d >= 2^29 (rows are < 2^14)
    const unsigned long long dnw = wc >> lane;         // bit 0 = own row, upwards = rows below it in the image
    const unsigned long long upw = wc << (63 - lane);  // bit 63 = own row
    const int d_dn = dnw ? __ffsll((long long)dnw) - 1 : nc - y;
    const int d_up = upw ? __clzll(upw) : y - pc;
    const int d = min(d_up, d_dn);
    const float df = (float)d;
    return d >= (1 << 29) ? FLT_MAX : (SQUARED ? df * df : df);  // d < 2^14: df * df is the exact integer
}

// The squared value for a SEEDED column whose descriptor fields are wave-uniform (read with v_readlane): most columns of
// a chunk have no seed inside the chunk's 64 rows (a line crosses a chunk in a few columns), and then the value is the
// distance to the neighbour seeds alone -- 5 vector instructions behind a scalar branch instead of ~22.  (A seeded
// column has a seed in some chunk, so a missing side is 2^30 away and the other one decides: never FLT_MAX here.)
__device__ __forceinline__ float column_value_sq_seeded(unsigned long long wc, int pc, int nc, int lane, int y) {
    if (wc == 0ull) {
        const int d = min(y - pc, nc - y);
        const float df = (float)d;
        return df * df;
    }
    return column_value<true>(wc, pc, nc, lane, y);
}

__device__ __forceinline__ void desc_lane(const ColDesc& d, int j, unsigned long long& wc, int& pc, int& nc) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(d.word & 0xffffffffull), j);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(d.word >> 32), j);
    wc = ((unsigned long long)hi << 32) | lo;
    pc = __builtin_amdgcn_readlane(d.prev, j);
    nc = __builtin_amdgcn_readlane(d.next, j);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct EnvEntry { float v2; float P; float z; };     // one stack entry (imgproc.h: v[k], f[v[k]], z[k]) as the sweep's tests use it: 2 v, f[v] + v^2, z
struct OwnEntry { unsigned pk; float b; };          // (first pixel << 16 | column), addend

__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ void desc_lane4(const uint4& d, int j, unsigned long long& wc, int& pc, int& nc) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)d.x, j);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)d.y, j);
    wc = ((unsigned long long)hi << 32) | lo;
    pc = __builtin_amdgcn_readlane((int)d.z, j);
    nc = __builtin_amdgcn_readlane((int)d.w, j);
}

}  // namespace fdcm
