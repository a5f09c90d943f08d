// fdcm_sweep.h -- the balanced L2 / L2^2 sweep (fdcm_sweep.hip), as run_build (fdcm_build.hip) sees it.
#pragma once
#include <hip/hip_runtime.h>

namespace fdcm {

struct EnvEntry;
struct OwnEntry;

static constexpr int kSweepSegments = 8;  // column ranges per row = waves per workgroup

// Scratch of the sweep, row-major (every lane streams through its own row's records).
struct SweepBuf {
    EnvEntry* ent;                        // stack entries [row][slot], eslots per row (>= W)
    OwnEntry* own;                        // owner list [row][index], lslots per row (>= W + 2)
    const int* order;                     // launch position -> chunk (longest chunks of the previous build first), or null
    int* cost;                            // per chunk: 100 MHz ticks from the block's start to the end of its owner walk
    int eslots, lslots;
    const unsigned long long* colmask;    // [slice][(W + 63) / 64]: the slice's seeded columns (k_coldesc_tile)
    int min_cols;                         // seeded columns a range holds at least (set by the launcher)
    int steal_min;                        // a wave out of columns starts a new range in a stretch of at least this many unclaimed blocks
                                          // (0: never, ranges of equal count only; < 0: the launcher's default; FDCM_SWEEP_STEAL overrides both)
    int steal_heavy_only;                 // 1: only the heaviest workgroups (launch rank / seeded columns) cut dynamically
    int steal_cols;                       // .. that holds this many columns at least
    int* steals;                          // [0]: ranges taken over so far (all launches of the handle), or null
#ifdef FDCM_LAB
    long long* lab;                       // lab builds (make LAB=1): 16 clock stamps / counters per (chunk, wave), or null
#endif
};

// column ranges per row of a slice with n seeded columns: at most kSweepSegments, each with min_cols columns at least
// (the kernel and fdcm_debug_sweep_ranges use this one function)
__host__ __device__ inline int sweep_ranges(int n, int min_cols) {
    const int s = n / (min_cols > 1 ? min_cols : 1);
    return s < 1 ? 1 : (s > kSweepSegments ? kSweepSegments : s);
}
// seeded columns a range holds at least: 16, or FDCM_SWEEP_MINCOLS = 1..64 (the tests' switch: small images then take all 8 ranges too)
int sweep_min_cols();

// the sweep applies when every value of the pass is an exact integer (see fdcm_sweep.hip)
inline bool sweep_balanced_applies(long W, long H) { return W * W + H * H <= (1L << 24); }

// queues the sweep of nchunks (slice, 64-row chunk) pairs on st; vol receives the transforms in the interleaved layout
void launch_sweep_balanced(hipStream_t st, const void* desc, float* vol, int W, int H, int HW64, long nchunks, const SweepBuf& B);
// launch order of the next sweep: chunks by decreasing cost (one workgroup; order = a permutation of 0 .. n - 1 whatever the costs are)
void launch_sweep_order(hipStream_t st, const int* cost, int n, int* order);

// fdcm_sweep_literal.hip: the reference's pass followed literally, one wave per chunk -- any feature size
size_t sweep_literal_scratch_bytes(int W, long nchunks);
void launch_sweep_literal(hipStream_t st, const void* desc, float* vol, int W, int H, int HW64, long nchunks, void* scratch);

}  // namespace fdcm
