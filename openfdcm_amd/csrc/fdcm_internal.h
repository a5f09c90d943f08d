// fdcm_internal.h -- shared declarations of libfdcm_hip.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fdcm.h"
#include "fdcm_math.h"

namespace fdcm {

// ---------------------------------------------------------------- error plumbing
void set_error(const std::string& msg);
struct HipError { hipError_t code; const char* what; int line; };
#define FDCM_HIP(call)                                                              \
    do {                                                                            \
        hipError_t e__ = (call);                                                    \
        if (e__ != hipSuccess) throw ::fdcm::HipError{e__, #call, __LINE__};        \
    } while (0)

// ---------------------------------------------------------------- device buffers (grow-only)
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    void reserve(size_t bytes);
    void release();
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};
struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    void reserve(size_t bytes);
    void release();
};

// ---------------------------------------------------------------- build plan (host -> device)
// One clipped scene line to rasterise (drawLines, drawing.h:111-125).  Each axis is either a
// constant or Eigen's LinSpaced(n, low, high) (restated in lin_spaced_value below).
struct RasterLine {
    int32_t slice;
    int32_t n;
    float xlow, xhigh, xstep;
    float ylow, yhigh, ystep;
    int32_t xmode, ymode;  // 0 = constant (xlow / ylow), 1 = LinSpaced, 2 = LinSpaced flipped
};
// One step of propagateOrientation (dt3cpu.cpp:86-101): S[c2] = min(S[c2], S[c1] + w).
struct PropStep {
    int32_t c1, c2;
    float w;
    int32_t pad;
};
// lineIntegral of one slice (imgproc.h:38-84): mode 1 sweeps along x (|rastvec.x| == 1), mode 2
// along y (|rastvec.y| == 1), mode 0 does nothing.  s = +-1 sweep direction, r = the other
// rastvec component (chain offset at step i is round(float(i) * r)).
struct IntegralDesc {
    int32_t mode, s;
    float r;
    int32_t pad;
};

struct LineBox { int32_t slice; float xlo, xhi, ylo, yhi; };

struct BuildPlan {
    int64_t W = 0, H = 0, m = 0;
    float tx = 0, ty = 0;
    std::vector<float> keys;
    std::vector<RasterLine> raster;      // ordered by slice
    std::vector<int32_t> slice_first;    // m + 1: the lines of slice k are raster[slice_first[k] .. slice_first[k + 1])
    std::vector<PropStep> prop;
    std::vector<IntegralDesc> integral;
    std::vector<LineBox> boxes;       // the clipped lines' bounding boxes (what sweep_cost_proxy works from)
};

// Host side of buildCpuFeaturemap (dt3cpu.h:174-198 + the scalar parts of :227-231).
void make_build_plan(const float* lines, int64_t n, int64_t depth, float coeff, float padding, BuildPlan& plan);
// per (slice, 64-row chunk): a proxy of the L2 sweep's time, for the launch order of a build without history
void sweep_cost_proxy(const BuildPlan& plan, std::vector<int32_t>& cost);

FDCM_HD float lin_spaced_value(int mode, float low, float high, float step, int n, int i) {
    // Eigen 3.4.0 linspaced_op_impl<float>::operator() (NullaryFunctors.h), scalar path.
    if (mode == 0) return low;
    const int size1 = (n == 1) ? 1 : n - 1;
    if (mode == 2) return (i == 0) ? low : (high - (float)(size1 - i) * step);
    return (i == size1) ? high : (low + (float)i * step);
}

// ---------------------------------------------------------------- the integrated volume
// From the propagation on the volume lives in a layout whose 64-byte sectors hold 4 x by 4 y pixels:
// [k][x/4][y][x%4] (columns of the last group past W are padding).  The search gathers single floats at positions that step by about one pixel per
// translation, in any direction; in the y-fastest layout of the build 16 steps along x touch 16 sectors, here 4 to 8.
// Slices are 4352 bytes longer than their pixels: feature sizes are powers of two in practice, and a candidate's
// gathers read the same place of up to `depth` slices -- with slices a power of two apart they all fall on the same
// memory channels (config 2': the search kernels took 0.24 - 0.36 ms depending on where the allocation landed,
// 0.21 - 0.22 ms with the padding; 256 bytes of padding were not enough, 2 to 20 KB all the same).
#ifndef FDCM_SLICE_PAD
#define FDCM_SLICE_PAD 1088
#endif
FDCM_HD size_t ivol_slice_floats(int64_t W, int64_t H) { return (size_t)((W + 3) / 4) * (size_t)H * 4 + FDCM_SLICE_PAD; }
FDCM_HD size_t ivol_index(int x, int y, int64_t H) { return ((size_t)(x >> 2) * (size_t)H + (size_t)y) * 4 + (size_t)(x & 3); }

// ---------------------------------------------------------------- handles
struct Timing {
    hipEvent_t ev[9] = {};  // 0-5 build stages, 6-7 search kernels, 8 search download
    bool created = false;
};

}  // namespace fdcm

struct fdcm_featuremap {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t prep_stream = nullptr;  // the search's preparation beside a running build (handles that have the GPU to themselves)
    hipEvent_t prep_done = nullptr;
    // parameters
    int64_t depth_param = 0;
    float coeff = 0, padding = 0;
    int distance = 0;
    int* sweep_steals = nullptr;  // device counter inside `stack`: ranges the L2 sweep's waves took over so far (fdcm_selftest_sweep_steals)
    long k2_cost_chunks = 0; int k2_cost_w = 0;  // the L2 sweep's per-chunk costs in `stack` are those of a build with this shape
    int off_m = 0, off_steps = 0;  // the group table in `offtab` is valid for this depth and feature width
    bool build_pending = false;  // the last build is queued on `stream` but has not been waited for
    bool seeds_fused = false;    // the last build drew its seeds inside k_coldesc_tile (no seeds stage, no event for it)
    int want_stage_events = 1;   // fdcm_featuremap_stage_timing: 0 no events, 1 an event between the build's stages, 2 around the build and the search only
    bool stage_events = true;    // the last build recorded its stages (fdcm_build_timing has per-stage times)
    bool total_events = true;    // .. its first and last event (total_ms has the device span)
    bool shares_gpu = false;     // a frame slot of a pipeline with several frames in flight: other frames' kernels run beside this handle's
    float build_host_ms = 0.f;   // host time of that call up to its first kernel launch
    // geometry
    int64_t W = 0, H = 0, m = 0;
    float tx = 0, ty = 0;
    std::vector<float> keys;
    // device state
    // The volume moves between two buffers: the sweeps write the distance transforms into `vol`, y-fastest [k][x][y];
    // the propagation reads them and writes `ivol` in the interleaved layout (ivol_index); the line integral reads
    // `ivol` and writes its sums back into `vol`, interleaved -- which is what the search gathers from.
    fdcm::DevBuf vol;      // max(m*W*H, m*ivol_slice_floats) floats
    fdcm::DevBuf ivol;     // m*ivol_slice_floats floats
    bool vol1_interleaved = false;  // the transforms of the last build (stage 1) are in the interleaved layout: segmented L2 sweep
    int vol_stage = 0;     // what the handle holds: 1 = transforms (vol; staged test builds), 2 = propagated
                           // (ivol, interleaved; staged test builds), 3 = integrated (vol, interleaved): complete
    const float* current() const { return vol_stage == 2 ? ivol.as<float>() : vol.as<float>(); }
    bool current_interleaved() const { return vol_stage >= 2 || (vol_stage == 1 && vol1_interleaved); }
    fdcm::DevBuf bitmap;   // m*W*ceil(H/64) uint64 seed bits along y
    fdcm::DevBuf coldesc;  // m*ceil(H/64)*W column-chunk descriptors (16 B)
    fdcm::DevBuf colmask;  // m*ceil(W/64) words: the seeded columns of every slice (k_coldesc_tile, for the L2 sweep)
    fdcm::DevBuf offtab;   // per slice: one word per group of 4 columns for the shallow sweeps of the line integral (k_groups)
    fdcm::DevBuf stack;    // K2 scratch: per row a (v, f, z) stack of W entries
    fdcm::DevBuf plan;     // RasterLine[] | PropStep[] | IntegralDesc[] | keys[]
    fdcm::PinnedBuf stage; // host staging for the plan
    size_t off_raster = 0, off_prop = 0, off_integral = 0, off_keys = 0, off_slice = 0, off_cost = 0;
    int64_t n_raster = 0, n_prop = 0;
    // search workspaces
    fdcm::DevBuf s_scene;   // scene lines + sorted lengths + sorted idx
    fdcm::DevBuf s_pairs;   // (template line, scene line) per search combination
    fdcm::DevBuf s_records; // per-candidate result records
    fdcm::DevBuf s_flags;   // per-candidate valid flag + scan scratch
    fdcm::DevBuf s_out;     // compacted matches
    fdcm::DevBuf s_work;    // search work list: valid pairs grouped by scene line
    fdcm::DevBuf s_tail;    // device tail (penalise + sort + top k) workspace
    fdcm::DevBuf s_tail_out; // the k best of the device tail before their download
    fdcm::DevBuf s_eval;    // fdcm_featuremap_evaluate / _minmax_translation: lines, translations, work items, results
    std::mutex seam_mutex;  // .. which the reference's optimisers call from pool threads on one feature map (batchoptimize.cpp:102-110):
                            // the two calls share s_eval and the stream, so they take turns
    int64_t last_n_out = 0; // matches of the last host-output search, still in s_out
    fdcm::DevBuf s_counter;
    fdcm::PinnedBuf s_stage;
    fdcm::PinnedBuf s_cnt;   // the search's counters, written by k_scatter (device-output searches)
    fdcm::DevBuf s_bins;     // orientation bins per candidate line from the host libm (only when it differs from the device's atanf)
    fdcm::PinnedBuf s_bins_stage;
    fdcm::Timing timing;
    fdcm_build_timing last_build = {};
    fdcm_search_timing last_search = {};
};

struct fdcm_templates {
    int device = 0;
    int64_t T = 0, n_lines = 0, max_lines = 0;
    std::vector<float> lines;       // host copy
    std::vector<int64_t> offsets;   // T+1
    std::vector<float> lengths;     // per line: getLength, math.h:306-308
    std::vector<int32_t> sorted;    // per template: line indices by descending length (argsort, math.h:106-116)
    fdcm::DevBuf d_lines, d_offsets, d_lengths, d_sorted;
};

namespace fdcm {
// implemented in fdcm_build.hip
// reserve_only: size every buffer a build of this plan takes and queue nothing (the handle keeps its geometry, its plan
// offsets and the sweep's cost history; its content too unless a volume buffer had to grow)
void run_build(fdcm_featuremap* fm, const BuildPlan& plan, int stop_after, bool reserve_only = false);
// Waits for a queued build (if any) and fills fm->last_build.  run_build only queues the kernels: the search
// that follows is ordered behind them on the same stream and its host-side preparation runs meanwhile.
void finish_build(fdcm_featuremap* fm);
void sweep_order_counts(int64_t* from_history, int64_t* from_proxy);
// implemented in fdcm_search.hip
int64_t search_capacity(const fdcm_templates* t, int64_t n_scene, int64_t maxT, int64_t maxS);
void run_search(fdcm_featuremap* fm, const fdcm_templates* t, const float* scene, int64_t n_scene, int64_t maxT,
                int64_t maxS, int optimizer, int64_t batch, int32_t base, fdcm_match* out_device, fdcm_match** out_host,
                int64_t* n_out);
// sizes the search's workspaces for this template set and scene size, queues nothing (run_search reserves the same sizes)
void reserve_search(fdcm_featuremap* fm, const fdcm_templates* t, int64_t n_scene, int64_t maxT, int64_t maxS);
// the searches of this process take the candidates' orientation bins from the host libm (decided once: see fdcm.h)
bool orientation_bins_on_host();
// implemented in fdcm_seam.hip: minmaxTranslation<Dt3Cpu> / evaluate<Dt3Cpu> batched over templates
void run_minmax(fdcm_featuremap* fm, const float* lines, const int64_t* offsets, int64_t T, const float* align, float* out);
void run_evaluate(fdcm_featuremap* fm, const float* lines, const int64_t* offsets, int64_t T, const float* translations,
                  const int64_t* tr_offsets, float* scores);
// implemented in fdcm_tail.hip
void run_topk(fdcm_featuremap* fm, const fdcm_templates* t, const fdcm_match* matches_device, int64_t n, int32_t base,
              int penalty, float tau, int64_t k, fdcm_match** out, int64_t* n_out);
void run_topk_device(fdcm_featuremap* fm, const fdcm_templates* t, const fdcm_match* matches_device, int64_t n, int32_t base,
                     int penalty, float tau, int64_t k, fdcm_match* out_device);
// records to the host without a copy command (a kernel writes into the mapped pinned destination); queued on st
void records_to_host(hipStream_t st, const fdcm_match* src_device, int64_t n, fdcm_match* dst_pinned);
// the valid records of n_blocks fixed-capacity blocks (count in the trailing record) into a pooled pinned array; waits for st
void blocks_to_host(hipStream_t st, const void* blocks_device, int32_t n_blocks, int64_t cap, fdcm_match** out, int64_t* n_out);
// the device tail's total order on float bit patterns (-0 < +0, NaNs at the ends), for host-side merges
inline unsigned ordered_key_host(float f) {
    unsigned u;
    __builtin_memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// the reference's line files (fdcm_lineio.cpp); lines_read hands out malloc'ed memory
void lines_read(const char* path, float** out, int64_t* n_out);
void lines_write(const char* path, const float* lines, int64_t n);
// compute units of a device (cached; fdcm_capi.cpp)
int device_cus(int device);
// pooled pinned host buffers for match arrays returned to the caller (fdcm_host.cpp)
fdcm_match* result_acquire(size_t bytes);
void result_release(fdcm_match* m);
#ifdef FDCM_LAB
// lab build: FDCM_LAB_SKIP=sweep,search,.. leaves the named kernels out of every frame (what does each cost the pipeline?)
inline bool lab_skip(const char* what) {
    const char* e = getenv("FDCM_LAB_SKIP");
    return e && std::strstr(e, what) != nullptr;
}
#endif
}  // namespace fdcm
