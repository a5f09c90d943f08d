// fdcm_build.hip -- DT3 feature-map build on gfx950 (buildCpuFeaturemap<D>, dt3cpu.h:174-234).
//
// Volume layout in HBM: float vol[k][x][y] (y fastest) -- per slice exactly the reference's
// RawImage<float>(H, W) column-major (math.h:57), so a slice read-back is one memcpy.
//
// Kernels (W x H = feature size, m = slices, V = 4*m*W*H bytes):
//   K0 k_seeds      clipped scene lines -> seed bitmap (1 bit per pixel, bits along y)   ~V/32
//   K1 k_pass1      1-D distance along y from the bitmap (exact integers)               write V
//   K2 k_pass2_l2   literal in-place lower-envelope pass along x (imgproc.h:91-130)      read V, write V
//      k_sweep_l1   L1 min-plus sweeps along x (imgproc.h:137-146)                       read V, write V (x2)
//   K3 k_propagate  orientation propagation, 4m steps per pixel in LDS (+ sqrt for L2)   read V, write V
//   K4 k_integral   directional prefix sum per slice, one sequential chain per thread    read V, write V
// Compiled with -ffp-contract=off; divide and sqrt are the correctly rounded forms.
#include <chrono>
#include <cstring>

#include "fdcm_internal.h"

namespace fdcm {

static constexpr int kWave = 64;

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

template <bool SQUARED>
__global__ void __launch_bounds__(256) k_pass1(const unsigned long long* __restrict__ bitmap, float* __restrict__ vol,
                                               int H, int HW64, long ncols) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long col = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    if (col >= ncols) return;
    const unsigned long long* bw = bitmap + (size_t)col * HW64;
    float* out = vol + (size_t)col * H;
    const int ngroups = (HW64 + 63) >> 6;  // groups of 64 words = 4096 rows
    int carry_prev = INT_MIN;              // last seed row in earlier groups
    for (int g = 0; g < ngroups; ++g) {
        const int wi = g * 64 + lane;
        const unsigned long long word = wi < HW64 ? bw[wi] : 0ull;
        const int last_i = word ? wi * 64 + 63 - __clzll(word) : INT_MIN;
        const int first_i = word ? wi * 64 + (__ffsll((long long)word) - 1) : INT_MAX;
        // first seed row in later groups (rare: only when H > 4096)
        int carry_next = INT_MAX;
        for (int g2 = ngroups - 1; g2 > g; --g2) {
            const int wj = g2 * 64 + lane;
            const unsigned long long w2 = wj < HW64 ? bw[wj] : 0ull;
            carry_next = min(carry_next, wave_min(w2 ? wj * 64 + (__ffsll((long long)w2) - 1) : INT_MAX));
        }
        const int prev_excl = max(wave_scan_max_excl(last_i, lane), carry_prev);
        const int next_excl = min(wave_scan_min_excl_rev(first_i, lane), carry_next);
        const int nchunks = min(64, HW64 - g * 64);
        for (int c = 0; c < nchunks; ++c) {
            const unsigned long long wc = __shfl(word, c);
            const int pc = __shfl(prev_excl, c), nc = __shfl(next_excl, c);
            const int y = (g * 64 + c) * 64 + lane;
            int d = INT_MAX;
            const unsigned long long below = wc & (~0ull >> (63 - lane));  // bits 0..lane
            if (below) d = lane - (63 - __clzll(below));
            else if (pc != INT_MIN) d = y - pc;
            const unsigned long long above = wc >> lane;  // bits lane..63 shifted down
            if (above) d = min(d, __ffsll((long long)above) - 1);
            else if (nc != INT_MAX) d = min(d, nc - y);
            if (y < H) {
                float f = FLT_MAX;
                if (d != INT_MAX) f = SQUARED ? (float)((long)d * (long)d) : (float)d;
                out[y] = f;
            }
        }
        carry_prev = max(carry_prev, wave_max(last_i));
    }
}

// ------------------------------------------------------------------------------------------ K2
// _distanceTransformColumnPassL2 along x (second call, imgproc.h:181-183), one thread per
// (slice, row), followed literally: float intersections s, pop while s <= z[k], and the fill
// that reads the image being overwritten (imgproc.h:122-128).  Lanes of a wave are consecutive
// y, so every access to column x is a coalesced 256-byte segment.  The (v, f[v], z) stack lives
// in HBM scratch, interleaved by thread ([slot][thread]); its top entry is kept in registers.
__global__ void __launch_bounds__(256) k_pass2_l2(float* __restrict__ vol, int W, int H, long nrows,
                                                  int* __restrict__ sv, float* __restrict__ sf,
                                                  float* __restrict__ sz) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nrows) return;
    const long k = gid / H, y = gid - k * H;
    float* row = vol + (size_t)k * W * H + y;  // element x at row[x * H]
    const size_t H_ = (size_t)H, NT = (size_t)nrows;
    const float inf = f_inf();
    // ---- envelope construction (imgproc.h:101-121)
    int tv = 0;
    float tf = row[0], tz = -inf;
    int below = 0;  // entries stored under the register-held top
    for (int q = 1; q < W; ++q) {
        const float fq = row[(size_t)q * H_];
        const float q2 = (float)((long)q * (long)q);
        while (true) {
            const float s = (fq + q2 - tf - (float)((long)tv * (long)tv)) / (float)(2 * (long)q - 2 * (long)tv);
            if (s > tz || below == 0) {  // below == 0 only guards non-finite inputs (z[0] = -inf)
                const size_t slot = (size_t)below * NT + gid;
                sv[slot] = tv; sf[slot] = tf; sz[slot] = tz;
                ++below;
                tv = q; tf = fq; tz = s;
                break;
            }
            --below;
            const size_t slot = (size_t)below * NT + gid;
            tv = sv[slot]; tf = sf[slot]; tz = sz[slot];
        }
    }
    {
        const size_t slot = (size_t)below * NT + gid;
        sv[slot] = tv; sf[slot] = tf; sz[slot] = tz;
    }
    const int n_entries = below + 1;
    // ---- fill (imgproc.h:122-128).  The owner's base value img(v_k) is read from the image in
    // place when v_k lies behind q (already overwritten), else it is the original f[v_k].
    int kk = 0;
    int cv = sv[gid];
    float cf = sf[gid];
    float nz = n_entries > 1 ? sz[NT + gid] : inf;
    bool fresh = true;
    float base = 0.f;
    for (int q = 0; q < W; ++q) {
        while (nz < (float)q) {
            ++kk;
            const size_t slot = (size_t)kk * NT + gid;
            cv = sv[slot]; cf = sf[slot];
            nz = (kk + 1 < n_entries) ? sz[slot + NT] : inf;
            fresh = true;
        }
        if (fresh) {
            base = (cv < q) ? row[(size_t)cv * H_] : cf;
            fresh = false;
        }
        const long dq = (long)q - (long)cv;
        row[(size_t)q * H_] = base + (float)(dq * dq);
    }
}

// _distanceTransformColumnPassL1 along x (imgproc.h:137-146): forward then backward
// col(q) = min(col(q), col(q -+ 1) + 1), one thread per (slice, row).
__global__ void __launch_bounds__(256) k_sweep_l1(float* __restrict__ vol, int W, int H, long nrows) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nrows) return;
    const long k = gid / H, y = gid - k * H;
    float* row = vol + (size_t)k * W * H + y;
    const size_t H_ = (size_t)H;
    float run = row[0];
    for (int q = 1; q < W; ++q) {
        const float c = row[(size_t)q * H_];
        run = std_min(c, run + 1);
        row[(size_t)q * H_] = run;
    }
    for (int q = W - 2; q >= 0; --q) {
        const float c = row[(size_t)q * H_];
        run = std_min(c, run + 1);
        row[(size_t)q * H_] = run;
    }
}

__global__ void k_sqrt(float* __restrict__ vol, size_t n) {  // only for staged (test) builds
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) vol[i] = sqrtf(vol[i]);
}

// ------------------------------------------------------------------------------------------ K3
// propagateOrientation (dt3cpu.cpp:77-107): each pixel's m-vector is loaded once into LDS
// ([slice][thread], conflict free), the 4m steps S[c2] = min(S[c2], S[c1] + w) run there, and it
// is stored once.  L2's final sqrt (imgproc.h:191-192) is applied on load.
__global__ void k_propagate(float* __restrict__ vol, size_t npix, int m, const PropStep* __restrict__ steps,
                            int nsteps, int apply_sqrt) {
    extern __shared__ float S[];
    const int bd = blockDim.x, tid = threadIdx.x;
    const size_t p = (size_t)blockIdx.x * bd + tid;
    const bool ok = p < npix;
    for (int j = 0; j < m; ++j) {
        float v = ok ? vol[(size_t)j * npix + p] : 0.f;
        if (apply_sqrt) v = sqrtf(v);
        S[j * bd + tid] = v;
    }
    for (int s = 0; s < nsteps; ++s) {
        const PropStep st = steps[s];
        const float a = S[st.c2 * bd + tid];
        const float b = S[st.c1 * bd + tid] + st.w;
        S[st.c2 * bd + tid] = std_min(a, b);
    }
    if (ok)
        for (int j = 0; j < m; ++j) vol[(size_t)j * npix + p] = S[j * bd + tid];
}

// ------------------------------------------------------------------------------------------ K4
// lineIntegral (imgproc.h:38-84).  The reference adds the previous (already integrated) line,
// shifted by dy_i = round(i r) - round((i-1) r), into the current one; the shifts telescope, so
// pixel (x_i, c + round(i r)) belongs to chain c and each chain is one sequential float32 sum.
// One thread per chain; the order of additions is the reference's.
__global__ void __launch_bounds__(256) k_integral(float* __restrict__ vol, int W, int H,
                                                  const IntegralDesc* __restrict__ desc) {
    const int k = blockIdx.y;
    const IntegralDesc d = desc[k];
    if (d.mode == 0) return;
    float* img = vol + (size_t)k * W * H;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int steps = d.mode == 1 ? W : H;   // sweep length
    const int span = d.mode == 1 ? H : W;    // extent across chains
    const int last_off = (int)roundf((float)(steps - 1) * d.r);
    const int cmin = -max(0, last_off), cmax = span - 1 - min(0, last_off);
    const int c = cmin + t;
    if (c > cmax) return;
    const int start = d.s < 0 ? steps - 1 : 0;
    float acc = 0.f;
    bool started = false;
    for (int i = 0; i < steps; ++i) {
        const int o = c + (int)roundf((float)i * d.r);
        if (o < 0 || o >= span) {
            if (started) break;  // offsets are monotone: a chain that left never returns
            continue;
        }
        const int a = start + i * d.s;
        const size_t idx = d.mode == 1 ? (size_t)a * H + o : (size_t)o * H + a;
        const float v = img[idx];
        acc = started ? v + acc : v;
        img[idx] = acc;
        started = true;
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
    hipStream_t st = fm->stream;
    fm->W = plan.W; fm->H = plan.H; fm->m = plan.m; fm->tx = plan.tx; fm->ty = plan.ty;
    fm->keys = plan.keys;
    fm->last_build = fdcm_build_timing{};
    if (plan.m == 0 || plan.W == 0) return;
    const int W = (int)plan.W, H = (int)plan.H, m = (int)plan.m;
    if (plan.W > 65536 || plan.H > 65536) throw std::string("feature size above 65536 is not supported");
    const int HW64 = (H + 63) / 64;
    const size_t npix = (size_t)W * H, nvox = npix * m;
    const long nrows = (long)m * H, ncols = (long)m * W;
    fm->vol.reserve(nvox * sizeof(float));
    fm->bitmap.reserve((size_t)ncols * HW64 * 8);
    if (fm->distance != FDCM_L1) fm->stack.reserve((size_t)W * nrows * 12);
    // ---- plan upload: one pinned blob, one async copy
    auto align16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    fm->off_raster = 0;
    fm->off_prop = align16(plan.raster.size() * sizeof(RasterLine));
    fm->off_integral = fm->off_prop + align16(plan.prop.size() * sizeof(PropStep));
    fm->off_keys = fm->off_integral + align16(plan.integral.size() * sizeof(IntegralDesc));
    const size_t blob = fm->off_keys + align16(plan.keys.size() * sizeof(float));
    fm->stage.reserve(blob);
    fm->plan.reserve(blob);
    char* hs = (char*)fm->stage.p;
    if (!plan.raster.empty()) std::memcpy(hs + fm->off_raster, plan.raster.data(), plan.raster.size() * sizeof(RasterLine));
    std::memcpy(hs + fm->off_prop, plan.prop.data(), plan.prop.size() * sizeof(PropStep));
    std::memcpy(hs + fm->off_integral, plan.integral.data(), plan.integral.size() * sizeof(IntegralDesc));
    std::memcpy(hs + fm->off_keys, plan.keys.data(), plan.keys.size() * sizeof(float));
    FDCM_HIP(hipMemcpyAsync(fm->plan.p, hs, blob, hipMemcpyHostToDevice, st));
    fm->n_raster = (int64_t)plan.raster.size();
    fm->n_prop = (int64_t)plan.prop.size();
    const char* dp = (const char*)fm->plan.p;
    const RasterLine* d_raster = (const RasterLine*)(dp + fm->off_raster);
    const PropStep* d_prop = (const PropStep*)(dp + fm->off_prop);
    const IntegralDesc* d_int = (const IntegralDesc*)(dp + fm->off_integral);
    float* vol = fm->vol.as<float>();
    hipEvent_t* ev = fm->timing.ev;

    FDCM_HIP(hipEventRecord(ev[0], st));
    FDCM_HIP(hipMemsetAsync(fm->bitmap.p, 0, (size_t)ncols * HW64 * 8, st));
    if (fm->n_raster > 0)
        hipLaunchKernelGGL(k_seeds, dim3((unsigned)fm->n_raster), dim3(256), 0, st, d_raster,
                           fm->bitmap.as<unsigned long long>(), W, H, HW64);
    FDCM_HIP(hipEventRecord(ev[1], st));
    {
        const unsigned blocks = (unsigned)((ncols + 3) / 4);
        if (fm->distance == FDCM_L1)
            hipLaunchKernelGGL(k_pass1<false>, dim3(blocks), dim3(256), 0, st, fm->bitmap.as<unsigned long long>(), vol,
                               H, HW64, ncols);
        else
            hipLaunchKernelGGL(k_pass1<true>, dim3(blocks), dim3(256), 0, st, fm->bitmap.as<unsigned long long>(), vol,
                               H, HW64, ncols);
    }
    FDCM_HIP(hipEventRecord(ev[2], st));
    {
        const unsigned blocks = (unsigned)((nrows + 255) / 256);
        if (fm->distance == FDCM_L1) {
            hipLaunchKernelGGL(k_sweep_l1, dim3(blocks), dim3(256), 0, st, vol, W, H, nrows);
        } else {
            int* sv = fm->stack.as<int>();
            float* sf = (float*)(sv + (size_t)W * nrows);
            float* sz = sf + (size_t)W * nrows;
            hipLaunchKernelGGL(k_pass2_l2, dim3(blocks), dim3(256), 0, st, vol, W, H, nrows, sv, sf, sz);
        }
    }
    FDCM_HIP(hipEventRecord(ev[3], st));
    const bool want_sqrt = fm->distance == FDCM_L2;
    if (stop_after >= 2) {
        int bd = 256;
        while (bd > 64 && (size_t)m * bd * sizeof(float) > 64 * 1024) bd >>= 1;
        const size_t lds = (size_t)m * bd * sizeof(float);
        if (lds > 64 * 1024)
            FDCM_HIP(hipFuncSetAttribute((const void*)k_propagate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_propagate, dim3((unsigned)((npix + bd - 1) / bd)), dim3(bd), lds, st, vol, npix, m, d_prop,
                           (int)fm->n_prop, want_sqrt ? 1 : 0);
    } else if (want_sqrt) {
        hipLaunchKernelGGL(k_sqrt, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, st, vol, nvox);
    }
    FDCM_HIP(hipEventRecord(ev[4], st));
    if (stop_after >= 3) {
        const int chains = 2 * (W > H ? W : H);
        hipLaunchKernelGGL(k_integral, dim3((unsigned)((chains + 255) / 256), (unsigned)m), dim3(256), 0, st, vol, W, H,
                           d_int);
    }
    FDCM_HIP(hipEventRecord(ev[5], st));
    FDCM_HIP(hipGetLastError());
    FDCM_HIP(hipStreamSynchronize(st));
    const auto t1 = std::chrono::steady_clock::now();
    fdcm_build_timing& bt = fm->last_build;
    bt.total_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
    FDCM_HIP(hipEventElapsedTime(&bt.seeds_ms, ev[0], ev[1]));
    FDCM_HIP(hipEventElapsedTime(&bt.pass1_ms, ev[1], ev[2]));
    FDCM_HIP(hipEventElapsedTime(&bt.pass2_ms, ev[2], ev[3]));
    FDCM_HIP(hipEventElapsedTime(&bt.propagate_ms, ev[3], ev[4]));
    FDCM_HIP(hipEventElapsedTime(&bt.integral_ms, ev[4], ev[5]));
}

}  // namespace fdcm
