// fdcm_tail.hip -- the step after the search in every caller of the reference (README.md:71-72,
// python/src/matching.cpp:291-307): penalize<Default/ExponentialPenalty> + sort_matches + "take the
// best k", on the raw matches while they are still in HBM, so that only k records leave the GPU
// (and, in sharded runs, only k records per rank cross xGMI).
//
// The penalty's denominator per template, max(len, 1e-6) or pow(max(len, 1e-6), tau)
// (defaultpenalty.cpp:37-41, exponentialpenalty.cpp:42-46), is computed on the host with the host's
// libm exactly as the reference does; the device only divides (IEEE, correctly rounded), so the
// penalised scores are the reference's bits.  Order: ascending score, ties in positional order
// (the reference's std::sort leaves ties unspecified): a stable radix sort of (score, position) with rocPRIM.
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>

#include "fdcm_internal.h"

namespace fdcm {

// total order on float bit patterns as unsigned integers (-0 < +0, NaNs at the ends)
__device__ __forceinline__ unsigned ordered_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void k_tail_keys(const fdcm_match* __restrict__ m, long long n, const float* __restrict__ denom, int base,
                            int T, unsigned* __restrict__ keys, unsigned* __restrict__ idx, float* __restrict__ pscore) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = m[i].score;
    if (denom) {
        const int t = m[i].tmpl_idx - base;
        s = (t >= 0 && t < T) ? s / denom[t] : __uint_as_float(0x7fc00000u);  // out of range: NaN, sorts last
    }
    pscore[i] = s;
    keys[i] = ordered_key(s);
    idx[i] = (unsigned)i;
}

__global__ void k_tail_gather(const fdcm_match* __restrict__ m, const unsigned* __restrict__ idx,
                              const float* __restrict__ pscore, long long k, fdcm_match* __restrict__ out) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    fdcm_match r = m[idx[j]];
    r.score = pscore[idx[j]];
    out[j] = r;
}

// Records to the host without a copy command: a kernel writes them into mapped pinned memory (a hipMemcpyAsync
// device -> host behind a stream's kernels completed 8 - 12 ms late now and then on this system; kernels never did).
__global__ void __launch_bounds__(256) k_records_to_host(const uint4* __restrict__ src, long long n16, uint4* __restrict__ dst) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) dst[i] = src[i];
}
void records_to_host(hipStream_t st, const fdcm_match* src_device, int64_t n, fdcm_match* dst_pinned) {
    if (n <= 0) return;
    fdcm_match* dst_dev = nullptr;
    FDCM_HIP(hipHostGetDevicePointer((void**)&dst_dev, dst_pinned, 0));
    const long long n16 = (long long)n * (long long)(sizeof(fdcm_match) / 16);
    hipLaunchKernelGGL(k_records_to_host, dim3((unsigned)std::min<long long>((n16 + 255) / 256, 2048)), dim3(256), 0, st,
                       reinterpret_cast<const uint4*>(src_device), n16, reinterpret_cast<uint4*>(dst_dev));
    FDCM_HIP(hipGetLastError());
}

// n_blocks blocks of (cap + 1) records back to back, the first int64 of a block's last record = its record count:
// the valid records of all blocks, in block order, into out (mapped pinned memory); total behind them at out[n_blocks * cap].
__global__ void __launch_bounds__(256) k_blocks_to_host(const uint4* __restrict__ blocks, int n_blocks, long long cap,
                                                        uint4* __restrict__ out) {
    const int b = blockIdx.y;
    const long long units = 2 * (cap + 1);  // 16-byte units per block
    long long before = 0, mine = 0, total = 0;
    for (int i = 0; i < n_blocks; ++i) {
        long long c = reinterpret_cast<const long long*>(blocks + (long long)i * units + 2 * cap)[0];
        c = c < 0 ? 0 : (c > cap ? cap : c);
        if (i < b) before += c;
        if (i == b) mine = c;
        total += c;
    }
    const uint4* src = blocks + (long long)b * units;
    uint4* dst = out + 2 * before;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < 2 * mine; i += (long long)gridDim.x * 256) dst[i] = src[i];
    if (b == 0 && blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<long long*>(out + 2 * (long long)n_blocks * cap)[0] = total;
}
void blocks_to_host(hipStream_t st, const void* blocks_device, int32_t n_blocks, int64_t cap, fdcm_match** out, int64_t* n_out) {
    *out = result_acquire(((size_t)n_blocks * (size_t)cap + 1) * sizeof(fdcm_match));
    fdcm_match* dst_dev = nullptr;
    FDCM_HIP(hipHostGetDevicePointer((void**)&dst_dev, *out, 0));
    const unsigned gx = (unsigned)std::min<long long>((2 * cap + 255) / 256 + 1, 256);
    hipLaunchKernelGGL(k_blocks_to_host, dim3(gx, (unsigned)n_blocks), dim3(256), 0, st, reinterpret_cast<const uint4*>(blocks_device),
                       (int)n_blocks, (long long)cap, reinterpret_cast<uint4*>(dst_dev));
    FDCM_HIP(hipGetLastError());
    FDCM_HIP(hipStreamSynchronize(st));
    long long total = 0;
    std::memcpy(&total, *out + (size_t)n_blocks * (size_t)cap, sizeof total);
    *n_out = total;
}

// The k best of n device-resident matches, penalised, into out_device (k <= n, both on fm's device); returns when
// the records are complete.
void run_topk_device(fdcm_featuremap* fm, const fdcm_templates* t, const fdcm_match* matches_device, int64_t n, int32_t base,
                     int penalty, float tau, int64_t k, fdcm_match* out_device) {
    FDCM_HIP(hipSetDevice(fm->device));
    if (!fm->stream) FDCM_HIP(hipStreamCreateWithFlags(&fm->stream, hipStreamNonBlocking));
    hipStream_t st = fm->stream;
    if (k <= 0) return;
    if (k > n) throw std::string("k exceeds the number of matches");
    if (n > 0x7fffffffll) throw std::string("more than 2^31 matches are not supported by the device tail");
    // ---- denominators on the host (getTemplateLengths + the penalty's formula), uploaded through pinned staging
    const bool pen = penalty >= 0;
    const size_t a256 = 255;
    const size_t o_den = 0, o_keys = ((size_t)t->T * 4 + a256) & ~a256, o_keys2 = o_keys + (((size_t)n * 4 + a256) & ~a256),
                 o_idx = o_keys2 + (((size_t)n * 4 + a256) & ~a256), o_idx2 = o_idx + (((size_t)n * 4 + a256) & ~a256),
                 o_ps = o_idx2 + (((size_t)n * 4 + a256) & ~a256), o_tmp = o_ps + (((size_t)n * 4 + a256) & ~a256);
    size_t tmp_bytes = 0;
    FDCM_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const unsigned*)nullptr, (unsigned*)nullptr,
                                       (const unsigned*)nullptr, (unsigned*)nullptr, (size_t)n, 0, 32, st));
    fm->s_tail.reserve(o_tmp + tmp_bytes + 256);
    char* d = (char*)fm->s_tail.p;
    if (pen) {
        std::vector<float> len((size_t)t->T);
        if (t->T) {
            if (fdcm_templates_lengths(t, len.data()) != FDCM_OK) throw std::string(fdcm_last_error());
        }
        fm->s_stage.reserve(std::max<size_t>(16, (size_t)t->T * 4));
        float* hd = (float*)fm->s_stage.p;
        for (int64_t i = 0; i < t->T; ++i) {
            const float l = std::max(len[(size_t)i], 1e-6f);
            hd[i] = penalty == FDCM_DEFAULT_PENALTY ? l : std::pow(l, tau);
        }
        if (t->T) FDCM_HIP(hipMemcpyAsync(d + o_den, hd, (size_t)t->T * 4, hipMemcpyHostToDevice, st));
    }
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_tail_keys, dim3(nb), dim3(256), 0, st, matches_device, (long long)n,
                       pen ? (const float*)(d + o_den) : nullptr, (int)base, (int)t->T, (unsigned*)(d + o_keys),
                       (unsigned*)(d + o_idx), (float*)(d + o_ps));
    FDCM_HIP(rocprim::radix_sort_pairs(d + o_tmp, tmp_bytes, (const unsigned*)(d + o_keys), (unsigned*)(d + o_keys2),
                                       (const unsigned*)(d + o_idx), (unsigned*)(d + o_idx2), (size_t)n, 0, 32, st));
    hipLaunchKernelGGL(k_tail_gather, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, st, matches_device,
                       (const unsigned*)(d + o_idx2), (const float*)(d + o_ps), (long long)k, out_device);
    FDCM_HIP(hipGetLastError());
    FDCM_HIP(hipStreamSynchronize(st));
}

void run_topk(fdcm_featuremap* fm, const fdcm_templates* t, const fdcm_match* matches_device, int64_t n, int32_t base,
              int penalty, float tau, int64_t k, fdcm_match** out, int64_t* n_out) {
    *out = nullptr;
    *n_out = 0;
    FDCM_HIP(hipSetDevice(fm->device));
    if (!fm->stream) FDCM_HIP(hipStreamCreateWithFlags(&fm->stream, hipStreamNonBlocking));
    k = std::min<int64_t>(std::max<int64_t>(k, 0), n);
    *out = result_acquire(std::max<size_t>(1, (size_t)k) * sizeof(fdcm_match));
    if (k == 0) return;
    fm->s_tail_out.reserve((size_t)k * sizeof(fdcm_match));
    run_topk_device(fm, t, matches_device, n, base, penalty, tau, k, fm->s_tail_out.as<fdcm_match>());
    records_to_host(fm->stream, fm->s_tail_out.as<fdcm_match>(), k, *out);
    FDCM_HIP(hipStreamSynchronize(fm->stream));
    *n_out = k;
}

}  // namespace fdcm
