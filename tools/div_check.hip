// div_check.hip -- exhaustive check (diagnostic, not product code) that the short division of the L2 sweep's envelope test
// (fdcm_build.hip: envelope_quotient) returns the IEEE quotient for every operand pair the sweep can produce:
//   D = 2 (q - v): every even integer in [2, 2 * 16383]   (columns are < 2^14)
//   N = (f_q + q^2) - f_v - v^2: an integer-valued float with |N| < 2^32 (sums and differences of squared distances and
//       squared columns, each < 2^29, rounded to float at every step -- so above 2^24 only the representable integers),
//       or -FLT_MAX (over the seedless bottom column of a row's first segment: f = FLT_MAX absorbs the other terms)
// This program compares the two for all of them (2.75e12 pairs, seconds); `div_check selftest` runs the same comparison on
// the uncorrected quotient, which must fail.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/div_check.hip -o div_check && ./div_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../openfdcm_amd/csrc/fdcm_quotient.h"

// every integer-valued float in [0, 2^32): index i -> value.  Below 2^24 the integers themselves; binade [2^e, 2^(e+1))
// for e >= 24 has 2^23 floats.
static constexpr unsigned long long kSmall = 1ull << 24, kPerBinade = 1ull << 23, kCount = kSmall + 8 * kPerBinade;
__device__ __forceinline__ float nth_integer_float(unsigned long long i) {
    if (i < kSmall) return (float)(unsigned)i;
    const unsigned long long j = i - kSmall, e = 24 + j / kPerBinade, m = j % kPerBinade;
    return __uint_as_float((unsigned)((127ull + e) << 23 | m));
}
// (self-test of the checker: the quotient without its corrections must be caught)
__device__ __forceinline__ float short_quotient(float N, float D) {
    return N * __builtin_amdgcn_rcpf(D);  // no correction of the quotient at all
}
template <bool SELFTEST>
__global__ void k_check(unsigned long long* bad, unsigned long long* first_bad, int d_from, int d_to) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= kCount) return;
    const float a = nth_integer_float(i);
    unsigned long long nb = 0;
    for (int d = d_from; d < d_to; ++d) {
        const float D = (float)(2 * d);
        const float r1 = a / D, s1 = SELFTEST ? short_quotient(a, D) : fdcm::envelope_quotient(a, D);
        // (N = -0 cannot occur: N is a difference of non-negative values, and x - x is +0; -0 / D = -0 is the one pair the
        // short form gets differently, +0)
        const float na = i == 0 ? a : -a;
        const float r2 = na / D, s2 = SELFTEST ? short_quotient(na, D) : fdcm::envelope_quotient(na, D);
        if (__float_as_uint(r1) != __float_as_uint(s1) || __float_as_uint(r2) != __float_as_uint(s2)) {
            if (nb == 0) atomicMin(first_bad, ((unsigned long long)d << 40) | i);
            ++nb;
        }
    }
    if (i == 0) {  // N = -FLT_MAX: what the test sees over the seedless bottom column of a row's first segment (f = FLT_MAX absorbs the rest)
        for (int d = d_from; d < d_to; ++d) {
            const float D = (float)(2 * d), big = -3.402823466e+38f;
            if (__float_as_uint(big / D) != __float_as_uint(SELFTEST ? short_quotient(big, D) : fdcm::envelope_quotient(big, D))) ++nb;
        }
    }
    if (nb) atomicAdd(bad, nb);
}
int main(int argc, char** argv) {
    const bool selftest = argc > 1 && !strcmp(argv[1], "selftest");
    unsigned long long *bad, *first;
    if (hipMalloc(&bad, 8) != hipSuccess || hipMalloc(&first, 8) != hipSuccess) { printf("no device\n"); return 2; }
    (void)hipMemset(bad, 0, 8); (void)hipMemset(first, 0xff, 8);
    const int per = 512;
    for (int d0 = 1; d0 <= 16383; d0 += per) {
        const int d1 = d0 + per > 16384 ? 16384 : d0 + per;
        if (selftest) hipLaunchKernelGGL(k_check<true>, dim3((unsigned)((kCount + 255) / 256)), dim3(256), 0, 0, bad, first, d0, d1);
        else hipLaunchKernelGGL(k_check<false>, dim3((unsigned)((kCount + 255) / 256)), dim3(256), 0, 0, bad, first, d0, d1);
    }
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    unsigned long long hb = 0, hf = 0;
    (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&hf, first, 8, hipMemcpyDeviceToHost);
    printf("{\"mode\": \"%s\", \"numerators\": %llu, \"denominators\": 16383, \"pairs_checked\": %.4g, \"signs\": 2, \"mismatches\": %llu", selftest ? "selftest: N * rcp(D) without the correction, must mismatch" : "envelope_quotient", kCount, 2.0 * (double)kCount * 16383.0, hb);
    if (hb) printf(", \"first\": {\"d\": %llu, \"numerator_index\": %llu}", hf >> 40, hf & ((1ull << 40) - 1));
    printf("}\n");
    return selftest ? (hb ? 0 : 1) : (hb ? 1 : 0);
}
