// fdcm_quotient.h -- the quotient of the L2 sweep's envelope test, s = N / D (imgproc.h:111), for the operands the sweep has:
// D = 2 (q - v) an even integer in [2, 2^15); N an integer-valued float with |N| < 2^32, or -FLT_MAX (the bottom of a row's
// first segment is column 0 whether it holds a seed or not; a seedless column's f is FLT_MAX, which absorbs the other terms).
// The compiler's correctly rounded f32 division is v_div_scale (twice), v_rcp, seven fused operations, v_div_fmas and
// v_div_fixup: 11 instructions, 9 of them in a row on the dependent chain of the sweep's inner loop, where an instruction
// costs its ~8 cycles of latency.  With these operands the scales return their inputs, the fix-up has nothing to fix, and
// ONE correction of N * rcp(D) -- with the hardware's reciprocal as it comes, a denominator of at most 15 significant
// bits -- is already the correctly rounded quotient: 4 instructions, the same bits for every operand pair.
// tools/div_check.hip compares the two on all of them (2.75e12 pairs, both signs, and -FLT_MAX over every D; its self-test
// makes sure the comparison bites), tests/test_gpu_parity.py runs it.  The table behind v_rcp_f32 belongs to the chip: the
// check is for gfx950.  (A NaN numerator -- the sweep gives lanes outside their column range one -- stays NaN.)
// Device code only; needs -ffp-contract=off like the rest (the fused operations below are explicit).
#pragma once
#include <hip/hip_runtime.h>

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "envelope_quotient is checked against the division for gfx950's v_rcp_f32 only (tools/div_check.hip): run the check for the new target before building for it"
#endif

namespace fdcm {

__device__ __forceinline__ float envelope_quotient(float N, float D) {
    const float r = __builtin_amdgcn_rcpf(D);
    const float q = N * r;
    const float e = __builtin_fmaf(-D, q, N);
    return __builtin_fmaf(e, r, q);
}

}  // namespace fdcm
