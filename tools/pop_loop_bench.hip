// pop_loop_bench.hip -- how long is one pass of the sweep's pop loop (fdcm_sweep.hip, local_run) for a wave by itself and with
// 1..4 waves per SIMD in the same loop?  Every pass pops (z = +inf), so a pass is: quotient, compare, pop from the register
// copy, read of the next ring entry.  Build: hipcc --offload-arch=gfx950 -O3 -o pop_loop_bench pop_loop_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(1024) k_pop(long long* out, int passes, float seed) {
    __shared__ float4 ring[4][1024];
    const int tid = threadIdx.x;
    for (int i = 0; i < 4; ++i) ring[i][tid] = make_float4(2.f * (float)(i + 1), 3.f * (float)(i + tid % 7), __builtin_inff(), 0.f);
    __syncthreads();
    float tv = seed, tp = seed + 1.f, tz = __builtin_inff(), uv = 2.f, up = 5.f, uz = __builtin_inff(), s;
    const float twoq = 4096.f + seed, hq = 1.0e6f;
    int cnt = passes + 8, base = -1, c1;
    unsigned long long sx;
    float qd, qn, qr, qe;
    unsigned ua;
    const unsigned lb = (unsigned)(size_t)&ring[0][tid];
    const long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile(
        "L_pop_%=:\n\t"
        "v_sub_f32 %[qd], %[twoq], %[tv]\n\t"
        "v_sub_f32 %[qn], %[hq], %[tp]\n\t"
        "v_rcp_f32 %[qr], %[qd]\n\t"
        "v_add_u32 %[c1], -1, %[cnt]\n\t"
        "v_mul_f32 %[s], %[qn], %[qr]\n\t"
        "v_fma_f32 %[qe], -%[qd], %[s], %[qn]\n\t"
        "v_fmac_f32 %[s], %[qe], %[qr]\n\t"
        "v_cmp_le_f32 vcc, %[s], %[tz]\n\t"
        "s_cbranch_vccz L_done_%=\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cndmask_b32 %[tv], %[tv], %[uv], vcc\n\t"
        "v_cndmask_b32 %[tp], %[tp], %[up], vcc\n\t"
        "v_cndmask_b32 %[tz], %[tz], %[uz], vcc\n\t"
        "v_cndmask_b32 %[cnt], %[cnt], %[c1], vcc\n\t"
        "v_add_u32 %[ua], -1, %[cnt]\n\t"
        "v_cmp_eq_u32 %[sx], %[cnt], %[base]\n\t"
        "v_and_b32 %[ua], 3, %[ua]\n\t"
        "v_lshl_add_u32 %[ua], %[ua], 14, %[lb]\n\t"
        "ds_read_b32 %[uv], %[ua]\n\t"
        "ds_read_b32 %[up], %[ua] offset:4\n\t"
        "ds_read_b32 %[uz], %[ua] offset:8\n\t"
        "s_and_b64 %[sx], %[sx], vcc\n\t"
        "v_cmp_gt_i32 vcc, %[cnt], %[lim]\n\t"
        "s_cbranch_vccnz L_pop_%=\n"
        "L_done_%=:\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [s] "=&v"(s), [tv] "+v"(tv), [tp] "+v"(tp), [tz] "+v"(tz), [uv] "+v"(uv), [up] "+v"(up), [uz] "+v"(uz), [cnt] "+v"(cnt),
          [sx] "=&s"(sx), [c1] "=&v"(c1), [qd] "=&v"(qd), [qn] "=&v"(qn), [qr] "=&v"(qr), [qe] "=&v"(qe), [ua] "=&v"(ua)
        : [twoq] "v"(twoq), [hq] "v"(hq), [base] "v"(base), [lb] "v"(lb), [lim] "v"(8)
        : "vcc", "scc", "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((tid & 63) == 0) out[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
    if (s == 12345.f) out[0] = (long long)(tv + tp + tz);
}

__global__ void __launch_bounds__(1024) k_pop2(long long* out, int passes, float seed) {
    __shared__ float4 ring[4][1024];
    const int tid = threadIdx.x;
    for (int i = 0; i < 4; ++i) ring[i][tid] = make_float4(2.f * (float)(i + 1), 3.f * (float)(i + tid % 7), __builtin_inff(), 0.f);
    __syncthreads();
    float tv = seed, tp = seed + 1.f, tz = __builtin_inff(), uv = 2.f, up = 5.f, uz = __builtin_inff(), s;
    const float twoq = 4096.f + seed, hq = 1.0e6f;
    int cnt = passes + 8, base = -1, c1;
    unsigned long long sx;
    float qd, qn, qr, qe;
    unsigned ua;
    const unsigned lb = (unsigned)(size_t)&ring[0][tid];
    const long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile(
        "L_pop_%=:\n\t"
        "v_sub_f32 %[qd], %[twoq], %[tv]\n\t"
        "v_sub_f32 %[qn], %[hq], %[tp]\n\t"
        "v_rcp_f32 %[qr], %[qd]\n\t"
        "s_nop 0\n\t"
        "v_mul_f32 %[s], %[qn], %[qr]\n\t"
        "v_fma_f32 %[qe], -%[qd], %[s], %[qn]\n\t"
        "v_fmac_f32 %[s], %[qe], %[qr]\n\t"
        "v_cmp_le_f32 vcc, %[s], %[tz]\n\t"
        "s_cbranch_vccz L_done_%=\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cndmask_b32 %[tv], %[tv], %[uv], vcc\n\t"
        "v_cndmask_b32 %[tp], %[tp], %[up], vcc\n\t"
        "v_cndmask_b32 %[tz], %[tz], %[uz], vcc\n\t"
        "v_subbrev_co_u32 %[cnt], %[sx], 0, %[cnt], vcc\n\t"
        "v_and_b32 %[ua], 3, %[cnt]\n\t"
        "v_cmp_eq_u32 %[sx], %[cnt], %[base]\n\t"
        "v_lshl_add_u32 %[ua], %[ua], 14, %[lb]\n\t"
        "ds_read_b32 %[uv], %[ua]\n\t"
        "ds_read_b32 %[up], %[ua] offset:4\n\t"
        "ds_read_b32 %[uz], %[ua] offset:8\n\t"
        "s_and_b64 %[sx], %[sx], vcc\n\t"
        "v_cmp_gt_i32 vcc, %[cnt], %[lim]\n\t"
        "s_cbranch_vccnz L_pop_%=\n"
        "L_done_%=:\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [s] "=&v"(s), [tv] "+v"(tv), [tp] "+v"(tp), [tz] "+v"(tz), [uv] "+v"(uv), [up] "+v"(up), [uz] "+v"(uz), [cnt] "+v"(cnt),
          [sx] "=&s"(sx), [c1] "=&v"(c1), [qd] "=&v"(qd), [qn] "=&v"(qn), [qr] "=&v"(qr), [qe] "=&v"(qe), [ua] "=&v"(ua)
        : [twoq] "v"(twoq), [hq] "v"(hq), [base] "v"(base), [lb] "v"(lb), [lim] "v"(8)
        : "vcc", "scc", "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((tid & 63) == 0) out[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
    if (s == 12345.f) out[0] = (long long)(tv + tp + tz);
}

__global__ void __launch_bounds__(1024) k_pop3(long long* out, int passes, float seed) {
    __shared__ float4 ring[4][1024];
    const int tid = threadIdx.x;
    {   // structure of arrays: [v | P | z] of 4 entries x 1024 lanes each
        float* f = reinterpret_cast<float*>(&ring[0][0]);
        for (int j = tid; j < 16384; j += blockDim.x) f[j] = (j >= 8192 && j < 12288) ? __builtin_inff() : (float)(1 + j % 13);
    }
__syncthreads();
    float tv = seed, tp = seed + 1.f, tz = __builtin_inff(), uv = 2.f, up = 5.f, uz = __builtin_inff(), s;
    const float twoq = 4096.f + seed, hq = 1.0e6f;
    int cnt = passes + 8, base = -1, c1;
    unsigned long long sx;
    float qd, qn, qr, qe;
    unsigned ua;
    const unsigned lb4 = (unsigned)(size_t)&ring[0][0] + 4u * (unsigned)tid;
    const long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile(
        "L_pop_%=:\n\t"
        "v_sub_f32 %[qd], %[twoq], %[tv]\n\t"
        "v_sub_f32 %[qn], %[hq], %[tp]\n\t"
        "v_rcp_f32 %[qr], %[qd]\n\t"
        "s_nop 0\n\t"
        "v_mul_f32 %[s], %[qn], %[qr]\n\t"
        "v_fma_f32 %[qe], -%[qd], %[s], %[qn]\n\t"
        "v_fmac_f32 %[s], %[qe], %[qr]\n\t"
        "v_cmp_le_f32 vcc, %[s], %[tz]\n\t"
        "s_cbranch_vccz L_done_%=\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cndmask_b32 %[tv], %[tv], %[uv], vcc\n\t"
        "v_cndmask_b32 %[tp], %[tp], %[up], vcc\n\t"
        "v_cndmask_b32 %[tz], %[tz], %[uz], vcc\n\t"
        "v_subbrev_co_u32 %[cnt], %[sx], 0, %[cnt], vcc\n\t"
        "v_and_b32 %[ua], 3, %[cnt]\n\t"
        "v_cmp_eq_u32 %[sx], %[cnt], %[base]\n\t"
        "v_lshl_add_u32 %[ua], %[ua], 12, %[lb4]\n\t"
        "ds_read_b32 %[uv], %[ua]\n\t"
        "ds_read_b32 %[up], %[ua] offset:16384\n\t"
        "ds_read_b32 %[uz], %[ua] offset:32768\n\t"
        "s_and_b64 %[sx], %[sx], vcc\n\t"
        "v_cmp_gt_i32 vcc, %[cnt], %[lim]\n\t"
        "s_cbranch_vccnz L_pop_%=\n"
        "L_done_%=:\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [s] "=&v"(s), [tv] "+v"(tv), [tp] "+v"(tp), [tz] "+v"(tz), [uv] "+v"(uv), [up] "+v"(up), [uz] "+v"(uz), [cnt] "+v"(cnt),
          [sx] "=&s"(sx), [c1] "=&v"(c1), [qd] "=&v"(qd), [qn] "=&v"(qn), [qr] "=&v"(qr), [qe] "=&v"(qe), [ua] "=&v"(ua)
        : [twoq] "v"(twoq), [hq] "v"(hq), [base] "v"(base), [lb4] "v"(lb4), [lim] "v"(8)
        : "vcc", "scc", "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((tid & 63) == 0) out[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
    if (s == 12345.f) out[0] = (long long)(tv + tp + tz);
}

int main() {
    long long* d;
    hipMalloc(&d, 1 << 20);
    const int passes = 4000;
    for (int variant = 0; variant < 3; ++variant)
    for (int waves : {1, 4, 8, 16}) {
        for (int blocks : {1, 256, 512}) {
            hipMemset(d, 0, 1 << 20);
            if (variant == 0) hipLaunchKernelGGL(k_pop, dim3(blocks), dim3(64 * waves), 0, 0, d, passes, 1.0f);
            else if (variant == 2) hipLaunchKernelGGL(k_pop3, dim3(blocks), dim3(64 * waves), 0, 0, d, passes, 1.0f);
            else hipLaunchKernelGGL(k_pop2, dim3(blocks), dim3(64 * waves), 0, 0, d, passes, 1.0f);
            hipDeviceSynchronize();
            std::vector<long long> h((size_t)blocks * 16);
            hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            double sum = 0; int n = 0;
            for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) { sum += (double)h[(size_t)b * 16 + w]; ++n; }
            printf("variant %d, waves per block %2d, blocks %3d: %.1f cycles per pass (s_memtime ticks)\n", variant, waves, blocks, sum / n / passes);
        }
    }
    return 0;
}
