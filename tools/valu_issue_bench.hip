// valu_issue_bench.hip -- what does a vector instruction cost a SIMD of gfx950 when 1, 2 or 4 waves run the same stream on it?
// (diagnostic, not product code).  Each kernel repeats a block of 32 instructions; the table is s_memtime cycles per
// instruction and wave, and the same divided by the waves per SIMD (= cycles of the SIMD per instruction when it is the bound).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue_bench valu_issue_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define R32(x) R8(x) R8(x) R8(x) R8(x)

#define KERNEL(name, body, ...)                                                                             \
    __global__ void __launch_bounds__(1024) name(long long* out, int iters, float seed) {                   \
        __shared__ float lds[4096];                                                                         \
        const int tid = threadIdx.x;                                                                        \
        for (int i = tid; i < 4096; i += blockDim.x) lds[i] = (float)i;                                     \
        __syncthreads();                                                                                    \
        float a0 = seed, a1 = seed + 1.f, a2 = seed + 2.f, a3 = seed + 3.f, a4 = seed + 4.f, a5 = seed + 5.f, a6 = seed + 6.f, a7 = seed + 7.f; \
        float b = 1.0001f, c = 0.5f;                                                                        \
        unsigned la = (unsigned)(size_t)&lds[tid];                                                          \
        const long long t0 = __builtin_amdgcn_s_memtime();                                                  \
        for (int it = 0; it < iters; ++it) {                                                                \
            asm volatile(body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(la) : "vcc", "memory", ##__VA_ARGS__); \
        }                                                                                                   \
        const long long t1 = __builtin_amdgcn_s_memtime();                                                  \
        if ((tid & 63) == 0) out[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;                                  \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) out[0] = 1;                                  \
    }

KERNEL(k_fma_indep, R4("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"))
KERNEL(k_fma_dep, R32("v_fma_f32 %0, %0, %8, %9\n"))
KERNEL(k_mov, R4("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"))
KERNEL(k_add_u32, R4("v_add_u32 %0, 1, %0\n v_add_u32 %1, 1, %1\n v_add_u32 %2, 1, %2\n v_add_u32 %3, 1, %3\n v_add_u32 %4, 1, %4\n v_add_u32 %5, 1, %5\n v_add_u32 %6, 1, %6\n v_add_u32 %7, 1, %7\n"))
KERNEL(k_cndmask, R4("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"))
KERNEL(k_cmp, R4("v_cmp_le_f32 vcc, %0, %1\n v_cmp_le_f32 vcc, %1, %2\n v_cmp_le_f32 vcc, %2, %3\n v_cmp_le_f32 vcc, %3, %4\n v_cmp_le_f32 vcc, %4, %5\n v_cmp_le_f32 vcc, %5, %6\n v_cmp_le_f32 vcc, %6, %7\n v_cmp_le_f32 vcc, %7, %0\n"))
KERNEL(k_rcp, R4("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"))
KERNEL(k_lshl_add, R4("v_lshl_add_u32 %0, %0, 1, %10\n v_lshl_add_u32 %1, %1, 1, %10\n v_lshl_add_u32 %2, %2, 1, %10\n v_lshl_add_u32 %3, %3, 1, %10\n v_lshl_add_u32 %4, %4, 1, %10\n v_lshl_add_u32 %5, %5, 1, %10\n v_lshl_add_u32 %6, %6, 1, %10\n v_lshl_add_u32 %7, %7, 1, %10\n"))
KERNEL(k_ds_read, R4("ds_read_b32 %0, %10\n ds_read_b32 %1, %10 offset:4096\n ds_read_b32 %2, %10 offset:8192\n ds_read_b32 %3, %10 offset:12288\n ds_read_b32 %4, %10\n ds_read_b32 %5, %10 offset:4096\n ds_read_b32 %6, %10 offset:8192\n ds_read_b32 %7, %10 offset:12288\n") "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_ds_write, R4("ds_write_b32 %10, %0\n ds_write_b32 %10, %1 offset:4096\n ds_write_b32 %10, %2 offset:8192\n ds_write_b32 %10, %3 offset:12288\n ds_write_b32 %10, %4\n ds_write_b32 %10, %5 offset:4096\n ds_write_b32 %10, %6 offset:8192\n ds_write_b32 %10, %7 offset:12288\n") "s_waitcnt lgkmcnt(0)\n")
// 16 vector instructions + 4 LDS reads + 2 LDS writes, twice: roughly the mix of a pass of the sweep's local run
KERNEL(k_mix, R4("v_sub_f32 %0, %1, %2\n v_rcp_f32 %3, %0\n v_add_u32 %4, 1, %4\n ds_read_b32 %5, %10\n v_mul_f32 %6, %0, %3\n v_fma_f32 %7, -%0, %6, %1\n ds_write_b32 %10, %2 offset:8192\n v_cmp_le_f32 vcc, %6, %7\n") "s_waitcnt lgkmcnt(0)\n")


KERNEL(k_cndmask_sgpr, "s_mov_b64 s[20:21], 0x5555\n" R4("v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cndmask_b32 %1, %1, %2, s[20:21]\n v_cndmask_b32 %2, %2, %3, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]\n v_cndmask_b32 %4, %4, %5, s[20:21]\n v_cndmask_b32 %5, %5, %6, s[20:21]\n v_cndmask_b32 %6, %6, %7, s[20:21]\n v_cndmask_b32 %7, %7, %0, s[20:21]\n"), "s20", "s21")
KERNEL(k_cndmask_vccs, "s_mov_b64 vcc, 0x5555\n" R4("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"))
KERNEL(k_cndmask_const, "s_mov_b64 vcc, 0x5555\n" R4("v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n"))
KERNEL(k_mov_exec, "s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0x5555\n" R4("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n") "s_mov_b64 exec, s[20:21]\n", "s20", "s21")
KERNEL(k_bfi, R4("v_bfi_b32 %0, %8, %0, %1\n v_bfi_b32 %1, %8, %1, %2\n v_bfi_b32 %2, %8, %2, %3\n v_bfi_b32 %3, %8, %3, %4\n v_bfi_b32 %4, %8, %4, %5\n v_bfi_b32 %5, %8, %5, %6\n v_bfi_b32 %6, %8, %6, %7\n v_bfi_b32 %7, %8, %7, %0\n"))
KERNEL(k_exec_flip, R8("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0x5555\n v_mov_b32 %0, %1\n s_mov_b64 exec, s[20:21]\n"), "s20", "s21")
KERNEL(k_cmp_branch, R8("v_cmp_le_f32 vcc, %0, %1\n s_cbranch_vccz 1f\n v_add_f32 %0, %0, %8\n1:\n v_add_f32 %1, %1, %9\n"))
KERNEL(k_sub_f32, R4("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n"))

KERNEL(k_cndmask_e64vcc, "s_mov_b64 vcc, 0x5555\n" R4("v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc\n v_cndmask_b32_e64 %2, %2, %3, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc\n v_cndmask_b32_e64 %4, %4, %5, vcc\n v_cndmask_b32_e64 %5, %5, %6, vcc\n v_cndmask_b32_e64 %6, %6, %7, vcc\n v_cndmask_b32_e64 %7, %7, %0, vcc\n"))
KERNEL(k_cmp_cnd, R4("v_cmp_le_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_add_f32 %5, %5, %9\n v_cmp_le_f32 vcc, %1, %8\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %0, %0, %5, vcc\n v_add_f32 %2, %2, %9\n"))
KERNEL(k_cmp_cnd64, R4("v_cmp_le_f32 s[20:21], %0, %8\n v_cndmask_b32 %1, %1, %2, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]\n v_add_f32 %5, %5, %9\n v_cmp_le_f32 s[20:21], %1, %8\n v_cndmask_b32 %6, %6, %7, s[20:21]\n v_cndmask_b32 %0, %0, %5, s[20:21]\n v_add_f32 %2, %2, %9\n"), "s20", "s21")
KERNEL(k_addc, "s_mov_b64 vcc, 0x5555\n" R4("v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n v_addc_co_u32 %2, vcc, %2, %3, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n v_addc_co_u32 %4, vcc, %4, %5, vcc\n v_addc_co_u32 %5, vcc, %5, %6, vcc\n v_addc_co_u32 %6, vcc, %6, %7, vcc\n v_addc_co_u32 %7, vcc, %7, %0, vcc\n"))
KERNEL(k_cndmask_ones, "s_mov_b64 vcc, -1\n" R4("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"))
KERNEL(k_min_f32, R4("v_min_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_min_f32 %3, %3, %4\n v_min_f32 %4, %4, %5\n v_min_f32 %5, %5, %6\n v_min_f32 %6, %6, %7\n v_min_f32 %7, %7, %0\n"))
KERNEL(k_readlane, R4("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s20, %2, 7\n v_readlane_b32 s21, %3, 9\n v_readlane_b32 s20, %4, 11\n v_readlane_b32 s21, %5, 13\n v_readlane_b32 s20, %6, 15\n v_readlane_b32 s21, %7, 17\n"), "s20", "s21")
KERNEL(k_dpp, R4("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"))

typedef void (*kfn)(long long*, int, float);
int main() {
    long long* d;
    hipMalloc(&d, 256 * 16 * 8);
    struct K { const char* name; kfn f; int per; } ks[] = {
        {"v_fma_f32 independent", k_fma_indep, 32}, {"v_fma_f32 dependent", k_fma_dep, 32}, {"v_mov_b32", k_mov, 32}, {"v_add_u32", k_add_u32, 32},
        {"v_cndmask_b32 vcc", k_cndmask, 32}, {"v_cmp_le_f32 vcc", k_cmp, 32}, {"v_rcp_f32", k_rcp, 32}, {"v_lshl_add_u32", k_lshl_add, 32},
        {"ds_read_b32 (free of conflicts)", k_ds_read, 32}, {"ds_write_b32", k_ds_write, 32}, {"mix (6 valu + 1 read + 1 write) x 4", k_mix, 32},
        {"v_cndmask_b32 sgpr pair", k_cndmask_sgpr, 32}, {"v_cndmask_b32 vcc set by s_mov", k_cndmask_vccs, 32}, {"v_cndmask_b32 vcc, fixed sources", k_cndmask_const, 32},
        {"v_mov_b32 under a half exec mask", k_mov_exec, 32}, {"v_bfi_b32", k_bfi, 32}, {"(save exec, set, v_mov, restore) per 4", k_exec_flip, 32},
        {"(v_cmp, s_cbranch_vccz, 2 v_add) per 4", k_cmp_branch, 32}, {"v_sub_f32", k_sub_f32, 32},
        {"v_cndmask_b32_e64 with vcc", k_cndmask_e64vcc, 32}, {"(v_cmp vcc, 2 v_cndmask vcc, v_add) x 2", k_cmp_cnd, 32}, {"(v_cmp sgpr, 2 v_cndmask sgpr, v_add) x 2", k_cmp_cnd64, 32},
        {"v_addc_co_u32 vcc", k_addc, 32}, {"v_cndmask_b32 vcc = all ones", k_cndmask_ones, 32}, {"v_min_f32", k_min_f32, 32}, {"v_readlane_b32", k_readlane, 32}, {"v_mov_b32_dpp row_shr", k_dpp, 32}};
    const int iters = 2000;
    printf("%-40s %28s %28s %28s\n", "cycles per instruction: wave / SIMD", "1 wave per SIMD", "2 waves per SIMD", "4 waves per SIMD");
    for (auto& k : ks) {
        printf("%-40s", k.name);
        for (int wps : {1, 2, 4}) {
            const int threads = 256 * wps;
            hipMemset(d, 0, 256 * 16 * 8);
            hipLaunchKernelGGL(k.f, dim3(256), dim3(threads), 0, 0, d, 10, 1.5f);
            hipLaunchKernelGGL(k.f, dim3(256), dim3(threads), 0, 0, d, iters, 1.5f);
            hipDeviceSynchronize();
            std::vector<long long> h(256 * 16);
            hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            double sum = 0; int n = 0;
            for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) { sum += (double)h[b * 16 + w]; ++n; }
            const double per = sum / n / ((double)iters * k.per);
            printf(" %13.2f / %-12.2f", per, per / wps);
        }
        printf("\n");
    }
    return 0;
}
