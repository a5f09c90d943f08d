// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE on gfx950 for the access patterns of this library
// (MI355X_MICROARCH.md, HBM: "Other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Every kernel reads a known number of distinct 64-byte sectors, each exactly once, from a 1 GiB table
// (four times the Infinity Cache, and every kernel starts behind a 1 GiB flush), so bytes-needed = sectors * 64:
//   calib_stream16    16 B per lane, consecutive (the build's streaming loads)            -> guide: counter = 1/2 bytes
//   calib_stream4     4 B per lane, consecutive (256 B per wave operation)
//   calib_gather4     one 4-byte load per lane, every lane in a different random 64-B sector (the search's worst case)
//   calib_gather4x4   one 4-byte load per lane, 4 neighbouring lanes share a 64-B sector, sectors random
//                     (the search on the interleaved volume: 4 x 4 pixels per sector)
//   calib_gather4p2   as calib_gather4, but the two 64-B sectors of a 128-B line are read by lanes l and l ^ 1
// Build + run on the GPU box: tools/fetch_calib.sh (rocprofv3 --pmc FETCH_SIZE, then TCC_EA0_RDREQ_sum passes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void calib_flush(const uint4* __restrict__ t, size_t n16, unsigned* sink) {
    unsigned a = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) a ^= t[i].x;
    if (a == 0x12345678u) *sink = a;
}
__global__ void calib_stream16(const uint4* __restrict__ t, size_t n16, unsigned* sink) {
    unsigned a = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) a ^= t[i].x ^ t[i].w;
    if (a == 0x12345678u) *sink = a;
}
__global__ void calib_stream4(const unsigned* __restrict__ t, size_t n4, unsigned* sink) {
    unsigned a = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) a ^= t[i];
    if (a == 0x12345678u) *sink = a;
}
// idx[i] = a sector number; lane i reads the dword `word(i)` of that sector
__global__ void calib_gather4(const unsigned* __restrict__ t, const unsigned* __restrict__ idx, size_t n, unsigned* sink) {
    unsigned a = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        a ^= t[(size_t)idx[i] * 16 + (i & 15)];
    if (a == 0x12345678u) *sink = a;
}
__global__ void calib_gather4x4(const unsigned* __restrict__ t, const unsigned* __restrict__ idx, size_t n, unsigned* sink) {
    unsigned a = 0;  // n lanes, n / 4 sectors: lanes 4j .. 4j+3 read dwords 0, 5, 10, 15 of sector idx[j]
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        a ^= t[(size_t)idx[i >> 2] * 16 + (i & 3) * 5];
    if (a == 0x12345678u) *sink = a;
}
__global__ void calib_gather4p2(const unsigned* __restrict__ t, const unsigned* __restrict__ idx, size_t n, unsigned* sink) {
    unsigned a = 0;  // n lanes, n / 2 lines of 128 B: lanes 2j, 2j+1 read the two sectors of line idx[j]
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        a ^= t[(size_t)idx[i >> 1] * 32 + (i & 1) * 16 + (i & 15)];
    if (a == 0x12345678u) *sink = a;
}

int main() {
    const size_t bytes = (size_t)1 << 30, nsec = bytes / 64;
    unsigned *table, *flush, *sink, *d_idx;
    CK(hipMalloc(&table, bytes)); CK(hipMalloc(&flush, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(table, 1, bytes)); CK(hipMemset(flush, 2, bytes));
    // a random quarter of the sectors (of the 128-B lines for p2), each once
    std::vector<unsigned> perm(nsec);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(7);
    std::shuffle(perm.begin(), perm.end(), rng);
    const size_t ns = nsec / 4;
    CK(hipMalloc(&d_idx, ns * 4));
    CK(hipMemcpy(d_idx, perm.data(), ns * 4, hipMemcpyHostToDevice));
    std::vector<unsigned> lines(nsec / 2);
    std::iota(lines.begin(), lines.end(), 0u);
    std::shuffle(lines.begin(), lines.end(), rng);
    unsigned* d_lines;
    const size_t nl = nsec / 8;  // 128-B lines read by gather4p2 (= ns / 2: the same number of sectors)
    CK(hipMalloc(&d_lines, nl * 4));
    CK(hipMemcpy(d_lines, lines.data(), nl * 4, hipMemcpyHostToDevice));
    const dim3 g(4096), b(256);
    auto fl = [&] { hipLaunchKernelGGL(calib_flush, g, b, 0, 0, (const uint4*)flush, bytes / 16, sink); };
    for (int rep = 0; rep < 3; ++rep) {
        fl(); hipLaunchKernelGGL(calib_stream16, g, b, 0, 0, (const uint4*)table, bytes / 16, sink);
        fl(); hipLaunchKernelGGL(calib_stream4, g, b, 0, 0, (const unsigned*)table, bytes / 4, sink);
        fl(); hipLaunchKernelGGL(calib_gather4, g, b, 0, 0, (const unsigned*)table, (const unsigned*)d_idx, ns, sink);
        fl(); hipLaunchKernelGGL(calib_gather4x4, g, b, 0, 0, (const unsigned*)table, (const unsigned*)d_idx, ns * 4, sink);
        fl(); hipLaunchKernelGGL(calib_gather4p2, g, b, 0, 0, (const unsigned*)table, (const unsigned*)d_lines, nl * 2, sink);
    }
    CK(hipDeviceSynchronize());
    // bytes each kernel needs from the table (the index arrays are streamed on top: 4 B per sector / per 4 lanes / per line)
    printf("{\"needed_bytes\": {\"calib_stream16\": %zu, \"calib_stream4\": %zu, \"calib_gather4\": %zu, \"calib_gather4x4\": %zu, "
           "\"calib_gather4p2\": %zu}, \"index_bytes\": {\"calib_gather4\": %zu, \"calib_gather4x4\": %zu, \"calib_gather4p2\": %zu}}\n",
           bytes, bytes, ns * 64, ns * 64, nl * 128, ns * 4, ns * 4, nl * 4);
    return 0;
}
