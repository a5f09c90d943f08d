// copy_bw.hip -- what a plain 1:1 read/write stream reaches on this GPU (the ceiling for the propagation and the line
// integral, which read V and write V): a 16-byte-per-lane copy kernel and hipMemcpyAsync device-to-device, 1 GiB.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy16(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    uint4 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192, 32768}) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, a, b, bytes / 16);
        hipEventRecord(e0);
        for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, a, b, bytes / 16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("copy16 grid %d: %.3f ms per GiB copied = %.2f TB/s (read + write)\n", grid, ms / 10, 2.0 * bytes / (ms / 10 * 1e-3) / 1e12);
    }
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpyAsync D2D: %.3f ms per GiB = %.2f TB/s (read + write)\n", ms / 10, 2.0 * bytes / (ms / 10 * 1e-3) / 1e12);
    return 0;
}
