// FETCH_SIZE calibration on gfx950 for the access patterns k_march uses: per-lane 3 x u8
// (stride 3 bytes), per-lane 1 x dword, and (reference) per-lane dwordx4.  Each kernel reads
// exactly `bytes` bytes once; compare rocprofv3 FETCH_SIZE with that.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void read_u8x3(const uint8_t* p, size_t npx, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npx; i += (size_t)gridDim.x * blockDim.x) {
        const uint8_t* q = p + i * 3;
        acc += q[0] + q[1] + q[2];
    }
    if (acc == 0xdeadbeef) out[0] = acc;
}
__global__ void read_dword(const uint32_t* p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0xdeadbeef) out[0] = acc;
}
__global__ void read_dwordx4(const uint4* p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 0xdeadbeef) out[0] = acc;
}
int main() {
    const size_t bytes = 600ull << 20;  // 600 MiB > 256 MiB Infinity Cache
    uint8_t* d; uint32_t* o;
    hipMalloc(&d, bytes); hipMalloc(&o, 64); hipMemset(d, 1, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(read_u8x3, dim3(4096), dim3(256), 0, 0, d, bytes / 3, o);
        hipLaunchKernelGGL(read_dword, dim3(4096), dim3(256), 0, 0, (const uint32_t*)d, bytes / 4, o);
        hipLaunchKernelGGL(read_dwordx4, dim3(4096), dim3(256), 0, 0, (const uint4*)d, bytes / 16, o);
    }
    hipDeviceSynchronize();
    printf("each kernel read %zu bytes = %.1f KiB\n", bytes, bytes / 1024.0);
    return 0;
}
