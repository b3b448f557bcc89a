// Does the HIP virtual-memory API let ONE physical allocation be mapped at several virtual addresses on this stack (ROCm 7.2, MI355X)?
// What h2e_ring (csrc/ring.hpp) relies on: three runs in flight whose big launch's rows share two physical copies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill(unsigned long long* p, size_t n, unsigned long long v) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = v + i;
}
int main() {
    int dev = 0;
    CK(hipSetDevice(dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t gmin = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    printf("granularity recommended %zu minimum %zu\n", gran, gmin);
    const size_t piece = ((size_t)3 << 30) / gran * gran;   // 3 GB
    hipMemGenericAllocationHandle_t hA, hB0, hB1;
    CK(hipMemCreate(&hA, piece, &prop, 0));
    CK(hipMemCreate(&hB0, piece, &prop, 0));
    CK(hipMemCreate(&hB1, piece, &prop, 0));
    void *va0 = nullptr, *va1 = nullptr;
    CK(hipMemAddressReserve(&va0, 2 * piece, gran, nullptr, 0));
    CK(hipMemAddressReserve(&va1, 2 * piece, gran, nullptr, 0));
    CK(hipMemMap(va0, piece, 0, hB0, 0));
    CK(hipMemMap((char*)va0 + piece, piece, 0, hA, 0));
    CK(hipMemMap(va1, piece, 0, hB1, 0));
    hipError_t e = hipMemMap((char*)va1 + piece, piece, 0, hA, 0);   // the same physical piece a second time
    printf("second mapping of one handle: %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 2;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va0, 2 * piece, &acc, 1));
    CK(hipMemSetAccess(va1, 2 * piece, &acc, 1));
    size_t n = piece / 8;
    unsigned long long* a0 = (unsigned long long*)((char*)va0 + piece);
    unsigned long long* a1 = (unsigned long long*)((char*)va1 + piece);
    hipLaunchKernelGGL(fill, dim3((n + 255) / 256), dim3(256), 0, 0, a0, n, 0x1111000000000000ull);
    CK(hipDeviceSynchronize());
    unsigned long long h[4];
    CK(hipMemcpy(h, a1 + 12345, 32, hipMemcpyDeviceToHost));
    printf("written through mapping 0, read through mapping 1: %llx %llx (want %llx)\n", h[0], h[1], 0x1111000000000000ull + 12345);
    hipLaunchKernelGGL(fill, dim3((n + 255) / 256), dim3(256), 0, 0, a1, n, 0x2222000000000000ull);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, a0 + n - 4, 32, hipMemcpyDeviceToHost));
    printf("written through mapping 1, read through mapping 0: %llx (want %llx)\n", h[3], 0x2222000000000000ull + n - 1);
    // a kernel that streams through a mapped range at full rate?
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int it = 0; it < 3; it++) {
        CK(hipEventRecord(t0));
        hipLaunchKernelGGL(fill, dim3((2 * n + 255) / 256), dim3(256), 0, 0, (unsigned long long*)va0, 2 * n, 7ull);
        CK(hipEventRecord(t1));
        CK(hipEventSynchronize(t1));
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        printf("fill of 6 GB through the mapped range: %.3f ms = %.2f TB/s\n", ms, 2.0 * piece / ms / 1e9);
    }
    void* plain = nullptr;
    CK(hipMalloc(&plain, 2 * piece));
    for (int it = 0; it < 3; it++) {
        CK(hipEventRecord(t0));
        hipLaunchKernelGGL(fill, dim3((2 * n + 255) / 256), dim3(256), 0, 0, (unsigned long long*)plain, 2 * n, 7ull);
        CK(hipEventRecord(t1));
        CK(hipEventSynchronize(t1));
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        printf("fill of 6 GB of hipMalloc memory:       %.3f ms = %.2f TB/s\n", ms, 2.0 * piece / ms / 1e9);
    }
    CK(hipMemUnmap(va0, piece)); CK(hipMemUnmap((char*)va0 + piece, piece)); CK(hipMemUnmap(va1, piece)); CK(hipMemUnmap((char*)va1 + piece, piece));
    CK(hipMemAddressFree(va0, 2 * piece)); CK(hipMemAddressFree(va1, 2 * piece));
    CK(hipMemRelease(hA)); CK(hipMemRelease(hB0)); CK(hipMemRelease(hB1));
    printf("ok\n");
    return 0;
}
