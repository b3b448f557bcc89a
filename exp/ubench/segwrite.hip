// micro-benchmark: every lane streams ROWS rows of 160 bytes into its own contiguous region; a wave writes SEG rows per
// lane per cooperative flush (consecutive lanes cover consecutive 16-byte pieces of one lane's segment), like the
// engine's row flush.  Question: how does HBM write throughput depend on the contiguous segment size?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int SEG>
__global__ void __launch_bounds__(64) k(unsigned long long* out, int rows, size_t lane_stride_u64) {
    __shared__ unsigned long long buf[64][SEG * 20 + 2];
    __shared__ unsigned long long* ptr[64];
    unsigned lane = threadIdx.x;
    size_t gl = (size_t)blockIdx.x * 64 + lane;
    unsigned long long* base = out + gl * lane_stride_u64;
    for (int r0 = 0; r0 < rows; r0 += SEG) {
        for (int j = 0; j < SEG * 20; j++) buf[lane][j] = gl * 1315423911ull + r0 + j;
        ptr[lane] = base + (size_t)r0 * 20;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        constexpr int PIECES = SEG * 10;
#pragma unroll
        for (int i = 0; i < PIECES; i++) {
            unsigned chunk = i * 64 + lane;
            unsigned r = chunk / PIECES, piece = chunk - r * PIECES;
            typedef unsigned long long v2 __attribute__((ext_vector_type(2)));
            v2 v = *(const v2*)&buf[r][piece * 2];
            *(v2*)(ptr[r] + piece * 2) = v;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}
template <int SEG>
void run(unsigned long long* d, size_t lanes, int rows) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    size_t stride = (size_t)rows * 20;
    for (int it = 0; it < 3; it++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<SEG>, dim3(lanes / 64), dim3(64), 0, 0, d, rows, stride);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it == 2) printf("SEG %d rows (%d B segments): %.2f ms, %.2f TB/s\n", SEG, SEG * 160, ms, lanes * (double)rows * 160 / ms / 1e9);
    }
}
int main() {
    size_t lanes = 3332480 / 4;   // 833k lanes
    int rows = 480;              // 76.8 KB per lane -> 64 GB total
    unsigned long long* d;
    if (hipMalloc(&d, lanes * (size_t)rows * 160) != hipSuccess) { printf("alloc failed\n"); return 1; }
    run<1>(d, lanes, rows);
    run<2>(d, lanes, rows);
    run<4>(d, lanes, rows);
    run<8>(d, lanes, rows);
    return 0;
}
