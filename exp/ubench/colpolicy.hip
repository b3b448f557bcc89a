// micro-benchmark: can a store cache policy make PARTIAL-line stores to per-instance column arrays merge in the L2 before they reach
// HBM?  exp/ubench/colwrite.hip / colrun.hip (rounds 3-4) found ~43 G write requests/s whatever their size (<= 128 B): per-lane 32-byte
// stores 1.2 TB/s, cooperative 64-byte runs 2.8, 128-byte runs 5.5.  If the partial writes of one line - issued back to back by the same
// wave - were merged in the L2, the expansion could emit halo2's columns without staging four rows of five columns in LDS (40 KB per wave).
//   per-lane: every lane writes RUN consecutive rows (32 B each) of its own instance's column, two 16-byte stores per cell
//   policies: gfx950 store bits sc0 / sc1 / nt in every combination
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef u64 v2 __attribute__((ext_vector_type(2)));
template <int POL>
__device__ __forceinline__ void st(u64* p, v2 v) {
    if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}
// per-lane: RUN rows of one column at a time, COLS columns, [inst][col][row]
template <int POL, int RUN>
__global__ void __launch_bounds__(64) k_lane(u64* out, size_t rows_total, int rows_per_wave) {
    constexpr int COLS = 5;
    unsigned lane = threadIdx.x;
    size_t row0 = (size_t)blockIdx.x * rows_per_wave;
    u64* base = out + (size_t)lane * COLS * rows_total * 4;
    for (int r0 = 0; r0 < rows_per_wave; r0 += RUN)
        for (int c = 0; c < COLS; c++)
#pragma unroll
            for (int r = r0; r < r0 + RUN; r++) {
                u64* p = base + ((size_t)c * rows_total + row0 + r) * 4;
                v2 v = {row0 + r, (u64)c};
                st<POL>(p, v);
                st<POL>(p + 2, v);
            }
}
// cooperative: lanes share runs of RUN bytes (16-byte pieces), one column
template <int POL, int RUN>
__global__ void __launch_bounds__(64) k_coop(u64* out, size_t rows_total, int rows_per_wave) {
    const unsigned lane = threadIdx.x;
    const size_t row0 = (size_t)blockIdx.x * rows_per_wave;
    constexpr int PIECES = RUN / 16, IPS = 64 / PIECES;
    for (int r = 0; r < rows_per_wave; r += RUN / 32)
        for (int i0 = 0; i0 < 64; i0 += IPS) {
            int inst = i0 + lane / PIECES, p = lane % PIECES;
            v2 v = {row0 + r + lane, (u64)inst};
            st<POL>(out + ((size_t)inst * rows_total + row0 + r) * 4 + (size_t)p * 2, v);
        }
}
template <class F>
void timeit(const char* what, double bytes, F&& launch) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int it = 0; it < 3; it++) {
        (void)hipEventRecord(a);
        launch();
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (it > 0 && ms < best) best = ms;
    }
    printf("%-60s %8.2f ms  %6.2f TB/s\n", what, best, bytes / best / 1e9);
}
template <int POL>
void all(u64* d, const char* pol) {
    char buf[128];
    {   // per-lane: 5 columns x 1.6 M rows x 64 instances x 32 B = 16.4 GB
        size_t rows_total = 1600000; int rpw = 128; unsigned grid = (unsigned)(rows_total / rpw);
        double bytes = (double)rows_total * 5 * 32 * 64;
        snprintf(buf, sizeof buf, "[%s] per-lane 32 B, 1 row at a time", pol);
        timeit(buf, bytes, [&] { hipLaunchKernelGGL((k_lane<POL, 1>), dim3(grid), dim3(64), 0, 0, d, rows_total, rpw); });
        snprintf(buf, sizeof buf, "[%s] per-lane 4 rows of a column (128 B per lane)", pol);
        timeit(buf, bytes, [&] { hipLaunchKernelGGL((k_lane<POL, 4>), dim3(grid), dim3(64), 0, 0, d, rows_total, rpw); });
    }
    {   // cooperative: 1 column x 8 M rows x 64 x 32 B = 16.4 GB
        size_t waves = 16384; int rpw = 512; size_t rows_total = waves * rpw; double bytes = (double)rows_total * 64 * 32;
        snprintf(buf, sizeof buf, "[%s] cooperative runs of 32 B", pol);
        timeit(buf, bytes, [&] { hipLaunchKernelGGL((k_coop<POL, 32>), dim3(waves), dim3(64), 0, 0, d, rows_total, rpw); });
        snprintf(buf, sizeof buf, "[%s] cooperative runs of 64 B", pol);
        timeit(buf, bytes, [&] { hipLaunchKernelGGL((k_coop<POL, 64>), dim3(waves), dim3(64), 0, 0, d, rows_total, rpw); });
        snprintf(buf, sizeof buf, "[%s] cooperative runs of 128 B", pol);
        timeit(buf, bytes, [&] { hipLaunchKernelGGL((k_coop<POL, 128>), dim3(waves), dim3(64), 0, 0, d, rows_total, rpw); });
    }
}
int main() {
    u64* d;
    size_t bytes = (size_t)8388608 * 64 * 32;   // 17.2 GB
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(d, 0, bytes);
    all<0>(d, "plain");
    all<1>(d, "sc0");
    all<2>(d, "sc1");
    all<3>(d, "sc0 sc1");
    all<4>(d, "nt");
    all<5>(d, "sc0 nt");
    all<6>(d, "sc1 nt");
    all<7>(d, "sc0 sc1 nt");
    return 0;
}
