// micro-benchmark for "consumer-ready columns straight from the expansion" (VERDICT r3 #7): 64 per-instance column
// arrays; a wave owns a stretch of rows of one column and writes it for all 64 instances, RUN bytes of one instance's column
// at a time (lanes cooperate: one 16-byte piece each, so a store instruction covers 1024 / RUN instances' runs, or - RUN > 1 KB -
// a 1 KB piece of one run).  What the LDS-transposed store path of the expansion would do, without the arithmetic.
//   mode 0: the batch-interleaved reference: the same bytes as 1 KB runs of [row][half][instance][16 B]
// Question: how large must the contiguous per-instance run be for HBM to take it at the rate of the interleaved layout?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef u64 v2 __attribute__((ext_vector_type(2)));
// rows_per_wave rows of 32 B per instance and wave; arrays: [instance][rows_total][32 B]
template <int RUN>
__global__ void __launch_bounds__(64) k_runs(u64* out, size_t rows_total, int rows_per_wave) {
    const unsigned lane = threadIdx.x;
    const size_t row0 = (size_t)blockIdx.x * rows_per_wave;
    constexpr int PIECES = RUN / 16;                        // 16-byte pieces per run
    constexpr int INST_PER_STORE = PIECES >= 64 ? 1 : 64 / PIECES;
    for (int r = 0; r < rows_per_wave; r += RUN / 32) {     // one run of every instance per iteration
        if constexpr (PIECES >= 64) {
            for (int inst = 0; inst < 64; inst++)
                for (int p = 0; p < PIECES; p += 64) {
                    v2 v = {row0 + r + lane, (u64)inst};
                    *(v2*)(out + ((size_t)inst * rows_total + row0 + r) * 4 + (size_t)(p + lane) * 2) = v;
                }
        } else {
            for (int i0 = 0; i0 < 64; i0 += INST_PER_STORE) {
                int inst = i0 + lane / PIECES, p = lane % PIECES;
                v2 v = {row0 + r + lane, (u64)inst};
                *(v2*)(out + ((size_t)inst * rows_total + row0 + r) * 4 + (size_t)p * 2) = v;
            }
        }
    }
}
__global__ void __launch_bounds__(64) k_interleaved(u64* out, size_t rows_total, int rows_per_wave) {
    const unsigned lane = threadIdx.x;
    const size_t row0 = (size_t)blockIdx.x * rows_per_wave;
    for (int r = 0; r < rows_per_wave; r++)
        for (int half = 0; half < 2; half++) {
            v2 v = {row0 + r + lane, (u64)half};
            *(v2*)(out + (((row0 + r) * 2 + half) * 64 + lane) * 2) = v;
        }
}
template <class F>
void timeit(const char* what, double bytes, F&& launch) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e9f;
    for (int it = 0; it < 4; it++) {
        hipEventRecord(a);
        launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (it > 0 && ms < best) best = ms;
    }
    printf("%-34s %8.2f ms  %6.2f TB/s\n", what, best, bytes / best / 1e9);
}
int main(int argc, char** argv) {
    const int rows_per_wave = argc > 1 ? atoi(argv[1]) : 512;   // a sub-range's stretch of one column
    const size_t waves = 32768;
    const size_t rows_total = waves * (size_t)rows_per_wave;     // per instance
    const double bytes = (double)rows_total * 64 * 32;
    u64* d;
    if (hipMalloc(&d, (size_t)bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, (size_t)bytes);
    printf("%zu waves x %d rows x 64 instances x 32 B = %.1f GB\n", waves, rows_per_wave, bytes / 1e9);
    timeit("batch-interleaved 1 KB runs", bytes, [&] { hipLaunchKernelGGL(k_interleaved, dim3(waves), dim3(64), 0, 0, d, rows_total, rows_per_wave); });
    timeit("per-instance runs of 64 B", bytes, [&] { hipLaunchKernelGGL(k_runs<64>, dim3(waves), dim3(64), 0, 0, d, rows_total, rows_per_wave); });
    timeit("per-instance runs of 128 B", bytes, [&] { hipLaunchKernelGGL(k_runs<128>, dim3(waves), dim3(64), 0, 0, d, rows_total, rows_per_wave); });
    timeit("per-instance runs of 256 B", bytes, [&] { hipLaunchKernelGGL(k_runs<256>, dim3(waves), dim3(64), 0, 0, d, rows_total, rows_per_wave); });
    timeit("per-instance runs of 512 B", bytes, [&] { hipLaunchKernelGGL(k_runs<512>, dim3(waves), dim3(64), 0, 0, d, rows_total, rows_per_wave); });
    timeit("per-instance runs of 1024 B", bytes, [&] { hipLaunchKernelGGL(k_runs<1024>, dim3(waves), dim3(64), 0, 0, d, rows_total, rows_per_wave); });
    timeit("per-instance runs of 2048 B", bytes, [&] { hipLaunchKernelGGL(k_runs<2048>, dim3(waves), dim3(64), 0, 0, d, rows_total, rows_per_wave); });
    return 0;
}
