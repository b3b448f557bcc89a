// micro-benchmark for a consumer-ready (per-instance, column-major) output layout written by the expansion itself:
//   out[instance][col][row][4 words]
// The expansion's lanes are instances (64 per wave) that walk the same rows: a wave's store instruction then puts 16 / 32
// bytes into 64 different arrays (one per instance) - the opposite of the batch-interleaved layout's contiguous 1 KB runs.
// Consecutive rows of one (instance, column) share 128-byte lines, so whether this is viable depends on L2 / Infinity Cache
// merging those partial-line stores before they reach HBM.  Variants: cells of a row over all columns (row order, like an
// op emits them) or RUN consecutive rows of one column at a time (what an LDS-staged epilogue per op could do).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef u64 v2 __attribute__((ext_vector_type(2)));
// mode 0: batch-interleaved reference  [row][col][half][instance]: 1 KB runs
// mode 1: column-major per instance, row order (for row: for col: 32 B)
// mode 2: column-major per instance, RUN rows of a column at a time
template <int MODE, int COLS, int RUN>
__global__ void __launch_bounds__(64) k(u64* out, size_t rows_total, int rows_per_wave, int n_inst) {
    unsigned lane = threadIdx.x;                 // instance
    size_t row0 = (size_t)blockIdx.x * rows_per_wave;
    u64 x = lane * 1315423911ull + blockIdx.x;
    if (MODE == 0) {
        for (int r = 0; r < rows_per_wave; r++)
            for (int c = 0; c < COLS; c++) {
                size_t cell = (row0 + r) * COLS + c;
                v2* p = (v2*)(out + (cell * 2) * (size_t)(2 * n_inst) + 2 * lane);
                v2 v = {x + r, x + c};
                p[0] = v;
                *(v2*)((u64*)p + 2 * n_inst) = v;
            }
    } else if (MODE == 1) {
        u64* base = out + (size_t)lane * COLS * rows_total * 4;
        for (int r = 0; r < rows_per_wave; r++)
            for (int c = 0; c < COLS; c++) {
                v2* p = (v2*)(base + ((size_t)c * rows_total + row0 + r) * 4);
                v2 v = {x + r, x + c};
                p[0] = v;
                p[1] = v;
            }
    } else {
        u64* base = out + (size_t)lane * COLS * rows_total * 4;
        for (int r0 = 0; r0 < rows_per_wave; r0 += RUN)
            for (int c = 0; c < COLS; c++)
#pragma unroll
                for (int r = r0; r < r0 + RUN; r++) {
                    v2* p = (v2*)(base + ((size_t)c * rows_total + row0 + r) * 4);
                    v2 v = {x + r, x + c};
                    p[0] = v;
                    p[1] = v;
                }
    }
}
template <int MODE, int COLS, int RUN>
void run(u64* d, size_t rows_total, int rows_per_wave, int n_inst, const char* what) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    unsigned grid = (unsigned)(rows_total / rows_per_wave);
    for (int it = 0; it < 3; it++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<MODE, COLS, RUN>), dim3(grid), dim3(64), 0, 0, d, rows_total, rows_per_wave, n_inst);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it == 2) printf("%-70s %8.2f ms  %6.2f TB/s\n", what, ms, (double)rows_total * COLS * 32 * n_inst / ms / 1e9);
    }
}
int main() {
    const int n_inst = 64, COLS = 5;
    size_t rows_total = 6400000;   // 64 x 5 x 6.4 M x 32 B = 65.5 GB (the MSM tile's base array)
    u64* d;
    if (hipMalloc(&d, rows_total * COLS * 32 * n_inst) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, rows_total * COLS * 32 * n_inst);
    run<0, COLS, 1>(d, rows_total, 128, n_inst, "batch-interleaved [row][col][half][inst] (1 KB runs), 128 rows/wave");
    run<1, COLS, 1>(d, rows_total, 128, n_inst, "[inst][col][row], row order, 128 rows/wave");
    run<1, COLS, 1>(d, rows_total, 1000, n_inst, "[inst][col][row], row order, 1000 rows/wave");
    run<2, COLS, 4>(d, rows_total, 128, n_inst, "[inst][col][row], 4 rows of a column at a time (128 B), 128 rows/wave");
    run<2, COLS, 16>(d, rows_total, 128, n_inst, "[inst][col][row], 16 rows of a column at a time (512 B), 128 rows/wave");
    run<2, COLS, 32>(d, rows_total, 128, n_inst, "[inst][col][row], 32 rows of a column at a time (1 KB), 128 rows/wave");
    return 0;
}
