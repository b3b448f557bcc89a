// Hand-off kernels that do not depend on a field pair: what leaves a finished run for its consumer.
//
//  * h2e_engine_unit_records - the per-unit record table of a job's one collective ({status, Offset, result point
//    cells, 32-byte digest per advice array} per MSM tile / pairing instance).  The reference has no multi-context driver - a
//    unit is its own `Context` (src/context.rs:136-143) whose `Offset` (src/circuit/ecc_chip.rs:36-62) and result handles the
//    caller reads back - so this is the build's definition; one kernel, one lane per (unit, word), straight into the buffer the
//    host hands to ncclAllGather.
//
//  * h2e_engine_instance_table - a run's table of per-instance descriptors (engine.hip InstanceDesc: where instance i's cells, inputs, status
//    word and workspace start), written ON the device from the nine base addresses and strides the host knows: every field is affine in
//    i.  Rounds 1-4 filled a pinned host table per job slot and copied it; the host then had to wait (hipEventSynchronize) until the
//    slot's previous copy had been read before rewriting the table - in a pipelined job that wait was where every submission stalled.
//
// A translation unit of its own (seconds to compile; engine.hip is minutes per field pair).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;

struct H2EUnitRecArgs {
    const u64* base;      // batch-interleaved base array [row][5][half][instance][2 words]
    const u32* status;    // [instance]
    const u64* digests;   // [3][instance][4] or null
    u64* out;             // [instance][out_stride]
    u64 offsets[3];       // the program's Offset (base, range, select rows consumed)
    u32 refs[9];          // cell references (col << 27 | row) of the result point's 2 x limbs coordinate limbs, then of z
    u32 limbs;            // 3 or 4
    u32 has_point;        // 0: a workload without a result point (the words stay zero)
    u32 n_instances;
    u32 out_stride;       // words per row of `out` (>= record words; a caller's leading index column is outside the record)
};

__global__ void __launch_bounds__(256) h2e_unit_records_k(H2EUnitRecArgs a) {
    const u32 R = 1 + 3 + 4 * a.limbs + 1 + 12;
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 unit = t / R, w = t % R;
    if (unit >= a.n_instances) return;
    const u32 n = a.n_instances;
    auto cell_word = [&](u32 ref, u32 word) -> u64 {   // word 0 / 1 of the cell's low half (limbs are < 2^128)
        u64 row = ref & 0x3FFFFFFu, col = (ref >> 27) & 7u;
        return a.base[(((row * 5 + col) * 2 + 0) * n + unit) * 2 + word];
    };
    u64 v = 0;
    if (w == 0) v = (u64)(int64_t)(int32_t)a.status[unit];   // (the status word as the int32 the boundary hands out, sign-extended)
    else if (w < 4) v = a.offsets[w - 1];
    else if (w < 4 + 4 * a.limbs) {
        if (a.has_point) v = cell_word(a.refs[(w - 4) >> 1], (w - 4) & 1);
    } else if (w == 4 + 4 * a.limbs) {
        if (a.has_point) v = cell_word(a.refs[2 * a.limbs], 0);
    } else if (a.digests) {
        u32 k = w - (R - 12);
        v = a.digests[((size_t)(k >> 2) * n + unit) * 4 + (k & 3)];
    }
    a.out[(size_t)unit * a.out_stride + w] = v;
}

extern "C" int h2e_engine_unit_records(const void* base, const void* status, const void* digests, void* out, const uint64_t* offsets3,
                                       const uint32_t* refs, uint32_t limbs, int has_point, uint32_t n_instances, uint32_t out_stride,
                                       hipStream_t stream) {
    if (n_instances == 0) return 0;
    H2EUnitRecArgs a;
    a.base = (const u64*)base;
    a.status = (const u32*)status;
    a.digests = (const u64*)digests;
    a.out = (u64*)out;
    for (int i = 0; i < 3; i++) a.offsets[i] = offsets3[i];
    for (int i = 0; i < 9; i++) a.refs[i] = (has_point && (u32)i <= 2 * limbs) ? refs[i] : 0;
    a.limbs = limbs;
    a.has_point = has_point ? 1 : 0;
    a.n_instances = n_instances;
    a.out_stride = out_stride;
    const u32 R = 1 + 3 + 4 * limbs + 1 + 12;
    const u32 threads = n_instances * R;
    hipLaunchKernelGGL(h2e_unit_records_k, dim3((threads + 255) / 256), dim3(256), 0, stream, a);
    return (int)hipGetLastError();
}

// A run may be made of several caller batches (h2e.h h2e_submit_batches): every batch has its own arrays, inputs and status words, each
// batch-interleaved over ITS instances - so the caller-side pointers are piecewise affine in the instance index (instance i = batch
// i / arr_n, position i % arr_n), the engine's workspace stays affine over the whole run.  One batch: arr_n = n_instances.
#define H2E_MAX_BATCHES 16
struct H2EInstTableArgs {
    u64 bfirst[5][H2E_MAX_BATCHES];   // per batch: address of its instance 0's base, range, select cells, inputs, status word
    u64 first[9];    // [5..8]: address of instance 0's hints, nd, jac, sel workspace ([0..4] unused: bfirst)
    u64 stride[9];   // bytes from one instance to the next (caller-side: inside a batch)
    u64* out;        // [n_instances][10 words]: the nine pointers, then {ws, hs} as two 32-bit words
    u32 ws;          // words between consecutive workspace value slots
    u32 hs;          // words between the two halves of a cell in the caller's arrays = 2 x instances per array
    u32 n_instances;
    u32 arr_n;
};
__global__ void __launch_bounds__(256) h2e_instance_table_k(H2EInstTableArgs a) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 i = t / 10u, w = t % 10u;
    if (i >= a.n_instances) return;
    u64 v;
    if (w < 5u) v = a.bfirst[w][i / a.arr_n] + (u64)(i % a.arr_n) * a.stride[w];
    else if (w < 9u) v = a.first[w] + (u64)i * a.stride[w];
    else v = (u64)a.ws | ((u64)a.hs << 32);
    a.out[(size_t)i * 10u + w] = v;
}
extern "C" int h2e_engine_instance_table(void* d_table, uint32_t n_instances, uint32_t arr_n, const uint64_t* bfirst5 /* [5][n_batches] */,
                                         const uint64_t* first9, const uint64_t* stride9, uint32_t ws, hipStream_t stream) {
    if (n_instances == 0) return 0;
    if (arr_n == 0 || n_instances % arr_n != 0 || n_instances / arr_n > H2E_MAX_BATCHES) return -1;
    const u32 nb = n_instances / arr_n;
    H2EInstTableArgs a;
    for (int k = 0; k < 9; k++) {
        a.first[k] = first9[k];
        a.stride[k] = stride9[k];
    }
    for (int w = 0; w < 5; w++)
        for (u32 b = 0; b < H2E_MAX_BATCHES; b++) a.bfirst[w][b] = b < nb ? bfirst5[(size_t)w * nb + b] : 0;
    a.out = (u64*)d_table;
    a.ws = ws;
    a.hs = 2u * arr_n;
    a.n_instances = n_instances;
    a.arr_n = arr_n;
    const u32 threads = n_instances * 10u;
    hipLaunchKernelGGL(h2e_instance_table_k, dim3((threads + 255) / 256), dim3(256), 0, stream, a);
    return (int)hipGetLastError();
}

// (debug builds: the device's constant-rate clock at this point of a stream - capi_common.hpp H2E_DEBUG_STAMPS)
__global__ void __launch_bounds__(64) h2e_stamp_k(u64* slot) {
    if (threadIdx.x == 0) *slot = wall_clock64();
}
extern "C" int h2e_engine_stamp(void* slot, hipStream_t stream) {
    hipLaunchKernelGGL(h2e_stamp_k, dim3(1), dim3(64), 0, stream, (u64*)slot);
    return (int)hipGetLastError();
}
