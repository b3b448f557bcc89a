// The C-ABI layer's common part: includes, the engine's entry points, error reporting, debug hooks, the instance table's host form.
// One translation unit: h2e_capi.cpp includes capi_common.hpp, program.hpp (+ the compiler passes), run.hpp and records_api.hpp in order.
#pragma once
// C ABI of the witness engine (include/h2e.h): program recording (host) + execution (HIP).
#include <hip/hip_runtime.h>
#include <functional>
#include <map>
#include <array>
#include <set>
#include <memory>
#include <chrono>
#include <mutex>
#include <atomic>
#include <string>
#include <unordered_map>
#include <cstring>
#include "../../include/h2e.h"
#include "recorder_pairing.hpp"
#include "field_chain.hpp"

extern "C" int h2e_engine_set_consts(int field_pair, const H2EFieldConsts* host);
extern "C" int h2e_engine_export(uint32_t cols, int columns, int mont, const void* in, void* out, const uint8_t* flags, uint64_t rows,
                                 uint32_t n_instances, const H2EFieldConsts* fc_dev, hipStream_t stream);
extern "C" int h2e_engine_digest(uint32_t cols, const void* in, const uint8_t* flags, uint64_t rows, uint32_t n_instances, void* out,
                                 hipStream_t stream);
extern "C" void h2e_engine_set_tuning(int key, int value);
extern "C" long long h2e_engine_scan_fallbacks(void);
extern "C" int h2e_engine_fixed(int field_pair, const uint32_t* ids, const uint64_t* dict, uint64_t rows, uint32_t cols, int columns, int mont,
                                const uint32_t* patches, uint32_t n_patches, const uint64_t* inputs, uint32_t n_slots, uint32_t slot_words,
                                uint32_t n_instances, const H2EFieldConsts* fc_dev, void* out, hipStream_t stream);
extern "C" int h2e_engine_range_table(int mont, const H2EFieldConsts* fc_dev, void* out, hipStream_t stream);
extern "C" int h2e_engine_patch_values(int field_pair, const uint32_t* patches, uint32_t n_patches, const uint64_t* inputs, uint32_t n_slots,
                                       uint32_t slot_words, uint32_t n_instances, const H2EFieldConsts* fc_dev, void* out, hipStream_t stream);
// checker.hip: the device-side constraint check (include/h2e.h h2e_check)
struct H2ECheckRegion {
    const void* adv;
    const uint8_t* flags;
    const uint32_t* fix;
    uint64_t rows, height;
};
extern "C" int h2e_engine_check_consts(const uint64_t n[4], uint64_t n_minv, const uint64_t r2[4]);
extern "C" int h2e_engine_check_to_mont(const uint64_t* in, uint64_t* out, uint64_t n, hipStream_t stream);
extern "C" int h2e_engine_check(const H2ECheckRegion* regs, const uint64_t* dict, const uint64_t* dict_m, const uint64_t* shifts_m,
                                const uint64_t* patch_vals, uint32_t n_patches, const uint64_t* sel_keys, const uint32_t* sel_key_rows,
                                uint32_t n_sel_keys, const uint32_t* perms, uint64_t n_pairs, uint32_t n_instances, uint32_t classes,
                                uint64_t* fail, hipStream_t stream);
extern "C" int h2e_engine_copy_constraints(const uint32_t* perms, uint64_t n, void* out, hipStream_t stream);
extern "C" int h2e_engine_or_status(const void* instances, uint32_t n_instances, uint32_t bits, hipStream_t stream);
extern "C" int h2e_engine_check_patch_values(const uint32_t* patches, uint32_t n_patches, const uint64_t* inputs, uint32_t n_slots, uint32_t slot_words,
                                             uint32_t n_instances, uint64_t* out, hipStream_t stream);   // checker.hip
extern "C" int h2e_engine_instance_table(void* d_table, uint32_t n_instances, uint32_t arr_n, const uint64_t* bfirst5, const uint64_t* first9, const uint64_t* stride9, uint32_t ws,
                                         hipStream_t stream);
extern "C" int h2e_engine_unit_records(const void* base, const void* status, const void* digests, void* out, const uint64_t* offsets3,
                                       const uint32_t* refs, uint32_t limbs, int has_point, uint32_t n_instances, uint32_t out_stride,
                                       hipStream_t stream);   // handoff.hip
extern "C" int h2e_engine_digest_reduce(const void* shards, uint32_t n_shards, uint32_t n_words, void* out, hipStream_t stream);
extern "C" int h2e_engine_gate(const uint32_t* counter, uint32_t target, hipStream_t stream);
// the column-emission units (engine.hip -DH2E_COLS, one per field pair): the expansion that stores halo2's advice columns itself
extern "C" int h2e_engine_set_consts_colsfp0(int field_pair, const H2EFieldConsts* host);
extern "C" int h2e_engine_set_consts_colsfp1(int field_pair, const H2EFieldConsts* host);
extern "C" int h2e_engine_set_consts_colsfp2(int field_pair, const H2EFieldConsts* host);
extern "C" int h2e_engine_launch_colsfp0(const H2ELaunch* launch, const void* instances, uint32_t n_instances, hipStream_t stream);
extern "C" int h2e_engine_launch_colsfp1(const H2ELaunch* launch, const void* instances, uint32_t n_instances, hipStream_t stream);
extern "C" int h2e_engine_launch_colsfp2(const H2ELaunch* launch, const void* instances, uint32_t n_instances, hipStream_t stream);
static int h2e_engine_launch_cols(const H2ELaunch* launch, const void* instances, uint32_t n_instances, hipStream_t stream) {
    switch (launch->field_pair) {
        case 0: return h2e_engine_launch_colsfp0(launch, instances, n_instances, stream);
        case 1: return h2e_engine_launch_colsfp1(launch, instances, n_instances, stream);
        case 2: return h2e_engine_launch_colsfp2(launch, instances, n_instances, stream);
        default: return -1;
    }
}
extern "C" int h2e_engine_launch(int field_pair, int mode, const H2ELaunch* launch, const void* instances,
                                 uint32_t n_instances, const H2EFieldConsts* fc_dev, hipStream_t stream);
extern "C" int h2e_engine_predict(int field_pair, int phase, const H2EPreKernel* k, const uint32_t* args_dev, const uint32_t* params_dev,
                                  const uint32_t* aux_dev, const void* instances, uint32_t n_instances,
                                  const H2EFieldConsts* fc_dev, hipStream_t stream);

namespace {

thread_local std::string g_last_error;
thread_local std::string g_last_warning;
int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) return fail(H2E_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// Debugging aids of the program compiler (tape dumps, switching compiler passes off).  Compiled out of the shipped
// library: build with -DH2E_DEBUG_HOOKS to get them back; the default build never reads the environment here.
#ifdef H2E_DEBUG_HOOKS
inline const char* dbg_env(const char* name) { return getenv(name); }
#else
inline const char* dbg_env(const char*) { return nullptr; }
#endif

// Ablation hooks of the scheduler (H2E_DEBUG_HOOKS builds only; exp/ablate.sh): H2E_DEBUG_SKIP is a bit mask of kernel
// classes run_impl leaves out - 1 inverse fix-ups, 2 finalize kernels, 4 MSM tail predictor, 8 MSM windows predictor,
// 16 select, 32 value replay of cut segments, 64 expansions.  The arrays of such a run are garbage: what it measures is
// what the class costs the step (its own time and what it takes from the kernels it runs beside).
#ifdef H2E_DEBUG_HOOKS
static uint32_t dbg_skip_mask() {
    static const uint32_t m = getenv("H2E_DEBUG_SKIP") ? (uint32_t)atoi(getenv("H2E_DEBUG_SKIP")) : 0u;
    return m;
}
// H2E_DEBUG_LOG=<file>: one line per engine call of run_impl - run number, segment, what, stream - so that a rocprofv3 kernel trace can
// be labelled by run and segment (exp/trace_labelled.py matches them in per-stream order)
static unsigned long long g_dbg_run = 0;
static int g_dbg_si = -1;
static FILE* dbg_log_file() {
    static FILE* f = getenv("H2E_DEBUG_LOG") ? fopen(getenv("H2E_DEBUG_LOG"), "a") : nullptr;
    return f;
}
static double dbg_host_us() {   // host clock of a log line: the gap to the next line is what the call in between cost the host
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static void dbg_log(const char* what, int a, unsigned b, unsigned c, unsigned d, hipStream_t st) {
    if (FILE* f = dbg_log_file()) {
        fprintf(f, "run %llu seg %d %s %d n_ops/kind %u strands/lanes %u n_sub %u stream %p host_us %.1f\n", g_dbg_run, g_dbg_si, what, a, b, c, d, (void*)st, dbg_host_us());
        fflush(f);
    }
}
// H2E_DEBUG_STAMPS=<file>: a DEVICE timeline that does not slow the host the way a profiler's kernel trace does (rocprofv3 costs a pipelined
// submission ~1 ms of host time: the trace of a small batch then shows the host, not the GPU).  One-lane kernels (handoff.hip
// h2e_engine_stamp) write the device's 100 MHz clock where the profiling events are recorded - per launched segment: 4 li + {0 chain begin,
// 1 chain end, 2 expansion begin, 3 expansion end}; 1000 = the run is complete, 1001 = its first kernel - and h2e_debug_dump_stamps()
// writes "run tag ticks" lines.
extern "C" int h2e_engine_stamp(void* slot, hipStream_t stream);
struct DbgStamps {
    uint64_t* d = nullptr;
    std::vector<std::pair<unsigned long long, unsigned>> labels;
    size_t cap = 1u << 18;
};
static DbgStamps g_dbg_stamps;
static void dbg_stamp(unsigned tag, hipStream_t st) {
    static const bool on = getenv("H2E_DEBUG_STAMPS") != nullptr;
    if (!on) return;
    DbgStamps& S = g_dbg_stamps;
    if (!S.d && hipMalloc((void**)&S.d, S.cap * 8) != hipSuccess) return;
    if (S.labels.size() >= S.cap) return;
    (void)h2e_engine_stamp(S.d + S.labels.size(), st);
    S.labels.emplace_back(g_dbg_run, tag);
}
extern "C" int h2e_debug_dump_stamps() {
    DbgStamps& S = g_dbg_stamps;
    const char* path = getenv("H2E_DEBUG_STAMPS");
    if (!path || !S.d) return 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    std::vector<uint64_t> h(S.labels.size());
    if (hipMemcpy(h.data(), S.d, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    FILE* f = fopen(path, "w");
    if (!f) return -1;
    for (size_t i = 0; i < h.size(); i++) fprintf(f, "%llu %u %llu\n", S.labels[i].first, S.labels[i].second, (unsigned long long)h[i]);
    fclose(f);
    return (int)h.size();
}
#define DBG_STAMP(tag, st) dbg_stamp((unsigned)(tag), st)
static void dbg_mark(const char* what) {
    if (FILE* f = dbg_log_file()) {
        fprintf(f, "run %llu mark %s host_us %.1f\n", g_dbg_run, what, dbg_host_us());
        fflush(f);
    }
}
static int dbg_engine_launch(int fpair, int mode, const H2ELaunch* l, const void* inst, uint32_t n, const H2EFieldConsts* fc, hipStream_t st) {
    uint32_t m = dbg_skip_mask();
    dbg_log("launch mode", mode, l->n_ops, l->n_strands, mode == 4 ? l->n_fixups : l->n_sub, st);
    if ((mode == 4 && (m & 1u)) || (mode == 1 && (m & 32u)) || (mode == 2 && l->n_sub > 1 && (m & 64u))) return 0;
    return h2e_engine_launch(fpair, mode, l, inst, n, fc, st);
}
static int dbg_engine_predict(int fpair, int phase, const H2EPreKernel* k, const uint32_t* a, const uint32_t* prm, const uint32_t* aux,
                              const void* inst, uint32_t n, const H2EFieldConsts* fc, hipStream_t st) {
    uint32_t m = dbg_skip_mask();
    dbg_log("predict phase", phase, k->kind, k->n_lanes, 0, st);
    if (m & 2u) phase &= ~2;
    if ((k->kind == H2E_PRE_MSM_TAIL && (m & 4u)) || (k->kind == H2E_PRE_MSM_WINDOWS && (m & 8u)) || (k->kind == H2E_PRE_MSM_SELECT && (m & 16u)))
        phase &= ~1;
    if (!phase) return 0;
    return h2e_engine_predict(fpair, phase, k, a, prm, aux, inst, n, fc, st);
}
#define H2E_LAUNCH dbg_engine_launch
#define H2E_PREDICT dbg_engine_predict
#else
#define H2E_LAUNCH h2e_engine_launch
#define H2E_PREDICT h2e_engine_predict
#define DBG_STAMP(tag, st) ((void)0)
#endif

const h2e::FieldPair& field_pair(int id) { return h2e::field_pair_of(id); }

struct InstanceDescHost {  // must match engine.hip InstanceDesc
    uint64_t* base;
    uint64_t* range;
    uint64_t* select;
    const uint64_t* inputs;
    uint32_t* status;
    uint64_t* hints;
    uint64_t* nd;
    uint64_t* jac;
    uint64_t* sel;
    uint32_t ws;     // words between consecutive workspace value slots = n_instances * words per slot (instance-minor)
    uint32_t hs;     // words between the two halves of a cell in the caller's arrays = 2 x instances per array (h2e_submit_batches: per batch)
};
static_assert(sizeof(InstanceDescHost) == 80, "handoff.hip h2e_instance_table_k writes the table as ten 64-bit words per instance");

}  // namespace

