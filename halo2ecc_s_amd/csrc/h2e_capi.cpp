// C ABI of the witness engine (include/h2e.h): program recording (host) + execution (HIP).
#include <hip/hip_runtime.h>
#include <functional>
#include <map>
#include <array>
#include <set>
#include <memory>
#include <mutex>
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
extern "C" int h2e_engine_unit_records(const void* base, const void* status, const void* digests, void* out, const uint64_t* offsets3,
                                       const uint32_t* refs, uint32_t limbs, int has_point, uint32_t n_instances, uint32_t out_stride,
                                       hipStream_t stream);   // handoff.hip
extern "C" int h2e_engine_digest_reduce(const void* shards, uint32_t n_shards, uint32_t n_words, void* out, hipStream_t stream);
extern "C" int h2e_engine_gate(const uint32_t* counter, uint32_t target, hipStream_t stream);
extern "C" int h2e_engine_launch(int field_pair, int mode, const H2ELaunch* launch, const void* instances,
                                 uint32_t n_instances, const H2EFieldConsts* fc_dev, hipStream_t stream);
extern "C" int h2e_engine_predict(int field_pair, int phase, const H2EPreKernel* k, const uint32_t* args_dev, const uint32_t* params_dev,
                                  const uint32_t* aux_dev, const void* instances, uint32_t n_instances,
                                  const H2EFieldConsts* fc_dev, hipStream_t stream);

namespace {

thread_local std::string g_last_error;
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
static void dbg_log(const char* what, int a, unsigned b, unsigned c, unsigned d, hipStream_t st) {
    if (FILE* f = dbg_log_file()) {
        fprintf(f, "run %llu seg %d %s %d n_ops/kind %u strands/lanes %u n_sub %u stream %p\n", g_dbg_run, g_dbg_si, what, a, b, c, d, (void*)st);
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
    uint32_t pad_;
};

}  // namespace

struct h2e_program {
    int field_pair;
    std::unique_ptr<h2e::Recorder> rec;
    uint64_t base_rows = 0, range_rows = 0, select_rows = 0;
    std::vector<uint32_t> perm_flat, patch_flat;
    // device copies (per device), created on first run
    int device = -1;
    H2EOp* d_tape = nullptr;
    uint32_t* d_aux = nullptr;
    uint64_t* d_pool = nullptr;
    uint32_t* d_params = nullptr;
    uint32_t* d_fixups = nullptr;
    uint32_t* d_pre_args = nullptr;
    uint32_t* d_subs = nullptr;
    std::vector<uint32_t> h_subs;          // per segment with cuts: [0, cut_1, ..., n_ops]
    std::vector<uint32_t> seg_sub_begin;   // per segment: index into h_subs (or ~0u)
    std::vector<uint32_t> seg_n_sub;
    // order tables of the packed expansion (tape.h H2ELaunch::pk_order), per cut segment and group count 2 << k
    std::vector<uint32_t> h_pk_order;
    std::vector<std::array<uint32_t, 5>> seg_pk_off, seg_pk_waves;
    uint32_t* d_pk_order = nullptr;
    bool pk_built = false;
    std::vector<uint8_t> seg_deferrable;   // a segment without cuts whose cells no later kernel reads: runs off the critical stream
    // compiled values-only replay (tape.h "V-tape"), per cut segment
    std::vector<H2EVRec> h_vtape;
    std::vector<uint32_t> seg_v_slots, seg_v_units;        // per segment: LDS sizing of the replay kernel
    std::vector<uint32_t> seg_piece_begin, seg_n_pieces;   // per segment: pieces = [first record, end record) pairs in h_vpieces
    std::vector<uint32_t> h_vpieces;
    H2EVRec* d_vtape = nullptr;
    uint32_t* d_vpieces = nullptr;
    // level-parallel replay (segments whose dependency graph is much shallower than it is long: the pairings)
    std::vector<H2EVRec> h_lrecs;               // 64 records per step (lane l of a step runs record 64 * step + l)
    std::vector<uint32_t> h_lrefs;              // cell refs of global integer operands
    std::vector<uint32_t> seg_l_begin, seg_l_steps, seg_l_slots, seg_l_pair;
    // hint store (field_chain.hpp): per segment with field hints, in place of a compiled replay
    std::vector<uint32_t> h_swords, h_soffsets, seg_s_begin, seg_so_begin, seg_sk_begin, seg_n_sops;
    std::vector<uint32_t> h_sext, seg_sx_begin;   // extension leaves of the store records (tape.h H2EStoreExt)
    uint32_t* d_sext = nullptr;
    std::vector<uint64_t> h_sktab;
    uint32_t *d_swords = nullptr, *d_soffsets = nullptr;
    uint64_t* d_sktab = nullptr;
    std::vector<uint32_t> h_lrounds;            // wave mode: per round (first record, count | kind << 8)
    std::vector<uint32_t> seg_lr_begin, seg_l_recs;
    H2EVRec* d_lrecs = nullptr;
    uint32_t* d_lrefs = nullptr;
    uint32_t* d_lrounds = nullptr;
    int64_t tail_from = -1;   // first segment of the program's serial tail (runs on the job slot's side stream), -1: none
    uint8_t* d_flags[3] = {nullptr, nullptr, nullptr};   // assigned / permute bytes on the device (h2e_export masks with them)
    // shape artefacts on the device (h2e_export_fixed / h2e_export_copy_constraints), uploaded on first use
    uint32_t* d_fix[3] = {nullptr, nullptr, nullptr};
    uint64_t* d_dict = nullptr;
    uint32_t* d_patches = nullptr;
    uint32_t* d_perms = nullptr;
    // h2e_check: base fixed ids with the cells made from instance inputs marked (bit 31 | patch index), the dictionary and the
    // range gates' shifts in Montgomery form, the select chip's table rows sorted by their encode cell, per-instance patch values
    uint32_t* d_fix_ck = nullptr;
    uint64_t *d_dict_m = nullptr, *d_shifts_m = nullptr, *d_sel_keys = nullptr, *d_patch_vals = nullptr;
    uint32_t* d_sel_key_rows = nullptr;
    uint32_t n_sel_keys = 0;
    size_t patch_vals_cap = 0;
    bool check_ready = false;

    ~h2e_program() {
        if (device >= 0) {
            (void)hipFree(d_tape);
            (void)hipFree(d_aux);
            (void)hipFree(d_pool);
            (void)hipFree(d_params);
            (void)hipFree(d_fixups);
            (void)hipFree(d_pre_args);
            (void)hipFree(d_subs);
            (void)hipFree(d_pk_order);
            (void)hipFree(d_vtape);
            (void)hipFree(d_vpieces);
            (void)hipFree(d_lrecs);
            (void)hipFree(d_lrefs);
            (void)hipFree(d_lrounds);
            (void)hipFree(d_swords);
            (void)hipFree(d_soffsets);
            (void)hipFree(d_sktab);
            (void)hipFree(d_sext);
            for (int i = 0; i < 3; i++) (void)hipFree(d_flags[i]);
        }
        for (int i = 0; i < 3; i++) (void)hipFree(d_fix[i]);
        (void)hipFree(d_dict);
        (void)hipFree(d_patches);
        (void)hipFree(d_perms);
        (void)hipFree(d_fix_ck);
        (void)hipFree(d_dict_m);
        (void)hipFree(d_shifts_m);
        (void)hipFree(d_sel_keys);
        (void)hipFree(d_patch_vals);
        (void)hipFree(d_sel_key_rows);
    }
    // Liveness over sub-ranges: an arithmetic op whose result cells are only read by ops of its own sub-range gets
    // H2E_FLAG_LOCAL_RESULT, so the values-only replay keeps that result in LDS and does not store it (the full
    // expansion of the sub-range recomputes and stores it anyway).  Row ownership: an op owns the rows from its
    // first row up to the next op's first row.  Every reference that can reach a cut segment is considered: op
    // refs of all segments, candidate tables (aux), strand parameters and the program's outputs.
    void mark_local_results() {
        h2e::Recorder& r = *rec;
        // a sub-range's integer results must all fit the replay's LDS ring (VCache::R in engine.hip)
        const uint32_t ring = r.fp.limbs == 3 ? 20 : 16;
        auto puts = [](const H2EOp& op) -> uint32_t {
            switch (op.opcode) {
                case H2E_OP_SELECT_POINT: return 2;
                case H2E_OP_INT_ADD: case H2E_OP_INT_SUB: case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: case H2E_OP_INT_MUL:
                case H2E_OP_REDUCE: case H2E_OP_DIV_CORE: case H2E_OP_MASK_INT: case H2E_OP_BISEC_INT: return 1;
                default: return 0;
            }
        };
        struct CutSeg {
            const h2e::Segment* sg;
            H2EOp* ops;
            uint32_t n_ops;
            std::vector<uint32_t> sub_of;
            std::vector<uint8_t> escapes, sub_fits;
            uint32_t first[3], last[3];
        };
        std::vector<CutSeg> cs;
        for (auto& sg : r.segments) {
            uint32_t n_ops = sg.tape_end - sg.tape_begin;
            if (sg.n_cuts == 0 || n_ops == 0) continue;
            CutSeg c;
            c.sg = &sg;
            c.ops = r.tape.data() + sg.tape_begin;
            c.n_ops = n_ops;
            std::vector<uint32_t> bounds;
            uint32_t lastb = 0;
            for (uint32_t k = 0; k < sg.n_cuts; k++) {
                uint32_t at = r.cuts[sg.cuts_begin + k];
                if (at > lastb && at < n_ops) {
                    bounds.push_back(at);
                    lastb = at;
                }
            }
            bounds.push_back(n_ops);
            bool any = false;
            uint32_t prev = 0;
            for (uint32_t bnd : bounds) {
                uint32_t np = 0, nsingle = 0;
                for (uint32_t i = prev; i < bnd; i++) {
                    np += puts(c.ops[i]);
                    uint16_t oc = c.ops[i].opcode;
                    if (oc == H2E_OP_IS_INT_ZERO || oc == H2E_OP_NOT || oc == H2E_OP_AND || oc == H2E_OP_OR || oc == H2E_OP_XNOR ||
                        oc == H2E_OP_PICK_INDEX)
                        nsingle++;
                }
                // too many results in a sub-range: all of its results stay stored
                c.sub_fits.push_back(np <= ring && nsingle <= 8);
                any = any || c.sub_fits.back();
                prev = bnd;
            }
            if (!any) continue;
            c.sub_of.assign(n_ops, 0);
            uint32_t sub = 0;
            for (uint32_t i = 0; i < n_ops; i++) {
                while (i >= bounds[sub]) sub++;
                c.sub_of[i] = sub;
            }
            c.escapes.assign(n_ops, 0);
            for (int reg = 0; reg < 3; reg++) {
                auto row_of = [&](uint32_t i) { return reg == 0 ? c.ops[i].base_row : reg == 1 ? c.ops[i].range_row : c.ops[i].select_row; };
                c.first[reg] = row_of(0);
                c.last[reg] = row_of(n_ops - 1) + 256;  // the last op writes < 256 rows
            }
            cs.push_back(std::move(c));
        }
        // producer of a row (strand-relative row for forks, absolute row for the main context) inside a cut segment
        auto producer = [&](const CutSeg& c, uint32_t region, uint32_t row) -> int {
            int lo = 0, hi = (int)c.n_ops - 1, ans = -1;
            while (lo <= hi) {
                int mid = (lo + hi) / 2;
                uint32_t first = region == 0 ? c.ops[mid].base_row : region == 1 ? c.ops[mid].range_row : c.ops[mid].select_row;
                if (first <= row) {
                    ans = mid;
                    lo = mid + 1;
                } else {
                    hi = mid - 1;
                }
            }
            return ans;
        };
        // Field hints (Recorder::begin_field_hints): a segment whose mul-like ops carry hints of the field-domain predictor
        // keeps them only if the predictor can be compiled for it (every op in the value cone of the hinted ops is one it
        // knows, every operand a result of the segment itself); otherwise the flags go and the segment replays as before.
        auto field_compiler = [&](const CutSeg& c) {
            h2e::FieldCompiler fcmp;
            fcmp.ops = c.ops;
            fcmp.n_ops = c.n_ops;
            fcmp.L = r.fp.limbs;
            fcmp.pw_check_limbs = r.fp.pure_w_check_limbs;
            fcmp.w_words = r.fp.w_words;
            fcmp.rel = c.sg->is_fork ? 1 : 0;
            for (int reg = 0; reg < 3; reg++) {
                fcmp.first[reg] = c.first[reg];
                fcmp.last[reg] = c.last[reg];
            }
            const CutSeg* cp = &c;
            fcmp.producer = [cp, &producer](uint32_t region, uint32_t row) { return producer(*cp, region, row); };
            // H2E_FIELD_CHAIN=lanes: the one-lane-per-record kernel (A/B); default: a 16-lane row per record, 60 rows per pass
            const char* fm = dbg_env("H2E_FIELD_CHAIN");   // (debug-hook builds linked with -DH2E_AB_KERNELS engine units only: the product has no lane kernel)
            fcmp.digit_rows = !(fm && !strcmp(fm, "lanes"));
            return fcmp;
        };
        // A context cut into several segments with field hints (a pairing check: Miller loop | final exponentiation ...): an
        // integer a segment reads from an EARLIER one is found here - the producing segment and op - and enters the reader's
        // programs as an import (field_chain.hpp FieldCompiler::Import).
        struct FieldProducer {
            const CutSeg* c = nullptr;
            int op = -1;
        };
        std::vector<h2e::FieldCompiler> seg_fcmp;   // per cut segment: the producer lookups (no program state)
        for (auto& c : cs) seg_fcmp.push_back(field_compiler(c));
        auto find_field_producer = [&](const CutSeg& self, uint32_t ref) -> FieldProducer {
            FieldProducer none;
            if (ref == H2E_NO_REF || H2E_REF_REGION(ref) > 1 || H2E_REF_REL(ref)) return none;
            uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
            for (size_t k = 0; k < cs.size(); k++) {
                const CutSeg& c = cs[k];
                if (&c == &self) break;   // (cs is in segment order: earlier segments only)
                if (!c.sg->field_hints || c.sg->is_fork || row < c.first[region] || row >= c.last[region]) continue;
                int q = seg_fcmp[k].int_producer(ref);
                if (q >= 0) {
                    FieldProducer fpd;
                    fpd.c = &c;
                    fpd.op = q;
                    return fpd;
                }
            }
            return none;
        };
        typedef std::map<const CutSeg*, std::map<uint32_t, uint32_t>> ExportMap;   // producing segment -> op -> hint slot
        // the import callbacks of segment `self`: slots from `slots` (a value not in there yet - the analysis pass - reads as slot 0),
        // every imported value that needs a slot noted in `wanted`
        auto wire_imports = [&](h2e::FieldCompiler& fcmp, const CutSeg& self, const ExportMap* slots, ExportMap* wanted) {
            const CutSeg* sp = &self;
            fcmp.import_of = [sp, slots, &find_field_producer](uint32_t ref, h2e::FieldCompiler::Import& imp) -> bool {
                FieldProducer fpd = find_field_producer(*sp, ref);
                if (!fpd.c) return false;
                const H2EOp& po = fpd.c->ops[fpd.op];
                if (po.opcode == H2E_OP_CONST_INT) { imp.kind = 1; imp.imm = po.imm; return true; }
                if (po.opcode == H2E_OP_ASSIGN_W || po.opcode == H2E_OP_CONST_INT_INPUT) {
                    if (po.flags & H2E_FLAG_INPUT_STRIDED) return false;
                    imp.kind = 2;
                    imp.imm = po.imm;
                    return true;
                }
                imp.kind = 0;
                imp.imm = 0;
                if (slots) {
                    auto it = slots->find(fpd.c);
                    if (it != slots->end()) {
                        auto jt = it->second.find((uint32_t)fpd.op);
                        if (jt != it->second.end()) imp.imm = jt->second;
                    }
                }
                return true;
            };
            fcmp.note_import = nullptr;
            if (wanted)
                fcmp.note_import = [sp, wanted, &find_field_producer](uint32_t ref) {
                    FieldProducer fpd = find_field_producer(*sp, ref);
                    if (!fpd.c) return;
                    uint16_t oc = fpd.c->ops[fpd.op].opcode;
                    if (oc == H2E_OP_CONST_INT || oc == H2E_OP_ASSIGN_W || oc == H2E_OP_CONST_INT_INPUT) return;
                    (*wanted)[fpd.c][(uint32_t)fpd.op] = 0;
                };
        };
        // all field-hint segments of the program, last to first: can each one's chain and store be compiled, given what the
        // later ones want exported?  (One verdict for the group: a segment that falls back to a replay stores no hint slots
        // for the others to import.)
        auto analyse_field_segments = [&](ExportMap& wanted, bool with_store, std::string& why) -> bool {
            for (size_t k = cs.size(); k-- > 0;) {
                const CutSeg& c = cs[k];
                const h2e::Segment& sg = *c.sg;
                if (!sg.field_hints) continue;
                if (!(sg.n_strands == 1 && !sg.is_fork && sg.field_pair == r.fp.id) || getenv("H2E_NO_FIELD_CHAIN")) { why = "not a single-strand segment of the program's field"; return false; }
                h2e::FieldCompiler fcmp = field_compiler(c);
                wire_imports(fcmp, c, nullptr, &wanted);
                h2e::StoreCompiler sc;
                sc.ops = c.ops;
                sc.n_ops = c.n_ops;
                sc.L = r.fp.limbs;
                sc.fc = &r.fp.fc;
                sc.fcmp = &fcmp;
                sc.next_aux = 0;
                h2e::HintStore hs;
                if (with_store) {
                    if (!sc.compile(hs)) { why = hs.why; return false; }
                    fcmp.aux = &hs.aux_hint;
                } else if (!sc.feasible(why)) return false;
                fcmp.exports = &wanted[&c];
                h2e::FieldChain chain;
                if (!fcmp.compile(chain, true)) { why = chain.why; return false; }
            }
            return true;
        };
        {
            bool any = false, all_cut = true;
            for (auto& sg : r.segments) {
                if (!sg.field_hints) continue;
                any = true;
                bool cut = false;
                for (auto& c : cs) cut = cut || c.sg == &sg;
                all_cut = all_cut && cut;
            }
            if (any) {
                ExportMap wanted;
                std::string why = "segment has no cuts";
                bool ok = all_cut;
                try {
                    ok = ok && analyse_field_segments(wanted, false, why);
                } catch (std::exception& e) {
                    ok = false;
                    why = e.what();
                }
                if (!ok)
                    for (auto& sg : r.segments) {
                        if (!sg.field_hints) continue;
                        for (uint32_t i = sg.tape_begin; i < sg.tape_end; i++) {
                            H2EOp& op = r.tape[i];
                            if ((op.flags & H2E_FLAG_HINTED) && !(op.flags & H2E_FLAG_HINT_STRIDED) &&
                                (op.opcode == H2E_OP_INT_MUL || op.opcode == H2E_OP_REDUCE || op.opcode == H2E_OP_DIV_CORE))
                                op.flags &= ~(uint16_t)H2E_FLAG_HINTED;
                        }
                        sg.field_hints = false;
                        if (dbg_env("H2E_DUMP_TAPE")) fprintf(stderr, "segment %zu: no field chain (%s)\n", (size_t)(&sg - r.segments.data()), why.c_str());
                    }
            }
        }
        if (cs.empty()) return;
        // an absolute reference from anywhere
        auto quote_abs = [&](uint32_t ref, const h2e::Segment* own = nullptr) {
            if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM || H2E_REF_REL(ref)) return;
            uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
            for (auto& c : cs) {
                const h2e::Segment& sg = *c.sg;
                if (&sg == own) continue;   // references inside the quoting segment itself are refined in step 1
                if (sg.is_fork) {
                    uint32_t b0 = region == 0 ? sg.base0 : region == 1 ? sg.range0 : sg.select0;
                    uint32_t d = region == 0 ? sg.dbase : region == 1 ? sg.drange : sg.dselect;
                    if (d == 0 || row < b0 || row >= b0 + (uint64_t)d * sg.n_strands) continue;
                    int pidx = producer(c, region, (row - b0) % d);
                    if (pidx >= 0) c.escapes[pidx] = 1;
                } else {
                    if (row < c.first[region] || row >= c.last[region]) continue;
                    int pidx = producer(c, region, row);
                    if (pidx >= 0) c.escapes[pidx] = 1;  // refined below for same-segment consumers
                }
            }
        };
        // 1. consumers inside the same segment (same addressing mode): only a different sub-range makes it escape
        for (auto& c : cs) {
            uint32_t rel = c.sg->is_fork ? 1 : 0;
            for (uint32_t i = 0; i < c.n_ops; i++)
                for (int k = 0; k < H2E_OP_MAX_REFS; k++) {
                    uint32_t ref = c.ops[i].refs[k];
                    if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM) continue;
                    if (H2E_REF_REL(ref) != rel) continue;
                    uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
                    if (!rel && (row < c.first[region] || row >= c.last[region])) continue;
                    int pidx = producer(c, region, row);
                    if (pidx >= 0 && c.sub_of[pidx] != c.sub_of[i]) c.escapes[pidx] = 1;
                }
        }
        // 2. every absolute reference from other places
        for (auto& sg : r.segments) {
            bool is_cut_main = false;
            for (auto& c : cs)
                if (c.sg == &sg && !sg.is_fork) is_cut_main = true;
            // a cut main-context segment: its references into itself were handled in step 1, those into other cut
            // segments (the MSM tail reads the windows' sums) count like anybody else's
            for (uint32_t i = sg.tape_begin; i < sg.tape_end; i++)
                for (int k = 0; k < H2E_OP_MAX_REFS; k++) quote_abs(r.tape[i].refs[k], is_cut_main ? &sg : nullptr);
        }
        for (uint32_t ref : r.aux) quote_abs(ref);
        for (uint32_t ref : r.params) quote_abs(ref);
        for (uint32_t ref : r.outputs) quote_abs(ref);
        for (uint32_t ref : r.pre_args) quote_abs(ref);  // (small integers in there never alias region/row of a cut segment)
        // 3. flag
        auto is_arithmetic = [](uint16_t oc) {
            return oc == H2E_OP_INT_ADD || oc == H2E_OP_INT_SUB || oc == H2E_OP_INT_NEG || oc == H2E_OP_INT_MUL_SMALL ||
                   oc == H2E_OP_INT_MUL || oc == H2E_OP_REDUCE || oc == H2E_OP_DIV_CORE || oc == H2E_OP_MASK_INT ||
                   oc == H2E_OP_IS_INT_ZERO || oc == H2E_OP_NOT;
        };
        for (auto& c : cs)
            for (uint32_t i = 0; i < c.n_ops; i++)
                if (is_arithmetic(c.ops[i].opcode) && !c.escapes[i] && c.sub_fits[c.sub_of[i]]) c.ops[i].flags |= H2E_FLAG_LOCAL_RESULT;
        // 4. dead ops of the values-only replay: a local result that no op the replay has to execute reads.  What an
        // op reads *in values mode* (exec_op_values): a hinted INT_MUL / REDUCE / DIV_CORE reads nothing.
        const int L = r.fp.limbs;
        for (auto& c : cs) {
            uint32_t rel = c.sg->is_fork ? 1 : 0;
            std::vector<uint8_t> used(c.n_ops, 0);
            for (uint32_t i = c.n_ops; i-- > 0;) {
                H2EOp& op = c.ops[i];
                bool needed = !(op.flags & H2E_FLAG_LOCAL_RESULT) || used[i];
                if (op.opcode == H2E_OP_PICK_INDEX && (op.flags & H2E_FLAG_PRESELECTED)) needed = false;   // the select pre-kernel did it
                if (!needed) {
                    op.flags |= H2E_FLAG_VALUES_SKIP;
                    // a value hint nobody consumes needs neither checking nor producing (the division's quotient is
                    // consumed by the expansion itself and keeps its hint)
                    if ((op.flags & H2E_FLAG_HINTED) && (op.opcode == H2E_OP_INT_MUL || op.opcode == H2E_OP_REDUCE))
                        op.flags &= ~(uint16_t)(H2E_FLAG_HINTED | H2E_FLAG_HINT_STRIDED);
                    continue;
                }
                uint32_t reads[H2E_OP_MAX_REFS];
                int nr = 0;
                bool hinted = (op.flags & H2E_FLAG_HINTED) != 0;
                switch (op.opcode) {
                    case H2E_OP_INT_MUL:
                        if (!hinted) { reads[nr++] = op.refs[0]; reads[nr++] = op.refs[L + 1]; }
                        break;
                    case H2E_OP_REDUCE:
                        if (!hinted) reads[nr++] = op.refs[0];
                        break;
                    case H2E_OP_DIV_CORE:
                        if (!hinted) { reads[nr++] = op.refs[0]; reads[nr++] = op.refs[L + 1]; }
                        break;
                    case H2E_OP_INT_ADD: case H2E_OP_INT_SUB:
                        reads[nr++] = op.refs[0]; reads[nr++] = op.refs[L + 1];
                        break;
                    case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: case H2E_OP_IS_INT_ZERO:
                        reads[nr++] = op.refs[0];
                        break;
                    case H2E_OP_MASK_INT:
                        reads[nr++] = op.refs[0]; reads[nr++] = op.refs[L + 1];
                        break;
                    case H2E_OP_ASSERT_CONST: case H2E_OP_CACHE_INT: case H2E_OP_SUM_LIMBS:
                        break;   // nothing in values mode
                    case H2E_OP_SELECT_POINT:
                        if (!(op.flags & H2E_FLAG_PRESELECTED)) reads[nr++] = op.refs[0];
                        break;
                    default:     // everything else: every reference
                        for (int k = 0; k < H2E_OP_MAX_REFS; k++) reads[nr++] = op.refs[k];
                        break;
                }
                for (int k = 0; k < nr; k++) {
                    uint32_t ref = reads[k];
                    if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM || H2E_REF_REL(ref) != rel) continue;
                    uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
                    if (!rel && (row < c.first[region] || row >= c.last[region])) continue;
                    int pidx = producer(c, region, row);
                    if (pidx >= 0 && (uint32_t)pidx < i) used[pidx] = 1;
                }
            }
        }
        // 5. compile the values-only replay of every cut segment
        seg_piece_begin.assign(r.segments.size(), 0);
        seg_n_pieces.assign(r.segments.size(), 0);
        seg_v_slots.assign(r.segments.size(), 1);
        seg_v_units.assign(r.segments.size(), 1);
        seg_l_begin.assign(r.segments.size(), 0);
        seg_l_steps.assign(r.segments.size(), 0);
        seg_l_slots.assign(r.segments.size(), 0);
        seg_l_pair.assign(r.segments.size(), 0);
        seg_lr_begin.assign(r.segments.size(), 0);
        seg_l_recs.assign(r.segments.size(), 0);
        seg_s_begin.assign(r.segments.size(), 0);
        seg_so_begin.assign(r.segments.size(), 0);
        seg_sk_begin.assign(r.segments.size(), 0);
        seg_n_sops.assign(r.segments.size(), 0);
        // A segment whose mul-like results all have hints from the MSM predictors (the windows' strands, the tail) needs no
        // chain to put its escaping values in place either: every one of them is a hint or a limb-wise combination of hints,
        // pre-selected candidates and integers that were stored before the segment started (field_chain.hpp "hint store",
        // records with extension leaves).  Whatever does not fit that description keeps its compiled replay.
        seg_sx_begin.assign(r.segments.size(), 0);
        auto compile_plain_store = [&](const CutSeg& c) -> bool {
            const h2e::Segment& sg = *c.sg;
            if (getenv("H2E_NO_PLAIN_STORE") || sg.field_pair != r.fp.id) return false;
            const int L = r.fp.limbs;
            const uint32_t rel = sg.is_fork ? 1 : 0;
            h2e::FieldCompiler fcmp = field_compiler(c);
            h2e::HintStore hs;
            h2e::StoreCompiler sc;
            sc.ops = c.ops;
            sc.n_ops = c.n_ops;
            sc.L = L;
            sc.fc = &r.fp.fc;
            sc.fcmp = &fcmp;
            sc.next_aux = 0;
            typedef h2e::StoreCompiler::Lin Lin;
            std::vector<uint32_t> ext;
            std::map<std::array<uint32_t, H2E_SX_WORDS>, uint32_t> ext_index;
            bool overflow = false;
            auto ext_leaf = [&](const std::array<uint32_t, H2E_SX_WORDS>& e) -> uint32_t {
                auto it = ext_index.find(e);
                uint32_t idx;
                if (it != ext_index.end()) idx = it->second;
                else {
                    idx = (uint32_t)(ext.size() / H2E_SX_WORDS);
                    ext.insert(ext.end(), e.begin(), e.end());
                    ext_index[e] = idx;
                }
                if (idx >= (1u << 22)) overflow = true;
                return (3u << 30) | (idx & 0x3fffffu);
            };
            auto small_leaf = [&](uint32_t kind, uint32_t index) -> uint32_t {
                if (index >= (1u << 22)) overflow = true;
                return (kind << 30) | (index & 0x3fffffu);
            };
            std::vector<Lin> lin(c.n_ops);
            std::vector<uint8_t> have(c.n_ops, 0);   // 1 done, 2 failed
            auto add_leaf = [](Lin& t, uint32_t leaf, int scale) {
                int& v = t.leaf[leaf];
                v += scale;
                if (v == 0) t.leaf.erase(leaf);
            };
            auto add_ceil = [](Lin& t, uint32_t times) {
                int& v = t.ceil[times];
                v += 1;
                if (v == 0) t.ceil.erase(times);
            };
            std::function<bool(uint32_t)> flatten = [&](uint32_t pi) -> bool {
                if (have[pi]) return have[pi] == 1;
                have[pi] = 2;
                const H2EOp& op = c.ops[pi];
                Lin rr;
                auto opd = [&](int refpos, int scale) -> bool {   // rr += scale * (the integer whose first limb cell is refs[refpos])
                    uint32_t ref = op.refs[refpos];
                    if (ref == H2E_NO_REF) return false;
                    uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
                    bool internal = region != H2E_REGION_PARAM && H2E_REF_REL(ref) == rel && (rel || (row >= c.first[region] && row < c.last[region]));
                    if (internal && region <= 1) {
                        int q = fcmp.int_producer(ref);
                        if (q < 0 || (uint32_t)q >= pi || !flatten((uint32_t)q)) return false;
                        h2e::StoreCompiler::add_scaled(rr, lin[q], scale);
                        return true;
                    }
                    if (internal) {   // select rows: a coordinate of the point the select pre-kernel picked
                        int q = producer(c, 2, row);
                        if (q < 0 || (uint32_t)q >= pi) return false;
                        const H2EOp& so = c.ops[q];
                        if (so.opcode != H2E_OP_SELECT_POINT || !(so.flags & H2E_FLAG_PRESELECTED) || ((so.flags >> 8) & 0xffu) != 0 || H2E_REF_COL(ref) != 0) return false;
                        uint32_t off = row - so.select_row;
                        if (off != 0 && off != (uint32_t)L + 1) return false;
                        std::array<uint32_t, H2E_SX_WORDS> e{};
                        e[0] = H2E_SX_SEL;
                        e[1] = so.refs[1];
                        e[2] = off ? 1u : 0u;
                        add_leaf(rr, ext_leaf(e), scale);
                        return true;
                    }
                    // an integer of another segment (or reached through the strand's parameters): its cells were stored before this
                    // segment's value chain started - the replay read them from there as well
                    std::array<uint32_t, H2E_SX_WORDS> e{};
                    e[0] = H2E_SX_CELLS;
                    for (int i = 0; i <= L; i++) e[1 + i] = op.refs[refpos + i];
                    add_leaf(rr, ext_leaf(e), scale);
                    return true;
                };
                bool ok = true;
                switch (op.opcode) {
                    case H2E_OP_INT_MUL: case H2E_OP_REDUCE: case H2E_OP_DIV_CORE:
                        if (!(op.flags & H2E_FLAG_HINTED)) { ok = false; break; }
                        if (op.flags & H2E_FLAG_HINT_STRIDED) {
                            std::array<uint32_t, H2E_SX_WORDS> e{};
                            e[0] = H2E_SX_HINT;
                            e[1] = op.imm;
                            add_leaf(rr, ext_leaf(e), 1);
                        } else add_leaf(rr, small_leaf(0, op.imm), 1);
                        break;
                    case H2E_OP_CONST_INT: add_leaf(rr, small_leaf(1, op.imm), 1); break;
                    case H2E_OP_ASSIGN_W: case H2E_OP_CONST_INT_INPUT:
                        if (op.flags & H2E_FLAG_INPUT_STRIDED) {
                            std::array<uint32_t, H2E_SX_WORDS> e{};
                            e[0] = H2E_SX_INPUT;
                            e[1] = op.imm;
                            add_leaf(rr, ext_leaf(e), 1);
                        } else add_leaf(rr, small_leaf(2, op.imm), 1);
                        break;
                    case H2E_OP_INT_ADD: ok = opd(0, 1) && opd(L + 1, 1); break;
                    case H2E_OP_INT_SUB:   // a - b + C_(b.times)   (integer_chip.rs:408-437)
                        ok = opd(0, 1) && opd(L + 1, -1);
                        add_ceil(rr, op.imm);
                        break;
                    case H2E_OP_INT_NEG:   // C_(a.times) - a       (:439-464)
                        ok = opd(0, -1);
                        add_ceil(rr, op.imm);
                        break;
                    case H2E_OP_INT_MUL_SMALL: ok = opd(0, (int)op.imm); break;
                    default: ok = false; break;
                }
                if (!ok) return false;
                lin[pi] = std::move(rr);
                have[pi] = 1;
                return true;
            };
            {
                Lin zero;
                sc.k_of(hs, zero);   // entry 0 of the K table
            }
            auto emit = [&](uint32_t kind, uint32_t k_idx, uint32_t w1, uint32_t w2, const std::vector<uint32_t>& terms) -> bool {
                if (terms.size() > 255 || k_idx > 0xffff) return false;
                hs.offsets.push_back((uint32_t)hs.words.size());
                hs.words.push_back(kind | ((uint32_t)terms.size() << 8) | (k_idx << 16));
                hs.words.push_back(w1);
                hs.words.push_back(w2);
                hs.words.insert(hs.words.end(), terms.begin(), terms.end());
                return true;
            };
            auto term = [](uint32_t leaf_word, int coef) -> uint32_t { return (leaf_word & 0xc0000000u) | ((uint32_t)(coef + 128) << 22) | (leaf_word & 0x3fffffu); };
            for (uint32_t i = 0; i < c.n_ops; i++) {
                const H2EOp& op = c.ops[i];
                if (op.flags & H2E_FLAG_VALUES_SKIP) continue;
                const bool stored = c.escapes[i] || !c.sub_fits[c.sub_of[i]];
                switch (op.opcode) {
                    case H2E_OP_NOP: case H2E_OP_ASSERT_CONST: case H2E_OP_SUM_LIMBS: case H2E_OP_CACHE_INT: break;   // nothing in values mode
                    case H2E_OP_PICK_INDEX:
                        if (!(op.flags & H2E_FLAG_PRESELECTED) && stored) return false;
                        break;
                    case H2E_OP_SELECT_POINT:
                        if (!(op.flags & H2E_FLAG_PRESELECTED) || stored) return false;
                        break;
                    case H2E_OP_ASSIGN_W: case H2E_OP_ASSIGN: case H2E_OP_ASSIGN_BIT: case H2E_OP_CONST: case H2E_OP_CONST_INT_INPUT:
                        if (!emit(H2E_S_FULL, 0, i, 0, {})) return false;   // ops without operands: run as they are
                        break;
                    case H2E_OP_CONST_INT:
                        if (stored && !emit(H2E_S_CONST, 0, op.base_row, 0, {term(small_leaf(1, op.imm), 1)})) return false;
                        break;
                    case H2E_OP_INT_MUL: case H2E_OP_REDUCE: case H2E_OP_DIV_CORE: {
                        if (!stored) break;
                        if (!flatten(i)) return false;
                        if (!emit(H2E_S_W, 0, op.base_row, op.range_row, {term(lin[i].leaf.begin()->first, 1)})) return false;
                    } break;
                    case H2E_OP_INT_ADD: case H2E_OP_INT_SUB: case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: {
                        if (!stored) break;
                        if (!flatten(i)) return false;
                        std::vector<uint32_t> terms;
                        long weight = 1;
                        for (auto& kv : lin[i].leaf) {
                            if (kv.second < -127 || kv.second > 127) return false;
                            terms.push_back(term(kv.first, kv.second));
                            weight += std::abs(kv.second);
                        }
                        if (weight >= 4096) return false;
                        if (!emit(H2E_S_LIN, sc.k_of(hs, lin[i]), op.base_row, 0, terms)) return false;
                    } break;
                    default:   // conditions, selections, decompositions ...: only if nothing outside their sub-range reads them
                        if (stored) return false;
                        break;
                }
            }
            if (overflow || hs.offsets.empty()) return false;
            size_t si = (size_t)(c.sg - r.segments.data());
            seg_s_begin[si] = (uint32_t)h_swords.size();
            seg_so_begin[si] = (uint32_t)h_soffsets.size();
            seg_sk_begin[si] = (uint32_t)h_sktab.size();
            seg_sx_begin[si] = (uint32_t)h_sext.size();
            seg_n_sops[si] = (uint32_t)hs.offsets.size();
            h_swords.insert(h_swords.end(), hs.words.begin(), hs.words.end());
            h_soffsets.insert(h_soffsets.end(), hs.offsets.begin(), hs.offsets.end());
            h_sktab.insert(h_sktab.end(), hs.ktab.begin(), hs.ktab.end());
            h_sext.insert(h_sext.end(), ext.begin(), ext.end());
            if (dbg_env("H2E_DUMP_TAPE"))
                fprintf(stderr, "segment %zu: plain hint store in place of the replay: %zu store ops per strand, %zu words, %zu extension leaves\n", si,
                        hs.offsets.size(), hs.words.size(), ext.size() / H2E_SX_WORDS);
            return true;
        };
        for (auto& c : cs)
            if (!c.sg->field_hints && !compile_plain_store(c))
                compile_replay(c.sg, c.ops, c.n_ops, c.first, c.last, [&](uint32_t region, uint32_t row) { return producer(c, region, row); });
        // 6. segments with field hints: the hint store in place of a replay, and the field-domain predictor whose program
        // goes into the pre-kernel args.  First (last segment to first) what each segment must leave in hint slots for the
        // later ones, then the programs themselves, first to last, every segment's compile-time slots (exports, conditions,
        // sink terms) in one block of their own.
        ExportMap wanted;
        {
            bool any = false;
            for (auto& c : cs) any = any || c.sg->field_hints;
            std::string why;
            if (any && !analyse_field_segments(wanted, true, why)) throw std::runtime_error("field chain: " + why);
        }
        ExportMap export_slots;
        for (auto& c : cs) {
            if (!c.sg->field_hints) continue;
            size_t si = (size_t)(c.sg - r.segments.data());
            const uint32_t hint_split = r.n_hint_slots;
            {
                std::map<uint32_t, uint32_t>& mine = export_slots[&c];
                for (auto& kv : wanted[&c]) {
                    const H2EOp& op = c.ops[kv.first];
                    bool own = (op.flags & H2E_FLAG_HINTED) && !(op.flags & H2E_FLAG_HINT_STRIDED) &&
                               (op.opcode == H2E_OP_INT_MUL || op.opcode == H2E_OP_REDUCE || op.opcode == H2E_OP_DIV_CORE);
                    mine[kv.first] = own ? op.imm : r.n_hint_slots++;
                }
            }
            h2e::FieldCompiler fcmp = field_compiler(c);
            wire_imports(fcmp, c, &export_slots, nullptr);
            h2e::HintStore hs;
            {
                h2e::StoreCompiler sc;
                sc.ops = c.ops;
                sc.n_ops = c.n_ops;
                sc.L = r.fp.limbs;
                sc.fc = &r.fp.fc;
                sc.fcmp = &fcmp;
                sc.next_aux = r.n_hint_slots;
                if (!sc.compile(hs)) throw std::runtime_error(hs.why);
                r.n_hint_slots = sc.next_aux;
                seg_s_begin[si] = (uint32_t)h_swords.size();
                seg_so_begin[si] = (uint32_t)h_soffsets.size();
                seg_sk_begin[si] = (uint32_t)h_sktab.size();
                seg_sx_begin[si] = (uint32_t)h_sext.size();
                seg_n_sops[si] = (uint32_t)hs.offsets.size();
                h_swords.insert(h_swords.end(), hs.words.begin(), hs.words.end());
                h_soffsets.insert(h_soffsets.end(), hs.offsets.begin(), hs.offsets.end());
                h_sktab.insert(h_sktab.end(), hs.ktab.begin(), hs.ktab.end());
                h_sext.insert(h_sext.end(), hs.ext.begin(), hs.ext.end());
                if (dbg_env("H2E_DUMP_TAPE"))
                    fprintf(stderr, "segment %zu: hint store: %zu store ops, %zu words, %zu K constants, at most %u terms, %zu aux hint slots\n", si,
                            hs.offsets.size(), hs.words.size(), hs.ktab.size() / (2 * (size_t)r.fp.limbs + 4), hs.n_terms_max, hs.aux_hint.size());
            }
            fcmp.aux = &hs.aux_hint;
            fcmp.exports = &export_slots[&c];
            fcmp.hint_split = hint_split;
            uint32_t next_hint = r.n_hint_slots;
            fcmp.next_hint = &next_hint;
            h2e::FieldChain chain;
            if (!fcmp.compile(chain, false)) throw std::runtime_error("field chain: " + chain.why);
            r.n_hint_slots = next_hint;
            h2e::PreKernel pk;
            std::memset(&pk.k, 0, sizeof(pk.k));
            pk.k.kind = H2E_PRE_FIELD_CHAIN;
            pk.k.n_lanes = 1;
            pk.k.hint_base = chain.hint_hi > chain.hint_lo ? chain.hint_lo : 0;
            pk.k.hints_per_lane = chain.hint_hi > chain.hint_lo ? chain.hint_hi - chain.hint_lo : 0;
            pk.k.hint2_base = chain.hint2_hi > chain.hint2_lo ? chain.hint2_lo : 0;
            pk.k.hints2_per_lane = chain.hint2_hi > chain.hint2_lo ? chain.hint2_hi - chain.hint2_lo : 0;
            pk.k.n_params = (uint32_t)r.fp.w_words;   // words per input slot
            while (r.pre_args.size() % 16) r.pre_args.push_back(0);   // records are read 16 bytes at a time from 64-byte aligned chunks
            pk.k.f_recs = (uint32_t)r.pre_args.size();
            pk.k.f_n_recs = (uint32_t)(chain.recs.size() / chain.rec_words);
            r.pre_args.insert(r.pre_args.end(), chain.recs.begin(), chain.recs.end());
            pk.k.f_rounds = (uint32_t)r.pre_args.size();
            pk.k.f_n_rounds = (uint32_t)(chain.rounds.size() / 2);
            r.pre_args.insert(r.pre_args.end(), chain.rounds.begin(), chain.rounds.end());
            pk.k.f_slots = chain.n_slots;
            pk.k.f_n_load_rounds = chain.n_load_rounds;
            pk.k.f_mode = fcmp.digit_rows ? 1 : 0;
            pk.k.f_sinks = (uint32_t)r.pre_args.size();
            r.pre_args.insert(r.pre_args.end(), chain.sink_offsets.begin(), chain.sink_offsets.end());
            pk.k.f_sink_words = (uint32_t)r.pre_args.size();
            r.pre_args.insert(r.pre_args.end(), chain.sink_words.begin(), chain.sink_words.end());
            pk.k.f_n_sinks = (uint32_t)chain.sink_offsets.size();
            pk.before_segment = (uint32_t)(c.sg - r.segments.data());
            pk.early_after_segment = -1;
            r.pre_kernels.push_back(pk);
            if (dbg_env("H2E_DUMP_TAPE") || getenv("H2E_FIELD_STATS"))
                fprintf(stderr, "segment %u: field chain: %u nodes (%u products, %u linear combinations), %u rounds, %u value slots, hint slots [%u, %u) + [%u, %u), %zu exports\n",
                        pk.before_segment, chain.n_nodes, chain.n_mul, chain.n_lin, pk.k.f_n_rounds, chain.n_slots, chain.hint_lo, chain.hint_hi,
                        chain.hint2_lo, chain.hint2_hi, export_slots[&c].size());
        }
    }

    // Compile one cut segment into V-tape records (tape.h).  Values = results of alive ops; each gets an LDS slot for
    // as long as later ops of the replay read it (furthest-next-use eviction when the slots run out: an evicted or
    // never cached value goes through its cells, so its producer stores it).
    void compile_replay(const h2e::Segment* sg, const H2EOp* ops, uint32_t n_ops, const uint32_t* first, const uint32_t* last,
                        const std::function<int(uint32_t, uint32_t)>& producer) {
        h2e::Recorder& r = *rec;
        const int L = r.fp.limbs;
        const uint32_t NS = L == 3 ? 22 : 18, NF = 4;   // VSlots in engine.hip (NF) and the LDS budget (NS)
        const uint32_t rel = sg->is_fork ? 1 : 0;
        enum { K_NONE, K_MUL, K_ADD, K_FE, K_SEL, K_FULL, K_CONST };
        auto kind_of = [](const H2EOp& op) -> int {
            switch (op.opcode) {
                case H2E_OP_INT_MUL: case H2E_OP_REDUCE: case H2E_OP_DIV_CORE: return K_MUL;
                case H2E_OP_INT_ADD: case H2E_OP_INT_SUB: case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: case H2E_OP_MASK_INT:
                case H2E_OP_BISEC_INT: return K_ADD;
                case H2E_OP_IS_INT_ZERO: case H2E_OP_NOT: case H2E_OP_AND: case H2E_OP_OR: case H2E_OP_XNOR: case H2E_OP_PICK_INDEX:
                    return K_FE;
                case H2E_OP_SELECT_POINT: return K_SEL;
                case H2E_OP_CONST_INT: return K_CONST;   // a constant of the pool: a value like any other (cells: column 0)
                case H2E_OP_ASSERT_CONST: case H2E_OP_CACHE_INT: case H2E_OP_SUM_LIMBS: case H2E_OP_NOP: return K_NONE;
                default: return K_FULL;
            }
        };
        auto fe_row = [&](const H2EOp& op) -> uint32_t {
            if (op.opcode == H2E_OP_IS_INT_ZERO) return op.base_row + 6 + 4 * (uint32_t)r.fp.pure_w_check_limbs;
            if (op.opcode == H2E_OP_PICK_INDEX) return op.base_row + (op.imm < 5 ? 0 : 1);
            return op.base_row;
        };
        struct Opd { uint32_t ref; bool is_int; int refpos; };
        auto operands = [&](const H2EOp& op, Opd* o) -> int {
            bool hinted = (op.flags & H2E_FLAG_HINTED) != 0;
            int n = 0;
            switch (op.opcode) {
                case H2E_OP_INT_MUL: case H2E_OP_DIV_CORE:
                    if (!hinted) { o[n++] = {op.refs[0], true, 0}; o[n++] = {op.refs[L + 1], true, L + 1}; }
                    break;
                case H2E_OP_REDUCE:
                    if (!hinted) o[n++] = {op.refs[0], true, 0};
                    break;
                case H2E_OP_INT_ADD: case H2E_OP_INT_SUB:
                    o[n++] = {op.refs[0], true, 0}; o[n++] = {op.refs[L + 1], true, L + 1};
                    break;
                case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: case H2E_OP_IS_INT_ZERO:
                    o[n++] = {op.refs[0], true, 0};
                    break;
                case H2E_OP_MASK_INT:
                    o[n++] = {op.refs[0], true, 0}; o[n++] = {op.refs[L + 1], false, L + 1};
                    break;
                case H2E_OP_BISEC_INT:
                    o[n++] = {op.refs[0], false, 0}; o[n++] = {op.refs[1], true, 1}; o[n++] = {op.refs[L + 2], true, L + 2};
                    break;
                case H2E_OP_SELECT_POINT:
                    if (!(op.flags & H2E_FLAG_PRESELECTED)) o[n++] = {op.refs[0], false, 0};
                    break;
                case H2E_OP_NOT:
                    o[n++] = {op.refs[0], false, 0};
                    break;
                case H2E_OP_AND: case H2E_OP_OR: case H2E_OP_XNOR:
                    o[n++] = {op.refs[0], false, 0}; o[n++] = {op.refs[1], false, 1};
                    break;
                default: break;
            }
            return n;
        };
        // value id = 2 * op + which; -1: not a value of this replay (read from its cell)
        auto value_of = [&](uint32_t ref, bool is_int) -> int {
            if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM || H2E_REF_REL(ref) != rel) return -1;
            uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref), col = H2E_REF_COL(ref);
            if (!rel && (row < first[region] || row >= last[region])) return -1;
            int p = producer(region, row);
            if (p < 0) return -1;
            const H2EOp& po = ops[p];
            int pk = kind_of(po);
            if (pk == K_FULL) return -1;   // its rows are written for real
            if (is_int) {
                if (pk == K_MUL && region == 1 && col == 0 && row == po.range_row) return 2 * p;
                if (pk == K_ADD && region == 0 && col == 4 && row == po.base_row) return 2 * p;
                if (pk == K_CONST && region == 0 && col == 0 && row == po.base_row) return 2 * p;
                if (pk == K_SEL && region == 2 && col == 0 && row == po.select_row) return 2 * p;
                if (pk == K_SEL && region == 2 && col == 0 && row == po.select_row + (uint32_t)L + 1) return 2 * p + 1;
            } else {
                if (pk == K_FE && region == 0 && col == 4 && row == fe_row(po)) return 2 * p;
            }
            return -2;   // a cell the values-only replay never writes
        };
        // the op of this segment (index into ops) that writes a referenced cell, or -1
        auto writer_of = [&](uint32_t ref) -> int {
            if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM || H2E_REF_REL(ref) != rel) return -1;
            uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
            if (!rel && (row < first[region] || row >= last[region])) return -1;
            return producer(region, row);
        };
        struct Val {
            std::vector<uint32_t> uses;   // positions (alive order) of the consumers that can read a slot
            size_t next = 0;
            int slot = -1;
            bool resident = false, force_store = false;
            int dst_slot = -1;                    // the slot it was given when produced
            bool evicted = false;                 // lost its slot before its last use
            uint32_t cell_use_last = 0xffffffffu; // last alive position of an op that reads its *cell* (V_FULL, PICK_INDEX)
        };
        std::vector<Val> vals(2 * (size_t)n_ops);
        std::vector<uint32_t> alive;
        for (uint32_t i = 0; i < n_ops; i++)
            if (!(ops[i].flags & H2E_FLAG_VALUES_SKIP) && kind_of(ops[i]) != K_NONE) alive.push_back(i);
        // pass 1: uses
        std::vector<uint32_t> full_read_last(n_ops, 0);   // per op written "for real" (V_FULL): last alive position reading its rows
        std::vector<uint32_t> alive_pos(n_ops, 0xffffffffu);
        for (uint32_t pos = 0; pos < alive.size(); pos++) alive_pos[alive[pos]] = pos;
        for (uint32_t pos = 0; pos < alive.size(); pos++) {
            const H2EOp& op = ops[alive[pos]];
            int k = kind_of(op);
            for (int q = 0; q < H2E_OP_MAX_REFS; q++) {
                int wtr = writer_of(op.refs[q]);
                if (wtr >= 0 && (uint32_t)wtr != alive[pos] && kind_of(ops[wtr]) == K_FULL) full_read_last[wtr] = std::max(full_read_last[wtr], pos);
            }
            if (k == K_FULL || op.opcode == H2E_OP_PICK_INDEX) {   // reads cells: whatever it reads must be stored
                for (int q = 0; q < H2E_OP_MAX_REFS; q++)
                    for (int as_int = 0; as_int < 2; as_int++) {
                        int v = value_of(op.refs[q], as_int != 0);
                        if (v >= 0) {
                            vals[v].force_store = true;
                            vals[v].cell_use_last = pos;
                        }
                    }
                continue;
            }
            Opd o[3];
            int n = operands(op, o);
            for (int q = 0; q < n; q++) {
                int v = value_of(o[q].ref, o[q].is_int);
                if (v == -2) throw std::runtime_error("replay compile: operand is not a replay result");
                if (v >= 0) {
                    if (ops[v / 2].flags & H2E_FLAG_VALUES_SKIP) throw std::runtime_error("replay compile: live operand of a dead op");
                    vals[v].uses.push_back(pos);
                }
            }
        }
        // pass 2: slot allocation
        struct Dec {
            uint8_t kind[3] = {0, 0, 0};
            uint32_t word[3] = {0, 0, 0};
            int val[3] = {-1, -1, -1};
            int dst[2] = {-1, -1};
        };
        std::vector<Dec> dec(alive.size());
        std::vector<int> int_owner(NS, -1), fe_owner(NF, -1);
        auto next_use = [&](int v) -> uint32_t { return vals[v].next < vals[v].uses.size() ? vals[v].uses[vals[v].next] : 0xffffffffu; };
        auto take_slot = [&](std::vector<int>& owner, int v) -> int {
            for (size_t sl = 0; sl < owner.size(); sl++)
                if (owner[sl] < 0) {
                    owner[sl] = v;
                    return (int)sl;
                }
            size_t victim = 0;
            for (size_t sl = 1; sl < owner.size(); sl++)
                if (next_use(owner[sl]) > next_use(owner[victim])) victim = sl;
            if (next_use(owner[victim]) <= next_use(v)) return -1;   // the new value is the one needed last
            Val& ev = vals[owner[victim]];
            ev.resident = false;
            ev.force_store = true;
            ev.evicted = true;
            ev.slot = -1;
            owner[victim] = v;
            return (int)victim;
        };
        for (uint32_t pos = 0; pos < alive.size(); pos++) {
            uint32_t i = alive[pos];
            const H2EOp& op = ops[i];
            int k = kind_of(op);
            Dec& d = dec[pos];
            if (k != K_FULL && op.opcode != H2E_OP_PICK_INDEX) {
                Opd o[3];
                int n = operands(op, o);
                for (int q = 0; q < n; q++) {
                    int v = value_of(o[q].ref, o[q].is_int);
                    d.val[q] = v;
                    if (v >= 0 && vals[v].resident) {
                        d.kind[q] = o[q].is_int ? H2E_VSRC_INT_SLOT : H2E_VSRC_FE_SLOT;
                        d.word[q] = (uint32_t)vals[v].slot;
                    } else {
                        d.kind[q] = H2E_VSRC_GLOBAL;
                        if (v >= 0) {
                            vals[v].force_store = true;
                            vals[v].cell_use_last = pos;   // read through its cell
                        }
                    }
                }
                for (int q = 0; q < n; q++) {
                    int v = d.val[q];
                    if (v < 0) continue;
                    if (vals[v].next < vals[v].uses.size() && vals[v].uses[vals[v].next] == pos) vals[v].next++;
                }
                for (int q = 0; q < n; q++) {
                    int v = d.val[q];
                    if (v < 0 || !vals[v].resident) continue;
                    if (vals[v].next >= vals[v].uses.size()) {   // last use: free the slot
                        auto& owner = (kind_of(ops[v / 2]) == K_FE) ? fe_owner : int_owner;
                        owner[vals[v].slot] = -1;
                        vals[v].resident = false;
                    }
                }
            }
            int nres = k == K_SEL ? 2 : (k == K_MUL || k == K_ADD || k == K_FE || k == K_CONST) ? 1 : 0;
            for (int w = 0; w < nres; w++) {
                int v = 2 * (int)i + w;
                if (vals[v].uses.empty()) continue;
                int sl = take_slot(k == K_FE ? fe_owner : int_owner, v);
                if (sl < 0) {
                    vals[v].force_store = true;
                } else {
                    vals[v].slot = sl;
                    vals[v].dst_slot = sl;
                    vals[v].resident = true;
                    d.dst[w] = sl;
                }
            }
        }
        // ---- level-parallel replay ---------------------------------------------------------------------------------
        // A pairing's replay is 175 k ops in one chain, but its dependency graph is only ~8.5 k levels deep (an Fq12
        // product is 54 independent Fq products).  When a segment is that shape, lanes are given to *ops*: a wave replays
        // one instance, each step runs up to 64 independent ops of one opcode, values live in LDS slots shared by the
        // wave (allocated over the step order).  Ops that go through cells (H2E_V_FULL: assign / constants / bisec rows)
        // are steps of their own behind a fence.
        {
            size_t si = (size_t)(sg - r.segments.data());
            bool eligible = !dbg_env("H2E_NO_LEVELS") && sg->n_strands == 1 && alive.size() >= 4096;
            std::vector<uint32_t> level(alive.size(), 0);
            uint32_t depth = 0;
            for (uint32_t pos = 0; pos < alive.size() && eligible; pos++) {
                const H2EOp& op = ops[alive[pos]];
                int k = kind_of(op);
                // (hinted chains are wide, not deep - the MSM tail would need > 1664 value slots - and have pieces instead)
                if (k == K_SEL || op.opcode == H2E_OP_PICK_INDEX || ((op.flags & H2E_FLAG_HINTED) && k == K_MUL)) eligible = false;
                uint32_t lv = 0;
                for (int q = 0; q < 3; q++)
                    if (dec[pos].val[q] >= 0) lv = std::max(lv, level[alive_pos[dec[pos].val[q] / 2]] + 1);
                for (int q = 0; q < H2E_OP_MAX_REFS; q++) {
                    int wtr = writer_of(op.refs[q]);   // rows an earlier op of this replay writes for real
                    if (wtr >= 0 && (uint32_t)wtr != alive[pos] && alive_pos[wtr] != 0xffffffffu && alive_pos[wtr] < pos)
                        lv = std::max(lv, level[alive_pos[wtr]] + 1);
                }
                level[pos] = lv;
                depth = std::max(depth, lv + 1);
            }
            if (eligible && (uint64_t)depth * 4 > alive.size()) eligible = false;   // not shallow enough to pay off
            if (eligible) {
                // as late as possible: an op runs just before its first consumer (values stay in slots for a short time: a
                // pairing's G2 line coefficients are then made next to the Miller-loop step that uses them); ops nothing
                // in the replay depends on run as early as they can, which frees their operands
                std::vector<uint32_t> late(alive.size(), 0xffffffffu);
                for (uint32_t pos = (uint32_t)alive.size(); pos-- > 0;) {
                    uint32_t lv = late[pos] == 0xffffffffu ? level[pos] : late[pos] - 1;
                    if (lv < level[pos]) throw std::runtime_error("replay compile: level order broken");
                    level[pos] = lv;
                    const H2EOp& op = ops[alive[pos]];
                    for (int q = 0; q < 3; q++)
                        if (dec[pos].val[q] >= 0) {
                            uint32_t pp = alive_pos[dec[pos].val[q] / 2];
                            late[pp] = std::min(late[pp], lv);
                        }
                    for (int q = 0; q < H2E_OP_MAX_REFS; q++) {
                        int wtr = writer_of(op.refs[q]);
                        if (wtr >= 0 && (uint32_t)wtr != alive[pos] && alive_pos[wtr] != 0xffffffffu && alive_pos[wtr] < pos)
                            late[alive_pos[wtr]] = std::min(late[alive_pos[wtr]], lv);
                    }
                }
            }
            if (eligible) {
                // steps: by level, then by opcode; V_FULL ops one per step
                std::vector<uint32_t> order(alive.size());
                for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
                auto vop_of = [&](uint32_t pos) -> uint32_t {
                    const H2EOp& op = ops[alive[pos]];
                    if ((op.flags & H2E_FLAG_HINTED) && kind_of(op) == K_MUL) return H2E_V_HINT;
                    switch (op.opcode) {
                        case H2E_OP_INT_MUL: return H2E_V_MUL;
                        case H2E_OP_REDUCE: return H2E_V_REDUCE;
                        case H2E_OP_DIV_CORE: return H2E_V_DIV;
                        case H2E_OP_INT_ADD: return H2E_V_ADD;
                        case H2E_OP_INT_SUB: return H2E_V_SUB;
                        case H2E_OP_INT_NEG: return H2E_V_NEG;
                        case H2E_OP_INT_MUL_SMALL: return H2E_V_MUL_SMALL;
                        case H2E_OP_MASK_INT: return H2E_V_MASK;
                        case H2E_OP_BISEC_INT: return H2E_V_BISEC_INT;
                        case H2E_OP_IS_INT_ZERO: return H2E_V_IS_ZERO;
                        case H2E_OP_NOT: return H2E_V_NOT;
                        case H2E_OP_AND: return H2E_V_AND;
                        case H2E_OP_OR: return H2E_V_OR;
                        case H2E_OP_XNOR: return H2E_V_XNOR;
                        case H2E_OP_CONST_INT: return H2E_V_CONST;
                        default: return H2E_V_FULL;
                    }
                };
                std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
                    if (level[a] != level[b]) return level[a] < level[b];
                    return vop_of(a) < vop_of(b);
                });
                size_t NW = H2E_LEVEL_WAVES;   // steps (waves) per round; 1 in wave mode
                std::vector<uint32_t> step_of(alive.size(), 0);   // the *round* an op runs in (see below)
                std::vector<std::vector<uint32_t>> steps;            // steps[NW * round + wave]: the ops one wave runs in a round
                std::vector<int> lslot;
                int n_slots = 0;
                size_t n_rounds = 0;
                int slot_cap = 0;
                // Two instances per workgroup when their value slots fit side by side in a CU's LDS: a step then holds up to
                // 32 ops, lanes 0-31 run them for one instance and lanes 32-63 for the other (levels are ~20 ops wide on
                // average: one instance leaves two thirds of every wave instruction idle).  Otherwise one instance per
                // workgroup and steps of 64.
                bool paired = false;
                std::function<bool(int)> alloc_slots;
                auto schedule = [&](size_t step_ops, int cap) -> bool {
                steps.clear();
                std::fill(step_of.begin(), step_of.end(), 0u);
                // rounds: H2E_LEVEL_WAVES waves share an instance's value slots; in a round each wave runs one step (up to 64
                // ops of one opcode), all steps of a round come from the same level, a barrier separates rounds.  An op
                // that goes through cells (V_FULL) is a round of its own.
                for (size_t i = 0; i < order.size();) {
                    uint32_t lv0 = level[order[i]];
                    std::vector<std::vector<uint32_t>> lvl_steps;
                    std::vector<uint32_t> lvl_full;
                    while (i < order.size() && level[order[i]] == lv0) {
                        uint32_t pos = order[i], vop = vop_of(pos);
                        if (vop == H2E_V_FULL) {
                            lvl_full.push_back(pos);
                            i++;
                            continue;
                        }
                        size_t j = i + 1;
                        while (j < order.size() && j - i < step_ops && level[order[j]] == lv0 && vop_of(order[j]) == vop) j++;
                        lvl_steps.emplace_back(order.begin() + i, order.begin() + j);
                        i = j;
                    }
                    auto new_round = [&]() {
                        for (size_t w = 0; w < NW; w++) steps.emplace_back();
                        return steps.size() / NW - 1;
                    };
                    for (uint32_t pos : lvl_full) {
                        size_t rd = new_round();
                        steps[rd * NW].push_back(pos);
                        step_of[pos] = (uint32_t)rd;
                    }
                    for (size_t k = 0; k < lvl_steps.size(); k++) {
                        if (k % NW == 0) new_round();
                        size_t rd = steps.size() / NW - 1;
                        // Most rounds hold a single step.  Wave w of a workgroup sits on SIMD w of its CU: if that step always
                        // went to wave 0, SIMD 0 would carry the chains of every instance on the CU and the others idle -
                        // the steps rotate over the waves from round to round.
                        steps[rd * NW + (k % NW + rd) % NW] = lvl_steps[k];
                        for (uint32_t pos : lvl_steps[k]) step_of[pos] = (uint32_t)rd;
                    }
                }
                return alloc_slots(cap);
                };   // schedule
                // ---- cost-class rounds (default) ---------------------------------------------------------------------
                // Rounds by level put a product into 62 % of the rounds of a pairing check although its multiplicative depth
                // is a tenth of its depth: an Fq12 product is one level of int_mul between a dozen levels of int_add / int_sub /
                // reduce, and ALAP levels scatter the products of independent branches over all of them.  A round costs what
                // its most expensive lane costs (a product ~2.5 us, a reduce ~1 us, an addition ~0.5 us), so the rounds are
                // built by cost class instead: walking the dependency graph from the results backwards (as late as possible:
                // a value is made just before its first use, which keeps the value slots few), a round takes every ready op of
                // the *cheapest* class that has ready ops - expensive ops wait until nothing cheaper can go, and so meet the
                // products of the other branches in one round.  Light ops of different opcodes share a step (the kernel
                // dispatches them per lane: H2E_VFLAG_MIXED); products, reduces and divisions keep one opcode per step.
                auto cls_of = [&](uint32_t pos) -> int {   // 0 light, 1 medium, 2 heavy, 3 through cells
                    switch (vop_of(pos)) {
                        case H2E_V_FULL: return 3;
                        case H2E_V_MUL: case H2E_V_DIV: return 2;
                        case H2E_V_REDUCE: case H2E_V_MUL_SMALL: case H2E_V_CONST: case H2E_V_HINT: return 1;
                        default: return 0;
                    }
                };
                std::vector<std::vector<uint32_t>> preds(alive.size()), succs(alive.size());
                auto build_deps = [&]() {
                    if (!succs.empty() && !preds.empty() && (!preds[alive.size() - 1].empty() || !succs[0].empty())) return;
                    for (uint32_t pos = 0; pos < alive.size(); pos++) {
                        const H2EOp& op = ops[alive[pos]];
                        auto add = [&](uint32_t pp) {
                            if (pp == pos || pp == 0xffffffffu) return;
                            if (std::find(preds[pos].begin(), preds[pos].end(), pp) != preds[pos].end()) return;
                            preds[pos].push_back(pp);
                            succs[pp].push_back(pos);
                        };
                        for (int q = 0; q < 3; q++)
                            if (dec[pos].val[q] >= 0) add(alive_pos[dec[pos].val[q] / 2]);
                        for (int q = 0; q < H2E_OP_MAX_REFS; q++) {
                            int wtr = writer_of(op.refs[q]);
                            if (wtr >= 0 && (uint32_t)wtr != alive[pos] && alive_pos[wtr] != 0xffffffffu && alive_pos[wtr] < pos) add(alive_pos[wtr]);
                        }
                    }
                };
                double modelled_us = 0;
                auto schedule_classes = [&](size_t step_ops, int cap, int policy) -> bool {
                    build_deps();
                    steps.clear();
                    std::fill(step_of.begin(), step_of.end(), 0u);
                    const size_t n = alive.size();
                    // sinks (results nothing in the replay reads: they are only stored) do not take part in the backward walk:
                    // at the end of the program they would pin their operands' slots; they are placed forward afterwards
                    std::vector<uint8_t> is_sink(n, 0);
                    for (uint32_t pos = 0; pos < n; pos++) is_sink[pos] = succs[pos].empty() ? 1 : 0;
                    std::vector<uint32_t> left(n, 0);
                    for (uint32_t pos = 0; pos < n; pos++)
                        for (uint32_t s : succs[pos])
                            if (!is_sink[s]) left[pos]++;
                    // ready ops by class; an op released while round r is formed may go into round r - 1 (backwards) at the earliest
                    std::vector<uint32_t> ready[4], released;
                    std::vector<uint8_t> done(n, 0);
                    size_t n_left = 0;
                    for (uint32_t pos = 0; pos < n; pos++) {
                        if (is_sink[pos]) {   // a sink whose operands are sinks' operands only: its preds count it as scheduled
                            continue;
                        }
                        n_left++;
                        if (left[pos] == 0) ready[cls_of(pos)].push_back(pos);
                    }
                    // sinks with non-sink consumers do not exist; sinks release their preds right away
                    std::vector<std::vector<std::vector<uint32_t>>> rounds_rev;   // [round][step] -> ops
                    std::vector<int> round_cls_rev;
                    auto take_steps = [&](std::vector<uint32_t>& pool, bool uniform_vop, std::vector<std::vector<uint32_t>>& rsteps) {
                        // as many steps as the round has waves left; ops that do not fit stay in the pool
                        std::vector<uint32_t> rest;
                        if (!uniform_vop) {
                            size_t i = 0;
                            while (i < pool.size() && rsteps.size() < NW) {
                                size_t j = std::min(pool.size(), i + step_ops);
                                rsteps.emplace_back(pool.begin() + i, pool.begin() + j);
                                i = j;
                            }
                            rest.assign(pool.begin() + i, pool.end());
                        } else {
                            std::stable_sort(pool.begin(), pool.end(), [&](uint32_t a, uint32_t b) { return vop_of(a) < vop_of(b); });
                            size_t i = 0;
                            while (i < pool.size()) {
                                size_t j = i + 1;
                                while (j < pool.size() && j - i < step_ops && vop_of(pool[j]) == vop_of(pool[i])) j++;
                                if (rsteps.size() < NW) rsteps.emplace_back(pool.begin() + i, pool.begin() + j);
                                else rest.insert(rest.end(), pool.begin() + i, pool.begin() + j);
                                i = j;
                            }
                        }
                        pool.swap(rest);
                    };
                    while (n_left > 0) {
                        std::vector<std::vector<uint32_t>> rsteps;
                        int rc = -1;
                        if (!ready[0].empty()) {
                            rc = 0;
                            take_steps(ready[0], false, rsteps);
                        } else if (!ready[3].empty()) {
                            rc = 3;
                            rsteps.push_back({ready[3].back()});
                            ready[3].pop_back();
                        } else if (policy == 0 ? !ready[1].empty() : (ready[2].empty() && !ready[1].empty())) {
                            rc = 1;   // policy 0: cheapest class first; policy 1: reduces ride along with products when both are ready
                            take_steps(ready[1], true, rsteps);
                        } else if (!ready[2].empty()) {
                            rc = 2;
                            take_steps(ready[2], true, rsteps);
                            if (policy == 1 && rsteps.size() < NW) take_steps(ready[1], true, rsteps);
                        } else {
                            throw std::runtime_error("replay compile: class scheduler stalled");
                        }
                        for (auto& st : rsteps)
                            for (uint32_t pos : st) {
                                done[pos] = 1;
                                n_left--;
                                for (uint32_t pp : preds[pos])
                                    if (--left[pp] == 0) released.push_back(pp);
                            }
                        for (uint32_t pp : released) ready[cls_of(pp)].push_back(pp);
                        released.clear();
                        rounds_rev.push_back(std::move(rsteps));
                        round_cls_rev.push_back(rc);
                    }
                    // forward order
                    std::vector<std::vector<std::vector<uint32_t>>> rounds(rounds_rev.rbegin(), rounds_rev.rend());
                    std::vector<int> round_cls(round_cls_rev.rbegin(), round_cls_rev.rend());
                    std::vector<uint32_t> round_of(n, 0);
                    for (size_t rd = 0; rd < rounds.size(); rd++)
                        for (auto& st : rounds[rd])
                            for (uint32_t pos : st) round_of[pos] = (uint32_t)rd;
                    // sinks: the first round after their operands that can take them without getting more expensive
                    for (uint32_t pos = 0; pos < n; pos++) {
                        if (!is_sink[pos]) continue;
                        size_t rd0 = 0;
                        for (uint32_t pp : preds[pos]) {
                            if (is_sink[pp] && !done[pp]) throw std::runtime_error("replay compile: sink reads an unplaced sink");
                            rd0 = std::max<size_t>(rd0, (size_t)round_of[pp] + 1);
                        }
                        int c = cls_of(pos);
                        uint32_t vop = vop_of(pos);
                        bool placed = false;
                        for (size_t rd = rd0; rd < rounds.size() && !placed; rd++) {
                            if (round_cls[rd] == 3 || c == 3) continue;
                            if (round_cls[rd] < c) continue;   // would make the round more expensive
                            for (auto& st : rounds[rd]) {
                                bool light_step = cls_of(st[0]) == 0;
                                if (st.size() < step_ops && ((c == 0 && light_step) || (c != 0 && vop_of(st[0]) == vop))) {
                                    st.push_back(pos);
                                    placed = true;
                                    break;
                                }
                            }
                            if (!placed && rounds[rd].size() < NW) {
                                rounds[rd].push_back({pos});
                                placed = true;
                            }
                            if (placed) round_of[pos] = (uint32_t)rd;
                        }
                        if (!placed) {
                            rounds.push_back({{pos}});
                            round_cls.push_back(c);
                            round_of[pos] = (uint32_t)rounds.size() - 1;
                        }
                        done[pos] = 1;
                    }
                    size_t cnt[4] = {0, 0, 0, 0};
                    for (size_t rd = 0; rd < rounds.size(); rd++) {
                        for (size_t w = 0; w < NW; w++) steps.emplace_back();
                        // heavier steps first, rotated over the waves like the level rounds
                        for (size_t k = 0; k < rounds[rd].size(); k++) {
                            steps[rd * NW + (round_cls[rd] == 3 ? k : (k % NW + rd) % NW)] = rounds[rd][k];
                            for (uint32_t pos : rounds[rd][k]) step_of[pos] = (uint32_t)rd;
                        }
                        cnt[round_cls[rd]]++;
                    }
                    n_rounds = rounds.size();
                    modelled_us = 0.55 * cnt[0] + 1.2 * cnt[1] + 3.0 * cnt[2] + 3.0 * cnt[3];
                    if (dbg_env("H2E_DUMP_TAPE"))
                        fprintf(stderr, "   class rounds (policy %d, steps of %zu): %zu light, %zu medium, %zu heavy, %zu through cells = %zu rounds, modelled %.2f ms\n",
                                policy, step_ops, cnt[0], cnt[1], cnt[2], cnt[3], n_rounds, modelled_us * 1e-3);
                    return alloc_slots(cap);
                };
                alloc_slots = [&](int cap) -> bool {
                slot_cap = cap;
                n_rounds = steps.size() / NW;
                // value slots over the round order: a slot freed in round r is reusable from round r + 1
                std::vector<uint32_t> last_step(2 * (size_t)n_ops, 0);
                for (uint32_t pos = 0; pos < alive.size(); pos++)
                    for (int q = 0; q < 3; q++)
                        if (dec[pos].val[q] >= 0) last_step[dec[pos].val[q]] = std::max(last_step[dec[pos].val[q]], step_of[pos]);
                lslot.assign(2 * (size_t)n_ops, -1);
                std::vector<std::vector<int>> free_at(n_rounds + 1);
                std::vector<int> free_list;
                n_slots = 0;
                bool fits = true;
                for (size_t rd = 0; rd < n_rounds && fits; rd++) {
                    for (int sl : free_at[rd]) free_list.push_back(sl);
                    for (size_t w = 0; w < NW; w++)
                        for (uint32_t pos : steps[rd * NW + w]) {
                            uint32_t i = alive[pos];
                            int k = kind_of(ops[i]);
                            int nres = (k == K_MUL || k == K_ADD || k == K_FE || k == K_CONST) ? 1 : 0;
                            for (int ww = 0; ww < nres; ww++) {
                                int v = 2 * (int)i + ww;
                                if (vals[v].uses.empty()) continue;
                                int sl;
                                if (!free_list.empty()) {
                                    sl = free_list.back();
                                    free_list.pop_back();
                                } else {
                                    sl = n_slots++;
                                }
                                lslot[v] = sl;
                                free_at[std::min<size_t>(last_step[v] + 1, n_rounds)].push_back(sl);
                            }
                        }
                    if (n_slots > slot_cap) fits = false;
                }
                return fits;
                };   // schedule
                // A/B knobs of the program compiler, read when a program is recorded (never while a run is queued):
                // H2E_LEVEL_MODE=pair (default) | single | wave : kernel shape - four waves and two / one instance(s) per workgroup
                //     (h2e_replay_levels) | one wave per instance, no barriers, compact records streamed through LDS (h2e_replay_wave:
                //     measured 26.5 vs 24.3 ms for 64 bn256 checks - a light round is ~3 k cycles of multi-word additions either way);
                // H2E_LEVEL_SCHED=levels|classes0|classes1 (default: the cheaper of the two class policies by the cost model;
                //     `levels` only with the four-wave kernels)
                bool by_classes = true, wave_mode = false;
                {
                    const uint32_t slot_bytes = (2 * (uint32_t)L + 4) * 8;
                    const int cap_pair = (int)((160u * 1024 - 4u * 1024) / 2 / slot_bytes), cap_single = (int)((160u * 1024 - 30u * 1024) / slot_bytes);
                    const char* mode = getenv("H2E_LEVEL_SCHED");
                    const char* kmode = dbg_env("H2E_LEVEL_MODE");   // ("wave": a kernel of -DH2E_AB_KERNELS engine units only)
                    const bool allow_pair = !(kmode && !strcmp(kmode, "single"));
                    wave_mode = kmode && !strcmp(kmode, "wave") && !(mode && !strcmp(mode, "levels"));
                    int forced = mode && !strcmp(mode, "classes0") ? 0 : mode && !strcmp(mode, "classes1") ? 1 : -1;
                    auto best_policy = [&](size_t step_ops, int cap) -> bool {
                        double best = 0;
                        int pick = -1;
                        for (int pol = 0; pol < 2; pol++) {
                            if (forced >= 0 && pol != forced) continue;
                            if (NW == 1 && pol == 1) continue;   // one step per round: nothing can ride along
                            if (schedule_classes(step_ops, cap, pol) && (pick < 0 || modelled_us < best)) {
                                pick = pol;
                                best = modelled_us;
                            }
                        }
                        if (pick < 0) return false;
                        return schedule_classes(step_ops, cap, pick);
                    };
                    if (wave_mode) {
                        NW = 1;
                        // (chunk buffers, ceil tables and a margin for other workgroups' static LDS come off the CU's 160 KB)
                        eligible = best_policy(64, (int)((160u * 1024 - 2u * H2E_WCHUNK * 32u - 8u * 1024) / slot_bytes));
                    } else if (mode && !strcmp(mode, "levels")) {
                        by_classes = false;
                        paired = allow_pair && schedule(32, cap_pair);
                        if (!paired) eligible = schedule(64, cap_single);
                    } else {
                        paired = allow_pair && best_policy(32, cap_pair);
                        if (!paired) eligible = best_policy(64, cap_single);
                    }
                }
                if (!eligible && dbg_env("H2E_DUMP_TAPE"))
                    fprintf(stderr, "segment %zu: level-parallel replay needs more than %d value slots (depth %u, %zu rounds)\n", si, slot_cap, depth, n_rounds);
                // one op of the schedule as a level record
                auto make_rec = [&](uint32_t pos, bool mixed) -> H2EVRec {
                    H2EVRec h{{0, 0, 0, 0, 0, 0, 0, 0}};
                    uint32_t i = alive[pos];
                    const H2EOp& op = ops[i];
                    int k = kind_of(op);
                    uint32_t vop = vop_of(pos), vflags = 0;
                    bool store = !(op.flags & H2E_FLAG_LOCAL_RESULT) || vals[2 * (size_t)i].force_store;
                    if (op.opcode == H2E_OP_AND || op.opcode == H2E_OP_OR || op.opcode == H2E_OP_XNOR || op.opcode == H2E_OP_BISEC_INT)
                        store = true;
                    if (store) vflags |= H2E_VFLAG_STORE;
                    if (vop == H2E_V_HINT && (op.flags & H2E_FLAG_HINT_STRIDED)) vflags |= H2E_VFLAG_HINT_STRIDED;
                    if (mixed) vflags |= H2E_VFLAG_MIXED;
                    int dsl = lslot[2 * (size_t)i];
                    h.w[0] = vop | (vflags << 8) | ((uint32_t)(dsl >= 0 ? dsl : 0xffff) << 16);
                    h.w[1] = vop == H2E_V_FULL ? i : op.imm;   // V_FULL: index of the tape op (segment relative)
                    h.w[5] = k == K_FE ? fe_row(op) : op.base_row;
                    h.w[6] = op.range_row;
                    if (vop != H2E_V_FULL) {
                        Opd o[3];
                        int n = operands(op, o);
                        for (int q = 0; q < n; q++) {
                            int v = dec[pos].val[q];
                            if (v >= 0) {
                                h.w[7] |= (uint32_t)(o[q].is_int ? H2E_VSRC_INT_SLOT : H2E_VSRC_FE_SLOT) << (3 * q);
                                h.w[2 + q] = (uint32_t)lslot[v];
                            } else {
                                h.w[7] |= (uint32_t)H2E_VSRC_GLOBAL << (3 * q);
                                if (o[q].is_int) {
                                    h.w[2 + q] = (uint32_t)h_lrefs.size();
                                    for (int j = 0; j <= L; j++) h_lrefs.push_back(op.refs[o[q].refpos + j]);
                                } else {
                                    h.w[2 + q] = o[q].ref;
                                }
                            }
                        }
                    }
                    return h;
                };
                if (eligible && wave_mode) {
                    // compact records in round order, padded so that no round straddles an H2E_WCHUNK-record chunk (the kernel
                    // streams the records through two LDS chunk buffers); per round: first record, count | kind << 8
                    // (kind: 0 = light ops of mixed opcodes, else the round's one opcode)
                    while (h_lrecs.size() % H2E_WCHUNK) h_lrecs.push_back(H2EVRec{{H2E_V_NOP, 0, 0, 0, 0, 0, 0, 0}});
                    seg_l_begin[si] = (uint32_t)h_lrecs.size();
                    seg_lr_begin[si] = (uint32_t)h_lrounds.size();
                    seg_l_steps[si] = (uint32_t)n_rounds;
                    seg_l_slots[si] = (uint32_t)std::max(1, n_slots);
                    seg_l_pair[si] = 2u;
                    for (size_t rd = 0; rd < n_rounds; rd++) {
                        auto& stp = steps[rd];
                        if (stp.empty() || stp.size() > 64) throw std::runtime_error("replay compile: bad wave round");
                        size_t at = h_lrecs.size() - seg_l_begin[si];
                        if (at % H2E_WCHUNK + stp.size() > H2E_WCHUNK)
                            while ((h_lrecs.size() - seg_l_begin[si]) % H2E_WCHUNK) h_lrecs.push_back(H2EVRec{{H2E_V_NOP, 0, 0, 0, 0, 0, 0, 0}});
                        at = h_lrecs.size() - seg_l_begin[si];
                        const bool mixed = cls_of(stp[0]) == 0;
                        h_lrounds.push_back((uint32_t)at);
                        h_lrounds.push_back((uint32_t)stp.size() | ((mixed ? 0u : vop_of(stp[0])) << 8));
                        for (uint32_t pos : stp) h_lrecs.push_back(make_rec(pos, mixed));
                    }
                    while ((h_lrecs.size() - seg_l_begin[si]) % H2E_WCHUNK) h_lrecs.push_back(H2EVRec{{H2E_V_NOP, 0, 0, 0, 0, 0, 0, 0}});
                    seg_l_recs[si] = (uint32_t)(h_lrecs.size() - seg_l_begin[si]);
                }
                if (eligible && !wave_mode) {
                    seg_l_begin[si] = (uint32_t)h_lrecs.size();
                    seg_l_steps[si] = (uint32_t)n_rounds;
                    seg_l_slots[si] = (uint32_t)std::max(1, n_slots);
                    seg_l_pair[si] = paired ? 1u : 0u;
                    for (size_t sidx = 0; sidx < steps.size(); sidx++) {
                        auto& stp = steps[sidx];
                        // the other waves of a V_FULL round fence their stores before the barrier (lane 0 of their NOP step says so)
                        bool full_round = false;
                        for (size_t w = 0; w < NW; w++) {
                            const auto& other = steps[sidx / NW * NW + w];
                            full_round = full_round || (!other.empty() && vop_of(other[0]) == H2E_V_FULL);
                        }
                        const size_t step_lanes = paired ? 32 : 64;
                        // a step of light ops holds any mix of their opcodes (class rounds): the kernel dispatches per lane
                        const bool mixed = by_classes && !stp.empty() && cls_of(stp[0]) == 0;
                        std::vector<H2EVRec> step_recs;
                        for (size_t lane = 0; lane < step_lanes; lane++) {
                            H2EVRec h{{H2E_V_NOP | ((full_round && lane == 0) ? (H2E_VFLAG_FENCE << 8) : 0u), 0, 0, 0, 0, 0, 0, 0}};
                            if (lane < stp.size()) h = make_rec(stp[lane], mixed);
                            step_recs.push_back(h);
                        }
                        // (paired: the second half of the wave runs the same records for the workgroup's other instance)
                        for (size_t rep2 = 0; rep2 < 64 / step_lanes; rep2++) h_lrecs.insert(h_lrecs.end(), step_recs.begin(), step_recs.end());
                    }
                }
                if (eligible) {
                    if (dbg_env("H2E_DUMP_TAPE")) {
                        fprintf(stderr, "segment %zu: level-parallel replay: %zu alive ops, depth %u, %zu rounds of %zu waves, %d value slots, %s\n", si,
                                alive.size(), depth, n_rounds, NW, n_slots, wave_mode ? "one wave per instance" : paired ? "two instances per workgroup" : "one instance per workgroup");
                        // rounds by their most expensive op kind, and how many of them read an operand from global cells
                        std::map<uint32_t, std::pair<size_t, size_t>> by_vop;
                        size_t global_rounds = 0, global_operands = 0;
                        for (size_t rd = 0; rd < n_rounds; rd++) {
                            uint32_t worst = 0;
                            bool g = false;
                            for (size_t w = 0; w < NW; w++)
                                for (uint32_t pos : steps[rd * NW + w]) {
                                    uint32_t vop = vop_of(pos);
                                    auto rank = [](uint32_t v) { return v == H2E_V_FULL ? 100u : v == H2E_V_DIV ? 90u : v == H2E_V_MUL ? 80u : v == H2E_V_REDUCE ? 70u : 10u; };
                                    if (rank(vop) > rank(worst) || worst == 0) worst = vop;
                                    Opd o[3];
                                    int n = vop == H2E_V_FULL ? 0 : operands(ops[alive[pos]], o);
                                    for (int q = 0; q < n; q++)
                                        if (dec[pos].val[q] < 0) {
                                            g = true;
                                            global_operands++;
                                        }
                                }
                            by_vop[worst].first++;
                            if (g) {
                                by_vop[worst].second++;
                                global_rounds++;
                            }
                        }
                        for (auto& kv : by_vop) fprintf(stderr, "   rounds led by vop %u: %zu (%zu with a global operand)\n", kv.first, kv.second.first, kv.second.second);
                        fprintf(stderr, "   %zu rounds with global operands, %zu global operands in all\n", global_rounds, global_operands);
                    }
                }
            }
        }
        // ---- pieces ------------------------------------------------------------------------------------------
        // The replay is one dependent chain only through values.  Where every value that is live across a position
        // can be rebuilt from hints / external cells by a few ops (its producers' closure), the chain is cut there:
        // the next piece starts with that closure as a prologue (results not stored) and runs in its own lanes.
        std::vector<uint32_t> pos_of_op(n_ops, 0xffffffffu);
        for (uint32_t pos = 0; pos < alive.size(); pos++) pos_of_op[alive[pos]] = pos;
        const uint32_t INF = 0xffffffffu;
        auto prod_pos = [&](int v) { return pos_of_op[v / 2]; };
        auto last_use = [&](int v) -> uint32_t { return vals[v].uses.empty() ? 0 : vals[v].uses.back(); };
        // values produced by the replay in program order, for the live-set scan
        std::vector<int> produced;
        for (uint32_t pos = 0; pos < alive.size(); pos++) {
            int k = kind_of(ops[alive[pos]]);
            int nres = k == K_SEL ? 2 : (k == K_MUL || k == K_ADD || k == K_FE || k == K_CONST) ? 1 : 0;
            for (int w = 0; w < nres; w++) produced.push_back(2 * (int)alive[pos] + w);
        }
        auto is_fe_val = [&](int v) { return kind_of(ops[v / 2]) == K_FE; };
        struct Restart {
            uint32_t pos;
            std::vector<uint32_t> prologue;            // alive positions, program order
            std::map<int, int> slot_of;                // value -> slot during the prologue
        };
        const uint32_t PIECE_TARGET = 96, PIECE_BUDGET = 40;
        static const bool pieces_on = !dbg_env("H2E_NO_PIECES");
        static const bool stage_on = !dbg_env("H2E_NO_STAGE");
        // positions that can never be cut: an op at or after p reads a *cell* written before p (rows of a V_FULL op, or
        // a value that lost / never had its slot) - difference arrays over (writer, last reader]
        std::vector<int32_t> blocked(alive.size() + 2, 0);
        for (uint32_t q = 0; q < alive.size(); q++)
            if (kind_of(ops[alive[q]]) == K_FULL && full_read_last[alive[q]] > q) {
                blocked[q + 1]++;
                blocked[full_read_last[alive[q]] + 1]--;
            }
        for (int v : produced)
            if (vals[v].cell_use_last != INF && vals[v].cell_use_last > prod_pos(v)) {
                blocked[prod_pos(v) + 1]++;
                blocked[vals[v].cell_use_last + 1]--;
            }
        for (size_t q = 1; q < blocked.size(); q++) blocked[q] += blocked[q - 1];
        // values whose last slot-use is at a given position (to keep the live set incrementally)
        std::vector<std::vector<int>> expires(alive.size() + 1);
        for (int v : produced)
            if (!vals[v].uses.empty()) expires[last_use(v)].push_back(v);
        auto try_restart = [&](uint32_t p, const std::set<int>& live_set, Restart& rs) -> bool {
            if (blocked[p] > 0) return false;
            std::vector<int> live;
            for (int v : live_set) {
                if (vals[v].dst_slot < 0 || vals[v].evicted) return false;
                live.push_back(v);
            }
            std::set<uint32_t> closure;
            std::set<int> cvals;
            std::vector<int> work(live.begin(), live.end());
            while (!work.empty()) {
                int v = work.back();
                work.pop_back();
                uint32_t q = prod_pos(v);
                // (both results of a SELECT_POINT come from one op)
                cvals.insert(v);
                if (!closure.insert(q).second) continue;
                if (closure.size() > PIECE_BUDGET) return false;
                if (kind_of(ops[alive[q]]) == K_FULL) return false;
                for (int j = 0; j < 3; j++)
                    if (dec[q].val[j] >= 0) work.push_back(dec[q].val[j]);
            }
            rs.pos = p;
            rs.prologue.assign(closure.begin(), closure.end());
            std::vector<bool> int_used(NS, false), fe_used(NF, false);
            for (int v : live) {
                rs.slot_of[v] = vals[v].dst_slot;
                (is_fe_val(v) ? fe_used : int_used)[vals[v].dst_slot] = true;
            }
            for (int v : cvals) {
                if (rs.slot_of.count(v)) continue;
                auto& used = is_fe_val(v) ? fe_used : int_used;
                int sl = -1;
                for (size_t t = 0; t < used.size(); t++)
                    if (!used[t]) {
                        sl = (int)t;
                        break;
                    }
                if (sl < 0) return false;
                used[sl] = true;
                rs.slot_of[v] = sl;
            }
            return true;
        };
        std::vector<Restart> restarts;
        const uint32_t INT_UNITS = (uint32_t)L + 2, HINT_UNITS = (uint32_t)r.fp.w_words / 2, FE_UNITS = 2;
        const uint32_t UNIT_TARGET = 44, UNIT_MAX = 64;   // staging units (16 bytes per lane) a piece may gather
        auto stageable = [&](uint32_t ref) { return ref != H2E_NO_REF && H2E_REF_REGION(ref) != H2E_REGION_PARAM && writer_of(ref) < 0; };
        auto units_of = [&](uint32_t pos) -> uint32_t {   // upper estimate of the memory inputs of one op
            const H2EOp& op = ops[alive[pos]];
            if ((op.flags & H2E_FLAG_HINTED) && kind_of(op) == K_MUL) return HINT_UNITS;
            if (op.opcode == H2E_OP_SELECT_POINT && (op.flags & H2E_FLAG_PRESELECTED)) return 2 * HINT_UNITS;
            Opd o[3];
            int n = operands(op, o);
            uint32_t u = 0;
            for (int q = 0; q < n; q++)
                if (dec[pos].val[q] < 0 && stageable(o[q].ref)) u += o[q].is_int ? INT_UNITS : FE_UNITS;
            return u;
        };
        if (pieces_on) {
            uint32_t since = 0, units = 0, retry_at = 0;
            std::set<int> live_set;   // values produced before pos with a slot-use at or after pos
            for (uint32_t pos = 1; pos < alive.size(); pos++) {
                {   // advance the live set from pos - 1 to pos
                    uint32_t i = alive[pos - 1];
                    int k = kind_of(ops[i]);
                    int nres = k == K_SEL ? 2 : (k == K_MUL || k == K_ADD || k == K_FE || k == K_CONST) ? 1 : 0;
                    for (int w = 0; w < nres; w++)
                        if (!vals[2 * (size_t)i + w].uses.empty() && last_use(2 * (int)i + w) >= pos) live_set.insert(2 * (int)i + w);
                    for (int v : expires[pos - 1]) live_set.erase(v);
                }
                since++;
                units += units_of(pos - 1);
                if ((since < PIECE_TARGET && units < UNIT_TARGET) || pos < retry_at) continue;
                Restart rs;
                if (try_restart(pos, live_set, rs)) {
                    restarts.push_back(std::move(rs));
                    since = 0;
                    units = 0;
                } else {
                    retry_at = pos + 4;   // (a failed attempt costs a closure walk: do not try every position)
                }
            }
        }
        // ---- emit ------------------------------------------------------------------------------------------------
        std::vector<H2EVRec> out;
        auto pad_chunk = [&]() {
            while (out.size() % H2E_VCHUNK) out.push_back(H2EVRec{{H2E_V_NOP, 0, 0, 0, 0, 0, 0, 0}});
        };
        auto pad_to = [&](size_t need) {
            if (out.size() % H2E_VCHUNK + need > H2E_VCHUNK) pad_chunk();
        };
        // remap: nullptr = the op in its own place; else the prologue copy (slots from the map, nothing stored)
        struct StageMap {
            std::map<uint32_t, uint32_t> hint, cells, sel;   // hint slot | strided << 31 -> unit ; first cell ref -> unit ; selection entry -> unit
            uint32_t units = 0;
        };
        auto emit = [&](uint32_t pos, const std::map<int, int>* remap, const StageMap& sm) {
            uint32_t i = alive[pos];
            const H2EOp& op = ops[i];
            int k = kind_of(op);
            const Dec& d = dec[pos];
            std::vector<uint32_t> ext;
            H2EVRec h{{0, 0, 0, 0, 0, 0, 0, 0}};
            auto dst_of = [&](int w) -> uint32_t {
                if (!remap) return d.dst[w] >= 0 ? (uint32_t)d.dst[w] : H2E_V_NO_SLOT;
                auto it = remap->find(2 * (int)i + w);
                return it == remap->end() ? H2E_V_NO_SLOT : (uint32_t)it->second;
            };
            uint32_t vop = H2E_V_NOP, vflags = 0, dst = dst_of(0);
            bool hinted = (op.flags & H2E_FLAG_HINTED) != 0;
            switch (op.opcode) {
                case H2E_OP_INT_MUL: vop = hinted ? H2E_V_HINT : H2E_V_MUL; break;
                case H2E_OP_REDUCE: vop = hinted ? H2E_V_HINT : H2E_V_REDUCE; break;
                case H2E_OP_DIV_CORE: vop = hinted ? H2E_V_HINT : H2E_V_DIV; break;
                case H2E_OP_INT_ADD: vop = H2E_V_ADD; break;
                case H2E_OP_INT_SUB: vop = H2E_V_SUB; break;
                case H2E_OP_INT_NEG: vop = H2E_V_NEG; break;
                case H2E_OP_INT_MUL_SMALL: vop = H2E_V_MUL_SMALL; break;
                case H2E_OP_MASK_INT: vop = H2E_V_MASK; break;
                case H2E_OP_BISEC_INT: vop = H2E_V_BISEC_INT; break;
                case H2E_OP_IS_INT_ZERO: vop = H2E_V_IS_ZERO; break;
                case H2E_OP_NOT: vop = H2E_V_NOT; break;
                case H2E_OP_AND: vop = H2E_V_AND; break;
                case H2E_OP_OR: vop = H2E_V_OR; break;
                case H2E_OP_XNOR: vop = H2E_V_XNOR; break;
                case H2E_OP_PICK_INDEX: vop = H2E_V_PICK_INDEX; break;
                case H2E_OP_SELECT_POINT: vop = (op.flags & H2E_FLAG_PRESELECTED) ? H2E_V_LOAD_SEL : H2E_V_SELECT_POINT; break;
                case H2E_OP_CONST_INT: vop = H2E_V_CONST; break;
                default: vop = H2E_V_FULL; break;
            }
            if (hinted && (op.flags & H2E_FLAG_HINT_STRIDED)) vflags |= H2E_VFLAG_HINT_STRIDED;
            uint32_t imm = op.imm;
            if (vop == H2E_V_HINT) {
                auto it = sm.hint.find(op.imm | ((op.flags & H2E_FLAG_HINT_STRIDED) ? 0x80000000u : 0));
                if (it != sm.hint.end()) {
                    vflags |= H2E_VFLAG_STAGED;
                    imm = it->second;
                }
            }
            bool store = !(op.flags & H2E_FLAG_LOCAL_RESULT) || vals[2 * (size_t)i].force_store || vals[2 * (size_t)i + 1].force_store;
            if (k == K_SEL || op.opcode == H2E_OP_AND || op.opcode == H2E_OP_OR || op.opcode == H2E_OP_XNOR || op.opcode == H2E_OP_PICK_INDEX ||
                op.opcode == H2E_OP_BISEC_INT)
                store = true;   // never flagged local
            if (vop == H2E_V_LOAD_SEL) store = false;   // the expansion writes the select rows
            if (store && !remap) vflags |= H2E_VFLAG_STORE;
            h.w[1] = imm;
            h.w[5] = k == K_FE ? fe_row(op) : op.base_row;
            h.w[6] = k == K_SEL ? op.select_row : op.range_row;
            if (vop == H2E_V_FULL) {
                const uint32_t* raw = (const uint32_t*)&op;
                ext.assign(raw, raw + 16);
            } else if (vop == H2E_V_PICK_INDEX) {
                for (uint32_t q = 0; q < op.imm && q < 5; q++) ext.push_back(op.refs[q]);
            } else if (vop == H2E_V_LOAD_SEL) {
                auto it = sm.sel.find(op.refs[1]);
                if (it != sm.sel.end()) {
                    h.w[2] = it->second;
                    h.w[7] |= H2E_VSRC_STAGE;
                } else {
                    h.w[2] = op.refs[1];
                    h.w[7] |= H2E_VSRC_GLOBAL;
                }
                h.w[7] |= dst_of(1) << 16;
            } else {
                Opd o[3];
                int n = operands(op, o);
                for (int q = 0; q < n; q++) {
                    uint32_t kind = d.kind[q], word = d.word[q];
                    if (remap && d.val[q] >= 0) {   // a value of the closure: in the slot the prologue gave it
                        kind = o[q].is_int ? H2E_VSRC_INT_SLOT : H2E_VSRC_FE_SLOT;
                        word = (uint32_t)remap->at(d.val[q]);
                    }
                    if (kind == H2E_VSRC_GLOBAL && d.val[q] < 0) {   // an input from memory: staged by this piece's gather?
                        auto it = sm.cells.find(o[q].ref);
                        if (it != sm.cells.end()) {
                            kind = H2E_VSRC_STAGE;
                            word = it->second;
                        }
                    }
                    h.w[7] |= kind << (3 * q);
                    if (kind != H2E_VSRC_GLOBAL) {
                        h.w[2 + q] = word;
                    } else if (o[q].is_int) {
                        h.w[2 + q] = (uint32_t)ext.size();
                        for (int j = 0; j <= L; j++) ext.push_back(op.refs[o[q].refpos + j]);
                    } else {
                        h.w[2 + q] = o[q].ref;
                    }
                }
                if (k == K_SEL) h.w[7] |= dst_of(1) << 16;
            }
            uint32_t n_ext = (uint32_t)((ext.size() + 7) / 8);
            h.w[0] = vop | (vflags << 8) | (dst << 16) | (n_ext << 24);
            pad_to(1 + n_ext);
            out.push_back(h);
            ext.resize((size_t)n_ext * 8, H2E_NO_REF);
            for (uint32_t e = 0; e < n_ext; e++) {
                H2EVRec x;
                for (int j = 0; j < 8; j++) x.w[j] = ext[e * 8 + j];
                out.push_back(x);
            }
        };
        size_t si = (size_t)(sg - r.segments.data());
        uint32_t vbase = (uint32_t)h_vtape.size();   // multiple of H2E_VCHUNK
        seg_piece_begin[si] = (uint32_t)h_vpieces.size() / 2;
        uint32_t max_units = 0;
        // LDS sizing: integer slots actually used; what is left of 130 KB (28 KB are static: row staging for H2E_V_FULL ops,
        // record chunks; h2e_engine_launch re-checks the sum) bounds the staging units
        uint32_t used_slots = 1;
        for (auto& d : dec)
            for (int w = 0; w < 2; w++)
                if (d.dst[w] >= 0 && kind_of(ops[alive[&d - dec.data()]]) != K_FE) used_slots = std::max(used_slots, (uint32_t)d.dst[w] + 1);
        for (auto& rs : restarts)
            for (auto& kv : rs.slot_of)
                if (!is_fe_val(kv.first)) used_slots = std::max(used_slots, (uint32_t)kv.second + 1);
        const uint32_t slot_bytes = (2 * (uint32_t)L + 4) * 512;
        const uint32_t unit_cap = std::min<uint32_t>(UNIT_MAX, (130u * 1024 - 8192 - used_slots * slot_bytes) / 1024);
        // one piece: gather records for its memory inputs, the prologue (if it restarts), the body
        auto emit_piece = [&](const Restart* rs, uint32_t pos_begin, uint32_t pos_end) {
            uint32_t piece_first = (uint32_t)out.size();
            StageMap sm;
            struct GEntry { uint32_t meta, ref; };
            std::vector<GEntry> gl;
            auto consider = [&](uint32_t pos, bool in_prologue) {
                const H2EOp& op = ops[alive[pos]];
                if ((op.flags & H2E_FLAG_HINTED) && kind_of(op) == K_MUL) {
                    uint32_t key = op.imm | ((op.flags & H2E_FLAG_HINT_STRIDED) ? 0x80000000u : 0);
                    if (sm.hint.count(key) || sm.units + HINT_UNITS > unit_cap) return;
                    sm.hint[key] = sm.units;
                    for (uint32_t hf = 0; hf < HINT_UNITS; hf++)
                        gl.push_back(GEntry{1u | (hf << 4) | ((op.flags & H2E_FLAG_HINT_STRIDED) ? 0x100u : 0), op.imm});
                    sm.units += HINT_UNITS;
                    return;
                }
                if (kind_of(op) == K_FULL || op.opcode == H2E_OP_PICK_INDEX) return;
                if (op.opcode == H2E_OP_SELECT_POINT && (op.flags & H2E_FLAG_PRESELECTED)) {
                    if (sm.sel.count(op.refs[1]) || sm.units + 2 * HINT_UNITS > unit_cap) return;
                    sm.sel[op.refs[1]] = sm.units;
                    for (uint32_t which = 0; which < 2; which++)
                        for (uint32_t hf = 0; hf < HINT_UNITS; hf++)
                            gl.push_back(GEntry{2u | ((which * (H2E_W_WORDS_MAX / 2) + hf) << 4), op.refs[1]});
                    sm.units += 2 * HINT_UNITS;
                    return;
                }
                Opd o[3];
                int n = operands(op, o);
                for (int q = 0; q < n; q++) {
                    bool external = dec[pos].val[q] < 0;
                    (void)in_prologue;
                    if (!external || !stageable(o[q].ref) || sm.cells.count(o[q].ref)) continue;
                    uint32_t need = o[q].is_int ? INT_UNITS : FE_UNITS;
                    bool ok = sm.units + need <= unit_cap;
                    if (o[q].is_int)
                        for (int j = 0; j <= L; j++) ok = ok && stageable(op.refs[o[q].refpos + j]);
                    if (!ok) continue;
                    sm.cells[o[q].ref] = sm.units;
                    if (o[q].is_int) {
                        for (int j = 0; j < L; j++) gl.push_back(GEntry{0u, op.refs[o[q].refpos + j]});   // low 16 bytes of a limb cell
                        gl.push_back(GEntry{0u, op.refs[o[q].refpos + L]});
                        gl.push_back(GEntry{0u | (1u << 4), op.refs[o[q].refpos + L]});
                    } else {
                        gl.push_back(GEntry{0u, o[q].ref});
                        gl.push_back(GEntry{0u | (1u << 4), o[q].ref});
                    }
                    sm.units += need;
                }
            };
            if (rs)
                for (uint32_t q : rs->prologue) consider(q, true);
            for (uint32_t pos = pos_begin; pos < pos_end; pos++) consider(pos, false);
            // Staging (asynchronous gathers into LDS ahead of the serial chain) pays for segments with few lanes, whose
            // time is load latency; a segment with thousands of workgroups (the MSM windows: 254 strands x 38 pieces) is
            // bound by how many of them fit on a CU, and the staging area is half of its LDS (1.85 -> 0.85 ms).
            if (!stage_on || (uint64_t)sg->n_strands * (restarts.size() + 1) >= 2048) {
                sm = StageMap();
                gl.clear();
            }
            for (size_t e = 0; e < gl.size(); e += 3) {
                uint32_t n = (uint32_t)std::min<size_t>(3, gl.size() - e);
                H2EVRec g{{H2E_V_GATHER | (n << 8), (uint32_t)e, 0, 0, 0, 0, 0, 0}};
                for (uint32_t j = 0; j < n; j++) {
                    g.w[2 + 2 * j] = gl[e + j].meta;
                    g.w[3 + 2 * j] = gl[e + j].ref;
                }
                out.push_back(g);
            }
            if (!gl.empty()) out.push_back(H2EVRec{{H2E_V_GATHER_WAIT, 0, 0, 0, 0, 0, 0, 0}});
            max_units = std::max(max_units, sm.units);
            if (rs)
                for (uint32_t q : rs->prologue) emit(q, &rs->slot_of, sm);
            for (uint32_t pos = pos_begin; pos < pos_end; pos++) emit(pos, nullptr, sm);
            h_vpieces.push_back(vbase + piece_first);
            h_vpieces.push_back(vbase + (uint32_t)out.size());
            pad_chunk();
        };
        {
            uint32_t begin = 0;
            for (size_t ri = 0; ri <= restarts.size(); ri++) {
                uint32_t end = ri < restarts.size() ? restarts[ri].pos : (uint32_t)alive.size();
                emit_piece(ri == 0 ? nullptr : &restarts[ri - 1], begin, end);
                begin = end;
            }
        }
        seg_n_pieces[si] = (uint32_t)h_vpieces.size() / 2 - seg_piece_begin[si];
        seg_v_slots[si] = used_slots;
        seg_v_units[si] = std::max(1u, max_units);
        h_vtape.insert(h_vtape.end(), out.begin(), out.end());
        if (dbg_env("H2E_DUMP_TAPE"))
            fprintf(stderr, "segment %zu: replay %zu alive ops, %zu records, %u pieces, %u int slots, %u staging units\n", si, alive.size(),
                    out.size(), seg_n_pieces[si], seg_v_slots[si], seg_v_units[si]);
    }

    // A segment without cuts normally runs on the caller's (critical) stream because later value-chain kernels may
    // read any of its cells.  If no reference anywhere (later ops, strand parameters, candidate tables, predictor
    // arguments, outputs) points into its rows, it can run on the expansion stream instead.
    void mark_deferrable() {
        h2e::Recorder& r = *rec;
        seg_deferrable.assign(r.segments.size(), 0);
        for (size_t si = 0; si < r.segments.size(); si++) {
            const h2e::Segment& sg = r.segments[si];
            if (sg.n_cuts != 0 || sg.tape_end <= sg.tape_begin || !sg.is_fork) continue;
            uint32_t lo[3] = {sg.base0, sg.range0, sg.select0};
            uint64_t hi[3] = {sg.base0 + (uint64_t)sg.dbase * sg.n_strands, sg.range0 + (uint64_t)sg.drange * sg.n_strands,
                              sg.select0 + (uint64_t)sg.dselect * sg.n_strands};
            auto hits = [&](uint32_t ref) {
                if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM || H2E_REF_REL(ref)) return false;
                uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
                return row >= lo[region] && row < hi[region];
            };
            bool referenced = false;
            for (size_t sj = 0; sj < r.segments.size() && !referenced; sj++) {
                if (sj == si) continue;
                for (uint32_t i = r.segments[sj].tape_begin; i < r.segments[sj].tape_end && !referenced; i++)
                    for (int k = 0; k < H2E_OP_MAX_REFS; k++) referenced = referenced || hits(r.tape[i].refs[k]);
            }
            bool dbg = dbg_env("H2E_DUMP_TAPE") != nullptr;
            if (dbg && referenced) fprintf(stderr, "segment %zu referenced by ops\n", si);
            for (uint32_t ref : r.aux) if (hits(ref)) { if (dbg && !referenced) fprintf(stderr, "segment %zu referenced by aux %08x\n", si, ref); referenced = true; }
            for (uint32_t ref : r.params) if (hits(ref)) { if (dbg && !referenced) fprintf(stderr, "segment %zu referenced by params %08x\n", si, ref); referenced = true; }
            for (uint32_t ref : r.outputs) referenced = referenced || hits(ref);
            for (uint32_t ref : r.pre_args) if (hits(ref)) { if (dbg && !referenced) fprintf(stderr, "segment %zu referenced by pre_args %08x\n", si, ref); referenced = true; }
            seg_deferrable[si] = referenced ? 0 : 1;
            if (dbg_env("H2E_DUMP_TAPE")) fprintf(stderr, "segment %zu deferrable %d\n", si, (int)seg_deferrable[si]);
        }
    }

    // A fork segment without cuts that only reads segments without cuts (the scalar decomposition: it reads the
    // assigned scalars, and only the MSM windows read its bits) need not sit in the value chain between its neighbours:
    // it runs on a side stream as soon as the last segment it reads is done, and the first segment that reads it waits.
    std::vector<int32_t> seg_side_dep;      // -2: not a side segment; else index of the last segment it depends on (-1: none)
    std::vector<uint32_t> seg_first_reader; // for side segments: first later segment that references its rows
    void mark_side_segments() {
        h2e::Recorder& r = *rec;
        size_t ns = r.segments.size();
        seg_side_dep.assign(ns, -2);
        seg_first_reader.assign(ns, (uint32_t)ns);
        // first rows of every non-empty segment, per region (rows are handed out in program order)
        std::vector<uint32_t> ids;
        std::vector<std::array<uint32_t, 3>> start;
        for (size_t si = 0; si < ns; si++) {
            const h2e::Segment& sg = r.segments[si];
            if (sg.tape_end <= sg.tape_begin) continue;
            std::array<uint32_t, 3> st;
            if (sg.is_fork) st = {sg.base0, sg.range0, sg.select0};
            else st = {r.tape[sg.tape_begin].base_row, r.tape[sg.tape_begin].range_row, r.tape[sg.tape_begin].select_row};
            ids.push_back((uint32_t)si);
            start.push_back(st);
        }
        auto segment_of = [&](uint32_t ref) -> int {
            if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM || H2E_REF_REL(ref)) return -1;
            uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
            int ans = -1;
            for (size_t k = 0; k < ids.size(); k++)
                if (start[k][region] <= row) ans = (int)ids[k];
            return ans;
        };
        for (size_t si = 0; si < ns; si++) {
            const h2e::Segment& sg = r.segments[si];
            if (sg.n_cuts != 0 || sg.tape_end <= sg.tape_begin || !sg.is_fork) continue;
            bool has_pre = false;
            for (auto& pk : r.pre_kernels) has_pre = has_pre || pk.before_segment == si;
            if (has_pre) continue;
            int last_dep = -1;
            bool ok = true;
            auto dep = [&](uint32_t ref) {
                int sj = segment_of(ref);
                if (sj < 0) return;
                if ((size_t)sj >= si) { ok = false; return; }
                if (r.segments[sj].n_cuts != 0) ok = false;
                if (seg_side_dep[sj] != -2) return;   // another side segment: the side stream runs them in order
                last_dep = std::max(last_dep, sj);
            };
            for (uint32_t i = sg.tape_begin; i < sg.tape_end; i++)
                for (int k = 0; k < H2E_OP_MAX_REFS; k++) dep(r.tape[i].refs[k]);
            for (size_t q = 0; q < (size_t)sg.n_params * sg.n_strands; q++)
                if (sg.params_begin + q < r.params.size()) dep(r.params[sg.params_begin + q]);
            if (!ok) continue;
            // first later reader
            uint32_t lo[3] = {sg.base0, sg.range0, sg.select0};
            uint64_t hi[3] = {sg.base0 + (uint64_t)sg.dbase * sg.n_strands, sg.range0 + (uint64_t)sg.drange * sg.n_strands,
                              sg.select0 + (uint64_t)sg.dselect * sg.n_strands};
            auto hits = [&](uint32_t ref) {
                if (ref == H2E_NO_REF || H2E_REF_REGION(ref) == H2E_REGION_PARAM || H2E_REF_REL(ref)) return false;
                uint32_t region = H2E_REF_REGION(ref), row = H2E_REF_ROW(ref);
                return row >= lo[region] && row < hi[region];
            };
            uint32_t first_reader = (uint32_t)ns;
            for (size_t sj = si + 1; sj < ns && first_reader == ns; sj++) {
                const h2e::Segment& sr = r.segments[sj];
                bool reads = false;
                for (uint32_t i = sr.tape_begin; i < sr.tape_end && !reads; i++)
                    for (int k = 0; k < H2E_OP_MAX_REFS; k++) reads = reads || hits(r.tape[i].refs[k]);
                for (size_t q = 0; q < (size_t)sr.n_params * sr.n_strands && !reads; q++)
                    if (sr.params_begin + q < r.params.size()) reads = hits(r.params[sr.params_begin + q]);
                for (auto& pk : r.pre_kernels) {
                    if (pk.before_segment != sj || reads) continue;
                    for (uint32_t q = 0; q < pk.k.n_params * pk.k.n_lanes && !reads; q++)
                        if (pk.k.params_begin + q < r.params.size()) reads = hits(r.params[pk.k.params_begin + q]);
                    reads = reads || true;   // predictor arguments are not delimited per kernel: be conservative
                }
                if (reads) first_reader = (uint32_t)sj;
            }
            bool in_aux = false;
            for (uint32_t ref : r.aux) in_aux = in_aux || hits(ref);
            if (in_aux) first_reader = std::min<uint32_t>(first_reader, (uint32_t)si + 1);
            if (first_reader <= si + 1) continue;   // nothing to overlap with
            seg_side_dep[si] = last_dep;
            seg_first_reader[si] = first_reader;
            if (dbg_env("H2E_DUMP_TAPE")) fprintf(stderr, "segment %zu: side stream after segment %d, first reader %u\n", si, last_dep, first_reader);
        }
    }

    // Expansion result cache (engine.hip ld_int_x / xc_put_x): per sub-range of a cut segment, which of the three LDS
    // entries an integer result goes to and which operands are read from them - furthest-next-use replacement over the
    // static op sequence.  Encoded in op.flags bits 8-15.
    void assign_expansion_slots() {
        h2e::Recorder& r = *rec;
        const int L = r.fp.limbs;
        const int NSLOT = 3;
        for (auto& sg : r.segments) {
            uint32_t n_ops = sg.tape_end - sg.tape_begin;
            if (sg.n_cuts == 0 || n_ops == 0) continue;
            H2EOp* ops = r.tape.data() + sg.tape_begin;
            const uint32_t rel = sg.is_fork ? 1 : 0;
            std::vector<uint32_t> bounds;
            uint32_t lastb = 0;
            for (uint32_t k = 0; k < sg.n_cuts; k++) {
                uint32_t at = r.cuts[sg.cuts_begin + k];
                if (at > lastb && at < n_ops) {
                    bounds.push_back(at);
                    lastb = at;
                }
            }
            bounds.push_back(n_ops);
            auto result_key = [&](const H2EOp& op) -> uint32_t {
                switch (op.opcode) {
                    case H2E_OP_INT_ADD: case H2E_OP_INT_SUB: case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL:
                        return H2E_MAKE_REF(0, 4, rel, op.base_row);
                    case H2E_OP_INT_MUL: case H2E_OP_REDUCE: case H2E_OP_DIV_CORE:
                        return H2E_MAKE_REF(1, 0, rel, op.range_row);
                    default: return H2E_NO_REF;
                }
            };
            auto operand_pos = [&](const H2EOp& op, int* pos) -> int {
                switch (op.opcode) {
                    case H2E_OP_INT_ADD: case H2E_OP_INT_SUB: case H2E_OP_INT_MUL: case H2E_OP_DIV_CORE:
                        pos[0] = 0; pos[1] = L + 1; return 2;
                    case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: case H2E_OP_REDUCE: case H2E_OP_IS_INT_ZERO: case H2E_OP_MASK_INT:
                        pos[0] = 0; return 1;
                    default: return 0;
                }
            };
            uint32_t lo = 0;
            for (uint32_t hi : bounds) {
                std::map<uint32_t, uint32_t> producer_of;          // key -> op index
                std::map<uint32_t, std::vector<uint32_t>> uses;     // producer op -> consumer op indices (ascending)
                for (uint32_t i = lo; i < hi; i++) {
                    ops[i].flags &= 0x00ff;
                    int pos[2];
                    int n = operand_pos(ops[i], pos);
                    for (int q = 0; q < n; q++) {
                        auto it = producer_of.find(ops[i].refs[pos[q]]);
                        if (it != producer_of.end()) uses[it->second].push_back(i);
                    }
                    uint32_t key = result_key(ops[i]);
                    if (key != H2E_NO_REF) producer_of[key] = i;
                }
                int owner[NSLOT];
                for (int sl = 0; sl < NSLOT; sl++) owner[sl] = -1;
                std::map<uint32_t, size_t> next;   // producer -> index of its next unconsumed use
                auto next_use = [&](uint32_t p) -> uint32_t {
                    auto& u = uses[p];
                    size_t k = next[p];
                    return k < u.size() ? u[k] : 0xffffffffu;
                };
                std::map<uint32_t, int> slot_of;
                for (uint32_t i = lo; i < hi; i++) {
                    int pos[2];
                    int n = operand_pos(ops[i], pos);
                    for (int q = 0; q < n; q++) {
                        auto it = producer_of.find(ops[i].refs[pos[q]]);
                        if (it == producer_of.end() || it->second >= i) continue;
                        uint32_t pr = it->second;
                        // (a key can be produced twice in a sub-range only if rows repeated, which they do not)
                        auto st = slot_of.find(pr);
                        if (st != slot_of.end() && owner[st->second] == (int)pr) ops[i].flags |= (uint16_t)((st->second + 1) << (10 + 2 * q));
                        auto& u = uses[pr];
                        while (next[pr] < u.size() && u[next[pr]] <= i) next[pr]++;
                    }
                    for (int sl = 0; sl < NSLOT; sl++)
                        if (owner[sl] >= 0 && next_use((uint32_t)owner[sl]) == 0xffffffffu) owner[sl] = -1;
                    if (result_key(ops[i]) != H2E_NO_REF && !uses[i].empty()) {
                        int pick = -1;
                        for (int sl = 0; sl < NSLOT && pick < 0; sl++)
                            if (owner[sl] < 0) pick = sl;
                        if (pick < 0) {
                            int far = 0;
                            for (int sl = 1; sl < NSLOT; sl++)
                                if (next_use((uint32_t)owner[sl]) > next_use((uint32_t)owner[far])) far = sl;
                            if (next_use((uint32_t)owner[far]) > uses[i][0]) pick = far;
                        }
                        if (pick >= 0) {
                            owner[pick] = (int)i;
                            slot_of[i] = pick;
                            ops[i].flags |= (uint16_t)((pick + 1) << 8);
                        }
                    }
                }
                lo = hi;
            }
        }
    }

    void finish() {
        h2e::Recorder& r = *rec;
        if (r.fp.id != r.primary_field) r.use_field(r.primary_field);   // the analyses below decode cut segments in the program's field
        r.close_segment();
        mark_local_results();
        assign_expansion_slots();
        mark_deferrable();
        mark_side_segments();
        // The serial tail of the program: the last cut single-strand segment whose predictor chain already starts early
        // on the side stream (it only needs an earlier segment's predictors - the MSM tail), provided nothing after it
        // forks again.  Its whole value chain, and whatever follows it, runs on the job slot's side stream (run_impl).
        tail_from = -1;
        for (size_t si = 0; si < r.segments.size(); si++) {
            const h2e::Segment& sg = r.segments[si];
            if (sg.tape_end <= sg.tape_begin || sg.n_strands != 1 || sg.n_cuts == 0) continue;
            bool early_chain = false;
            for (auto& pk : r.pre_kernels) early_chain = early_chain || (pk.before_segment == si && pk.early_after_segment >= 0);
            bool forks_later = false;
            for (size_t sj = si + 1; sj < r.segments.size(); sj++)
                forks_later = forks_later || (r.segments[sj].tape_end > r.segments[sj].tape_begin && r.segments[sj].n_strands > 1);
            if (early_chain && !forks_later) {
                tail_from = (int64_t)si;
                break;
            }
        }
        // which of the 8 value-hint slots per ecc op does anything read?  (finalize_ecc skips the others)
        for (auto& pk : r.pre_kernels) {
            if (!pk.k.ecc_ops) continue;
            uint32_t lo = pk.k.hint_base, per = pk.k.hints_per_lane, mask = 1u << H2E_HINT_LAMBDA;
            for (const H2EOp& op : r.tape)
                if ((op.flags & H2E_FLAG_HINTED) && op.imm >= lo && op.imm < lo + per) mask |= 1u << ((op.imm - lo) % H2E_ECC_HINT_SLOTS);
            pk.k.used_slots = mask;
            if (dbg_env("H2E_DUMP_TAPE")) fprintf(stderr, "predictor kind %u: value-hint slots in use: 0x%02x\n", pk.k.kind, mask);
        }
        if (dbg_env("H2E_DUMP_TAPE")) {   // debugging aid: per segment, ops by opcode (alive / skipped by the values replay)
            for (size_t si = 0; si < r.segments.size(); si++) {
                auto& sg = r.segments[si];
                std::map<int, std::array<uint32_t, 4>> h;
                for (uint32_t i = sg.tape_begin; i < sg.tape_end; i++) {
                    auto& e = h[r.tape[i].opcode];
                    e[(r.tape[i].flags & H2E_FLAG_VALUES_SKIP) ? 1 : 0]++;
                    if (!(r.tape[i].flags & H2E_FLAG_VALUES_SKIP) && (r.tape[i].flags & H2E_FLAG_HINTED)) e[2]++;
                    if (!(r.tape[i].flags & H2E_FLAG_VALUES_SKIP) && !(r.tape[i].flags & H2E_FLAG_LOCAL_RESULT)) e[3]++;
                }
                fprintf(stderr, "segment %zu: ops %u strands %u cuts %u fork %d\n", si, sg.tape_end - sg.tape_begin, sg.n_strands, sg.n_cuts, (int)sg.is_fork);
                for (auto& kv : h)
                    fprintf(stderr, "   opcode %2d alive %6u (hinted %6u, stored %6u) skipped %6u\n", kv.first, kv.second[0], kv.second[2], kv.second[3], kv.second[1]);
            }
        }
        base_rows = std::max<uint64_t>(r.base_height, r.base_offset) + 1;
        range_rows = std::max<uint64_t>(r.range_height, r.range_offset) + 1;
        select_rows = std::max<uint64_t>(r.select_height, r.select_offset) + 1;
        if (r.emit_shape) {
            r.base_fix.resize(base_rows * 9, 0);
            r.range_fix.resize(range_rows * 2, 0);
            r.select_fix.resize(select_rows * 2, 0);
            r.base_flags.resize(base_rows * 5, 0);
            r.range_flags.resize(range_rows * 3, 0);
            r.select_flags.resize(select_rows * 2, 0);
            perm_flat.reserve(r.permutations.size() * 2);
            for (auto& p : r.permutations) {
                perm_flat.push_back(p.first);
                perm_flat.push_back(p.second);
            }
            for (auto& f : r.fixed_patches) {
                patch_flat.push_back(f.row);
                patch_flat.push_back(f.col);
                patch_flat.push_back(f.input_slot);
                patch_flat.push_back((uint32_t)f.limb);
            }
        }
    }
};

// Everything one run owns while it is in flight: engine workspace, instance table, events.  A context keeps a small
// ring of these, so that h2e_submit can queue the value chain of run k + 1 (caller's stream) while run k's expansion is
// still streaming on the expansion stream; h2e_run uses the same slots and joins before it returns.
#define H2E_DG_SHARDS 64u
struct JobSlot {
    // engine workspace (grow-only): quotient hints, numerator/denominator pairs, Jacobian scratch, selected points
    uint64_t *ws_hints = nullptr, *ws_nd = nullptr, *ws_jac = nullptr, *ws_sel = nullptr;
    size_t ws_hints_words = 0, ws_nd_words = 0, ws_jac_words = 0, ws_sel_words = 0;
    InstanceDescHost* d_inst = nullptr;
    uint32_t inst_cap = 0;
    uint64_t* dg_shards = nullptr;    // stream digest accumulators of the slot's run: [H2E_DG_SHARDS][3][instances][4] words
    uint32_t dg_cap = 0;              // instances they are sized for
    InstanceDescHost* h_inst = nullptr;   // pinned (hipHostMalloc): the upload below is a real asynchronous copy ...
    hipEvent_t upload_ev = nullptr;       // ... and this event says when the host may rewrite the table
    std::vector<hipEvent_t> ev;       // profiling: 4 per launched segment (value-chain begin/end, expansion begin/end)
    std::vector<hipEvent_t> sync_ev;  // cross-stream dependencies
    hipEvent_t done = nullptr;        // recorded when every stream of the slot's last run has finished
    hipEvent_t order_ev = nullptr;    // the caller's stream at submission (h2e_submit)
    // Side stream of the slot: early predictors, fork segments outside the chain, and the *serial tail* of a program
    // (the MSM tail: a single-wave 13 ms predictor chain + its replay).  Per slot, so that the tail of run k and the
    // value chain of run k + 1 (caller's stream) run side by side - the tail chain is latency-, not throughput-bound.
    hipStream_t side_stream = nullptr;
    // Chain stream of the slot (h2e_submit only): the run's value chain is queued here, ordered after what the caller's
    // stream held at submission, so that the value chains of consecutive runs overlap each other as well - each of their
    // kernels is latency-bound and leaves most of the GPU idle.
    hipStream_t chain_stream = nullptr;
    // start counter of the slot's digit chains and what it will read once every chain launched so far has started (engine.hip h2e_gate)
    uint32_t* d_gate = nullptr;
    uint32_t gate_total = 0;
    bool used = false;
    bool profiled = false;            // the last run on this slot recorded `ev`
    uint32_t n_launches = 0;
    std::vector<uint32_t> x_kernels;  // per launched segment: expansion kernel launches of the last run (2 = split)
    void release() {
        for (auto e : ev) (void)hipEventDestroy(e);
        for (auto e : sync_ev) (void)hipEventDestroy(e);
        if (done) (void)hipEventDestroy(done);
        if (order_ev) (void)hipEventDestroy(order_ev);
        if (side_stream) (void)hipStreamDestroy(side_stream);
        if (chain_stream) (void)hipStreamDestroy(chain_stream);
        (void)hipFree(ws_hints);
        (void)hipFree(ws_nd);
        (void)hipFree(ws_jac);
        (void)hipFree(ws_sel);
        (void)hipFree(d_inst);
        (void)hipFree(dg_shards);
        (void)hipFree(d_gate);
        if (h_inst) (void)hipHostFree(h_inst);
        if (upload_ev) (void)hipEventDestroy(upload_ev);
    }
};

struct h2e_ctx {
    int device;
    H2EFieldConsts* d_fc[3] = {nullptr, nullptr, nullptr};
    std::map<std::string, h2e_program*> cache;
    bool profiling = false;
    static constexpr int N_SLOTS = 16;
    uint32_t depth = 2;      // job slots in use = runs in flight (H2E_OPT_PIPELINE_DEPTH); each slot brings its own streams   // runs in flight (h2e_submit): 2 hide an MSM step's value chain; the pairing checks' 34 ms
                                        // level-parallel chains (one workgroup per instance) want 4
    JobSlot slots[N_SLOTS];
    uint64_t n_runs = 0;     // runs submitted so far: run k uses slot k % N_SLOTS
    int last_slot = -1;
    hipStream_t expand_stream = nullptr;
    hipStream_t fixup_stream = nullptr;
    hipStream_t small_stream = nullptr;   // small expansions of pipelined runs (H2E_SCHED & 4)
    // tuning knobs, read once at h2e_ctx_create (H2E_X_SPLIT, H2E_X_SPLIT_MIN_LANES); h2e_ctx_set_option overrides
    uint32_t x_split_pct = 45;
    // launches a big expansion goes out as (H2E_X_PARTS): the part behind the first x_split_pct percent in parts - 1 equal launches.  The last
    // part's inverse fix-up is the one nothing runs under, and the run is complete - its buffer set free for the run after the next - only
    // behind it: 64 x 1024-point tiles pipelined, alternating in one box: 2 launches 15.38 / 15.43 ms per step, 3: 15.13 / 15.15, 4: 15.11 / 15.18
    uint32_t x_parts = 3;
    uint64_t x_split_min_lanes = 1ull << 21;
    uint64_t small_x_lanes = 1u << 18;   // an expansion with fewer lanes is "small" (H2E_SMALL_X_LANES)
    uint32_t sched = 4;      // scheduling experiments (H2E_SCHED bit mask): 1 = a pipelined run's small fix-ups go to the slot's side
                             // stream, 2 = its small expansions too (instead of queueing on the shared expansion stream)
    int prio_expand = 0, prio_side = 0, prio_fixup = 0;   // HIP stream priorities (H2E_STREAM_PRIORITIES="x,s,f"; lower = higher priority)
    int64_t test_skip_expansion = INT64_MIN;   // test hook (h2e_ctx_set_option): see H2E_OPT_TEST_SKIP_EXPANSION
    uint32_t last_split_segments = 0;          // segments of the last run whose expansion was split (h2e_ctx_get_stat)
    std::mutex mu;                             // h2e_run / h2e_submit on one context are serialised on the host
    // Operator API: programs of the ops recorded so far, keyed by (op, arguments, operand handles, cursors, heights, msm prefix):
    // a records object that repeats an op sequence (the next batch of the same circuit) re-uses them - no host-side recording,
    // no new device tapes.  `outs` = the handles the op returned, byte for byte.
    // The cache is bounded (keys hold value-dependent arguments - constants, offsets - so a long-lived context would otherwise
    // keep one program with its device tapes per distinct key): at `op_cache_cap` entries the least recently used ones that no
    // call is running go (H2E_OP_CACHE_CAP, default 4096; a proving loop's working set is its ops per batch).
    struct OpEntry {
        h2e_program* prog = nullptr;
        std::vector<std::vector<uint8_t>> outs;
        size_t msm_prefix_after = 0;
        uint64_t last_use = 0;
        uint32_t in_use = 0;
    };
    std::map<std::string, OpEntry> op_cache;
    std::mutex op_mu;
    uint64_t op_hits = 0, op_misses = 0, op_tick = 0, op_evictions = 0;
    size_t op_cache_cap = 4096;
    void op_cache_trim() {   // (op_mu held)
        while (op_cache.size() > op_cache_cap) {
            auto victim = op_cache.end();
            for (auto it = op_cache.begin(); it != op_cache.end(); ++it)
                if (it->second.in_use == 0 && (victim == op_cache.end() || it->second.last_use < victim->second.last_use)) victim = it;
            if (victim == op_cache.end()) break;
            delete victim->second.prog;   // (frees its device tapes: hipFree waits for the work that still reads them)
            op_cache.erase(victim);
            op_evictions++;
        }
    }
    ~h2e_ctx() {
        for (auto& kv : op_cache) delete kv.second.prog;
        for (auto& kv : cache) delete kv.second;
        for (int i = 0; i < 3; i++)
            if (d_fc[i]) (void)hipFree(d_fc[i]);
        for (auto& sl : slots) sl.release();
        if (expand_stream) (void)hipStreamDestroy(expand_stream);
        if (fixup_stream) (void)hipStreamDestroy(fixup_stream);
        if (small_stream) (void)hipStreamDestroy(small_stream);
    }
};

extern "C" {

const char* h2e_last_error(void) { return g_last_error.c_str(); }
const char* h2e_version(void) { return "h2e 0.2 (gfx950, batch-interleaved advice)"; }

int h2e_ctx_create(int device, h2e_ctx** out) {
    if (!out) return fail(H2E_ERR_INVALID, "out is null");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) return fail(H2E_ERR_HIP, "no HIP device available: the witness engine has no CPU fallback");
    if (device < 0 || device >= count) return fail(H2E_ERR_INVALID, "bad device index");
    h2e_ctx* c = new h2e_ctx();
    c->device = device;
    // tuning knobs are read once, here (nothing reads the environment while a run is being queued)
    if (const char* e1 = getenv("H2E_X_SPLIT")) c->x_split_pct = (uint32_t)std::max(0, std::min(100, atoi(e1)));
    if (const char* e2 = getenv("H2E_X_SPLIT_MIN_LANES")) c->x_split_min_lanes = (uint64_t)atoll(e2);
    if (const char* e2b = getenv("H2E_X_PARTS")) c->x_parts = (uint32_t)std::max(2, std::min(8, atoi(e2b)));
    if (const char* e4 = getenv("H2E_TUNE")) {   // "reserve,xcache,xpad,scan test mask,persistent workgroups per CU,no packed expansion" (engine.hip g_tune)
        int a = 0, b = 0, d = 0, t = 0, pw = 0, nopack = 0;
        sscanf(e4, "%d,%d,%d,%d,%d,%d", &a, &b, &d, &t, &pw, &nopack);
        h2e_engine_set_tuning(4, pw);   // persistent expansion: workgroups per CU (experiment)
        h2e_engine_set_tuning(5, nopack);   // 1: batches smaller than a wave through the plain expansion kernel (A/B)
        h2e_engine_set_tuning(0, a);
        h2e_engine_set_tuning(1, b);
        h2e_engine_set_tuning(2, d);
        if (t) {
            (void)hipSetDevice(device);
            h2e_engine_set_tuning(3, t);   // scan predictor test mask (H2E_OPT_TEST_SCAN_FALLBACK)
        }
    }
    if (const char* e5 = getenv("H2E_SCHED")) c->sched = (uint32_t)atoi(e5);
    if (const char* e8 = getenv("H2E_OP_CACHE_CAP")) c->op_cache_cap = (size_t)std::max(1, atoi(e8));
    if (const char* e7 = getenv("H2E_SMALL_X_LANES")) c->small_x_lanes = (uint64_t)atoll(e7);
    if (const char* e3 = getenv("H2E_STREAM_PRIORITIES")) sscanf(e3, "%d,%d,%d", &c->prio_expand, &c->prio_side, &c->prio_fixup);
    *out = c;
    return 0;
}
void h2e_ctx_destroy(h2e_ctx* ctx) { delete ctx; }
void h2e_program_destroy(h2e_program* p) { delete p; }

static int new_program(int fp, int emit_shape, h2e_program** out, h2e_program*& p) {
    if (!out) return fail(H2E_ERR_INVALID, "out is null");
    if (fp < 0 || fp > 2) return fail(H2E_ERR_INVALID, "bad field pair");
    p = new h2e_program();
    p->field_pair = fp;
    p->rec.reset(new h2e::Recorder(field_pair(fp)));
    p->rec->emit_shape = emit_shape != 0;
    return 0;
}
#define GUARDED(...)                                    \
    try {                                               \
        __VA_ARGS__                                     \
    } catch (std::exception & e) {                      \
        delete p;                                       \
        return fail(H2E_ERR_SHAPE, e.what());           \
    }

int h2e_program_int_mul_batch(int fp, uint32_t n, int emit_shape, h2e_program** out) {
    h2e_program* p = nullptr;
    int rc = new_program(fp, emit_shape, out, p);
    if (rc) return rc;
    GUARDED({
        h2e::Recorder& r = *p->rec;
        uint32_t s0 = r.alloc_inputs(2 * n);
        r.fork(n, 2, [&](uint32_t) {
            h2e::AssignedInteger a = r.assign_w(s0, true);
            h2e::AssignedInteger b = r.assign_w(s0 + 1, true);
            r.int_mul(a, b);
        });
        p->finish();
    })
    *out = p;
    return 0;
}

int h2e_program_integer_chip_st(int fp, int emit_shape, h2e_program** out) {
    h2e_program* p = nullptr;
    int rc = new_program(fp, emit_shape, out, p);
    if (rc) return rc;
    GUARDED({
        h2e::Recorder& r = *p->rec;
        uint32_t s = r.alloc_inputs(6);
        h2e::AssignedInteger a = r.assign_w(s + 0), b = r.assign_w(s + 1);
        h2e::AssignedInteger c1 = r.assign_w(s + 2);
        h2e::AssignedInteger c2 = r.int_add(a, b);
        r.assert_int_equal(c1, c2);
        h2e::AssignedInteger d1 = r.assign_w(s + 3);
        h2e::AssignedInteger d2 = r.int_sub(a, b);
        r.assert_int_equal(d1, d2);
        h2e::AssignedInteger e1 = r.assign_w(s + 4);
        h2e::AssignedInteger e2 = r.int_mul(a, b);
        r.assert_int_equal(e1, e2);
        h2e::AssignedInteger f1 = r.assign_w(s + 5);
        h2e::AssignedInteger f2 = r.int_div(a, b).second;
        r.assert_int_equal(f1, f2);
        h2e::AssignedInteger zero = r.int_sub(a, a);
        auto g = r.int_div(a, zero);
        r.assert_true(g.first);
        p->finish();
    })
    *out = p;
    return 0;
}

static int program_msm_bn256_tile(uint32_t n, int emit_shape, bool with_select, h2e_program** out);
int h2e_program_msm_bn256_tile(uint32_t n, int emit_shape, h2e_program** out) { return program_msm_bn256_tile(n, emit_shape, true, out); }
int h2e_program_msm_bn256_tile_no_select(uint32_t n, int emit_shape, h2e_program** out) {
    return program_msm_bn256_tile(n, emit_shape, false, out);
}
static int program_msm_bn256_tile(uint32_t n, int emit_shape, bool with_select, h2e_program** out) {
    h2e_program* p = nullptr;
    if (n == 0) return fail(H2E_ERR_INVALID, "n_points must be > 0");
    int rc = new_program(H2E_FIELD_BN256_FQ, emit_shape, out, p);
    if (rc) return rc;
    GUARDED({
        h2e::Recorder& r = *p->rec;
        uint32_t s = r.alloc_inputs(4 * n + 9);
        h2e::NativeScalarEccContext ecc(r, h2e::bn256_g1_params(), 0);
        ecc.with_select = with_select;
        h2e::NativeScalarEccContext::MsmInputs mi{s + 4 * n + 2, s + 4 * n + 3, s + 4 * n + 4, s + 4 * n + 5};
        h2e::AssignedPoint res = ecc.msm_unsafe_from_inputs(n, s, mi, s + 4 * n, s + 4 * n + 1);
        h2e::AssignedPoint res_expect = ecc.assign_point(h2e::PointInput{s + 4 * n + 6, s + 4 * n + 7, s + 4 * n + 8, false});
        ecc.ecc_assert_equal(res, res_expect);
        for (int i = 0; i < r.fp.limbs; i++) r.outputs.push_back(res.x.limbs_le[i]);
        r.outputs.push_back(res.x.native);
        for (int i = 0; i < r.fp.limbs; i++) r.outputs.push_back(res.y.limbs_le[i]);
        r.outputs.push_back(res.y.native);
        r.outputs.push_back(res.z.v.ref);
        p->finish();
    })
    *out = p;
    return 0;
}

// body of test_bls12_381_ecc_chip_over_bn256_fr (src/tests/general_scalar_ecc_chip.rs:14-49) for one tile of n points:
// GeneralScalarEccContext<bls12_381::G1Affine, bn256::Fr> - points over the 4-limb bls12_381 Fq, scalars as integers of
// the other integer context (3-limb bls12_381 Fr), 3 x 108 = 324 one-bit windows (general_scalar_ecc_chip.rs:96-147)
int h2e_program_msm_bls12_381_tile(uint32_t n, int emit_shape, h2e_program** out) {
    h2e_program* p = nullptr;
    if (n == 0) return fail(H2E_ERR_INVALID, "n_points must be > 0");
    int rc = new_program(H2E_FIELD_BLS12_381_FQ, emit_shape, out, p);
    if (rc) return rc;
    GUARDED({
        h2e::Recorder& r = *p->rec;
        uint32_t s = r.alloc_inputs(4 * n + 9);
        h2e::NativeScalarEccContext ecc(r, h2e::bls12_381_g1_params(), 0);
        ecc.scalar_field = H2E_FIELD_BLS12_381_FR;
        h2e::NativeScalarEccContext::MsmInputs mi{s + 4 * n + 2, s + 4 * n + 3, s + 4 * n + 4, s + 4 * n + 5};
        h2e::AssignedPoint res = ecc.msm_unsafe_from_inputs(n, s, mi, s + 4 * n, s + 4 * n + 1);
        h2e::AssignedPoint res_expect = ecc.assign_point(h2e::PointInput{s + 4 * n + 6, s + 4 * n + 7, s + 4 * n + 8, false});
        ecc.ecc_assert_equal(res, res_expect);
        for (int i = 0; i < r.fp.limbs; i++) r.outputs.push_back(res.x.limbs_le[i]);
        r.outputs.push_back(res.x.native);
        for (int i = 0; i < r.fp.limbs; i++) r.outputs.push_back(res.y.limbs_le[i]);
        r.outputs.push_back(res.y.native);
        r.outputs.push_back(res.z.v.ref);
        p->finish();
    })
    *out = p;
    return 0;
}

// ops per expansion sub-range of the pairing programs (a cut wherever the recorder allows one after that many ops);
// H2E_PAIRING_CUT overrides it when the program is recorded (experiments)
// bn256: 16 (64 checks per GPU fill the waves; finer cuts only add hint-store work to the pipelined step: 3.26 -> 3.38 ms);
// bls12_381: 8 (16 checks per GPU: waves of two sub-ranges, twice as many of them - expansion 1.55 -> 1.30 ms, 2 checks: 1.34 -> 0.52)
// how finely a pairing is cut into launches (recorder_pairing.hpp PairingOps::stage_splits); H2E_PAIRING_SPLITS overrides it when
// the program is recorded
// Default 1: Miller loop | final exponentiation.  64 x bn256: one batch alone 5.29 -> 4.19 ms, pipelined 3.46 -> 3.11 ms; 16 x
// bls12_381: 4.29 -> 3.61 ms alone, pipelined unchanged within noise (2.12 / 2.27).  Finer cuts (2: after each exponentiation
// by x) gain little more alone and lose pipelined (3.29 / 2.72 ms): a chain next to its own context's expansion runs at about
// 60 % of its rate alone (profiles/r4_*), so the overlap pays back only part of what it hides.
static int pairing_stage_splits() {
    if (const char* e = getenv("H2E_PAIRING_SPLITS")) return std::max(0, atoi(e));
    return 1;
}
static uint32_t pairing_cut_every(int curve) {
    if (const char* e = getenv("H2E_PAIRING_CUT")) return (uint32_t)std::max(2, atoi(e));
    return curve == 0 ? 16 : 8;
}
int h2e_program_pairing_check_bn256(int emit_shape, h2e_program** out) {
    h2e_program* p = nullptr;
    int rc = new_program(H2E_FIELD_BN256_FQ, emit_shape, out, p);
    if (rc) return rc;
    GUARDED({
        h2e::Recorder& r = *p->rec;
        r.auto_cut_every = pairing_cut_every(0);
        uint32_t s = r.alloc_inputs(10);
        h2e::NativeScalarEccContext ecc(r, h2e::bn256_g1_params(), 0);
        h2e::Bn256PairingOps po(r);
        po.stage_splits = pairing_stage_splits();
        r.begin_field_hints();
        h2e::AssignedFq2 bx{r.assign_int_constant_input(s + 0), r.assign_int_constant_input(s + 1)};
        h2e::AssignedFq2 by{r.assign_int_constant_input(s + 2), r.assign_int_constant_input(s + 3)};
        h2e::AssignedG2Affine B{bx, by, h2e::AssignedCondition{r.assign_constant_u64(0)}};
        h2e::AssignedPoint neg_a = ecc.assign_point(h2e::PointInput{s + 4, s + 5, s + 6, false});
        h2e::AssignedPoint a = ecc.assign_point(h2e::PointInput{s + 7, s + 8, s + 9, false});
        po.check_pairing({h2e::PairingOps::Term(&a, &B), h2e::PairingOps::Term(&neg_a, &B)});
        r.end_field_hints();
        p->finish();
    })
    *out = p;
    return 0;
}

int h2e_program_pairing_check_bls12_381(int emit_shape, h2e_program** out) {
    h2e_program* p = nullptr;
    int rc = new_program(H2E_FIELD_BLS12_381_FQ, emit_shape, out, p);
    if (rc) return rc;
    GUARDED({
        h2e::Recorder& r = *p->rec;
        r.auto_cut_every = pairing_cut_every(1);
        uint32_t s = r.alloc_inputs(14);
        h2e::NativeScalarEccContext ecc(r, h2e::bls12_381_g1_params(), 0);  // EccChipBaseOps of GeneralScalarEccContext
        h2e::Bls12381PairingOps po(r);
        po.stage_splits = pairing_stage_splits();
        r.begin_field_hints();
        h2e::AssignedFq2 bx{r.assign_int_constant_input(s + 0), r.assign_int_constant_input(s + 1)};
        h2e::AssignedFq2 by{r.assign_int_constant_input(s + 2), r.assign_int_constant_input(s + 3)};
        h2e::AssignedG2Affine B{bx, by, h2e::AssignedCondition{r.assign_constant_u64(0)}};
        h2e::AssignedFq2 bcx{r.assign_int_constant_input(s + 4), r.assign_int_constant_input(s + 5)};
        h2e::AssignedFq2 bcy{r.assign_int_constant_input(s + 6), r.assign_int_constant_input(s + 7)};
        h2e::AssignedG2Affine BC{bcx, bcy, h2e::AssignedCondition{r.assign_constant_u64(0)}};
        h2e::AssignedPoint neg_a = ecc.assign_point(h2e::PointInput{s + 8, s + 9, s + 10, false});
        h2e::AssignedPoint ac = ecc.assign_point(h2e::PointInput{s + 11, s + 12, s + 13, false});
        po.check_pairing({h2e::PairingOps::Term(&ac, &B), h2e::PairingOps::Term(&neg_a, &BC)});
        r.end_field_hints();
        p->finish();
    })
    *out = p;
    return 0;
}

// pairing(terms) [== expected]: the first block of the reference's pairing tests
// (src/tests/native_scalar_pairing_chip.rs:20-65 with one pair, general_scalar_pairing_chip.rs:20-72 with two)
int h2e_program_pairing(int curve, uint32_t n_pairs, int with_expected, int emit_shape, h2e_program** out) {
    h2e_program* p = nullptr;
    if (curve != 0 && curve != 1) return fail(H2E_ERR_INVALID, "curve must be 0 (bn256) or 1 (bls12_381)");
    if (n_pairs == 0 || n_pairs > 8) return fail(H2E_ERR_INVALID, "n_pairs must be 1..8");
    int rc = new_program(curve == 0 ? H2E_FIELD_BN256_FQ : H2E_FIELD_BLS12_381_FQ, emit_shape, out, p);
    if (rc) return rc;
    GUARDED({
        h2e::Recorder& r = *p->rec;
        r.auto_cut_every = pairing_cut_every(curve);
        uint32_t s = r.alloc_inputs(7 * n_pairs + (with_expected ? 12 : 0));
        h2e::NativeScalarEccContext ecc(r, curve == 0 ? h2e::bn256_g1_params() : h2e::bls12_381_g1_params(), 0);
        std::unique_ptr<h2e::PairingOps> po;
        if (curve == 0) po.reset(new h2e::Bn256PairingOps(r));
        else po.reset(new h2e::Bls12381PairingOps(r));
        po->stage_splits = pairing_stage_splits();
        std::vector<h2e::AssignedG2Affine> g2;
        r.begin_field_hints();
        for (uint32_t k = 0; k < n_pairs; k++) {
            h2e::AssignedFq2 x{r.assign_int_constant_input(s + 4 * k + 0), r.assign_int_constant_input(s + 4 * k + 1)};
            h2e::AssignedFq2 y{r.assign_int_constant_input(s + 4 * k + 2), r.assign_int_constant_input(s + 4 * k + 3)};
            g2.push_back(h2e::AssignedG2Affine{x, y, h2e::AssignedCondition{r.assign_constant_u64(0)}});
        }
        uint32_t e0 = s + 4 * n_pairs;
        h2e::AssignedFq12 expected;
        if (with_expected) {   // fq12_assign_constant: c0.c0.c0, c0.c0.c1, c0.c1.c0, ... (fq12.rs:453-458)
            h2e::AssignedInteger v[12];
            for (int i = 0; i < 12; i++) v[i] = r.assign_int_constant_input(e0 + i);
            expected = h2e::AssignedFq12{h2e::AssignedFq6{{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}},
                                         h2e::AssignedFq6{{v[6], v[7]}, {v[8], v[9]}, {v[10], v[11]}}};
        }
        uint32_t p0 = e0 + (with_expected ? 12 : 0);
        std::vector<h2e::AssignedPoint> g1;
        for (uint32_t k = 0; k < n_pairs; k++) g1.push_back(ecc.assign_point(h2e::PointInput{p0 + 3 * k, p0 + 3 * k + 1, p0 + 3 * k + 2, false}));
        std::vector<h2e::PairingOps::Term> terms;
        for (uint32_t k = 0; k < n_pairs; k++) terms.push_back(h2e::PairingOps::Term(&g1[k], &g2[k]));
        h2e::AssignedFq12 res = po->pairing(terms);
        if (with_expected) po->fq12_assert_eq(expected, res);
        const h2e::AssignedFq2* parts[6] = {&res.c0.c0, &res.c0.c1, &res.c0.c2, &res.c1.c0, &res.c1.c1, &res.c1.c2};
        for (auto* f2 : parts)
            for (const h2e::AssignedInteger* a : {&f2->c0, &f2->c1}) {
                for (int i = 0; i < r.fp.limbs; i++) r.outputs.push_back(a->limbs_le[i]);
                r.outputs.push_back(a->native);
            }
        r.end_field_hints();
        p->finish();
    })
    *out = p;
    return 0;
}

int h2e_program_shape(const h2e_program* p, h2e_shape* out) {
    if (!p || !out) return fail(H2E_ERR_INVALID, "null argument");
    const h2e::Recorder& r = *p->rec;
    std::memset(out, 0, sizeof(*out));
    out->field_pair = p->field_pair;
    out->slot_words = r.fp.w_words;
    out->n_input_slots = r.n_input_slots;
    out->base_offset = r.base_offset;
    out->range_offset = r.range_offset;
    out->select_offset = r.select_offset;
    out->base_height = r.base_height;
    out->range_height = r.range_height;
    out->select_height = r.select_height;
    out->base_rows = p->base_rows;
    out->range_rows = p->range_rows;
    out->select_rows = p->select_rows;
    out->n_advice_cells = r.n_advice_cells;
    out->n_permutations = r.permutations.size();
    out->n_dict = r.dict.size();
    out->n_fixed_patches = r.fixed_patches.size();
    uint32_t nseg = 0;
    for (auto& s : r.segments)
        if (s.tape_end > s.tape_begin) nseg++;
    out->n_segments = nseg;
    out->n_ops = r.tape.size();
    if (r.emit_shape) {
        out->dict = (const uint64_t*)r.dict.data();
        out->base_fix = r.base_fix.data();
        out->range_fix = r.range_fix.data();
        out->select_fix = r.select_fix.data();
        out->base_flags = r.base_flags.data();
        out->range_flags = r.range_flags.data();
        out->select_flags = r.select_flags.data();
        out->permutations = p->perm_flat.data();
        out->fixed_patches = p->patch_flat.data();
    }
    return 0;
}

// Order tables of the packed expansion (tape.h H2ELaunch::pk_order).  A wave of h2e_run_tape_packed takes G sub-ranges and every
// step runs ONE opcode for the groups whose cursor shows it, so a wave of G different opcode sequences costs up to the sum of
// them.  The programs it serves repeat themselves (a pairing check: 8 649 sub-ranges, 391 different opcode sequences), so the
// sub-ranges are classed by their sequence, every wave takes sub-ranges of one class (a class's last wave is padded with empty
// slots), and the heaviest waves are dispatched first.  Replayed on the real tapes (exp/pack_sim.py): the longest SIMD's work
// falls 2.0-2.9 x against taking the sub-ranges in tape order.
static void pack_orders_of(const h2e::Recorder& r, const h2e::Segment& sg, const uint32_t* subs, uint32_t n_sub, std::vector<uint32_t>& out,
                           std::array<uint32_t, 5>& off, std::array<uint32_t, 5>& n_waves) {
    // what an op costs the wave ~ the cells it writes
    auto op_cost = [](uint16_t opc) -> uint32_t {
        switch (opc) {
            case H2E_OP_DIV_CORE: return 140;
            case H2E_OP_INT_MUL: return 125;
            case H2E_OP_REDUCE: case H2E_OP_IS_INT_ZERO: return 40;
            case H2E_OP_ASSIGN_W: case H2E_OP_DECOMPOSE_NATIVE: return 23;
            case H2E_OP_BISEC_INT: case H2E_OP_SELECT_POINT: return 20;
            case H2E_OP_INT_ADD: case H2E_OP_INT_SUB: case H2E_OP_MASK_INT: return 13;
            case H2E_OP_INT_NEG: case H2E_OP_INT_MUL_SMALL: case H2E_OP_CACHE_INT: return 10;
            default: return 4;
        }
    };
    struct Class { uint64_t cost; std::vector<uint32_t> members; };
    std::vector<Class> classes;
    std::unordered_map<uint64_t, std::vector<uint32_t>> by_hash;   // hash -> classes with it (compared op by op: a collision must not mix sequences)
    auto same_sequence = [&](uint32_t a, uint32_t b) {
        if (subs[a + 1] - subs[a] != subs[b + 1] - subs[b]) return false;
        for (uint32_t i = 0; i < subs[a + 1] - subs[a]; i++)
            if (r.tape[sg.tape_begin + subs[a] + i].opcode != r.tape[sg.tape_begin + subs[b] + i].opcode) return false;
        return true;
    };
    for (uint32_t k = 0; k < n_sub; k++) {
        uint64_t h = 0xcbf29ce484222325ull, cost = 0;
        for (uint32_t o = subs[k]; o < subs[k + 1]; o++) {
            uint16_t opc = r.tape[sg.tape_begin + o].opcode;
            h = (h ^ opc) * 0x100000001b3ull;
            cost += op_cost(opc);
        }
        std::vector<uint32_t>& cand = by_hash[h];
        uint32_t cls = ~0u;
        for (uint32_t c : cand)
            if (same_sequence(classes[c].members[0], k)) cls = c;
        if (cls == ~0u) {
            cls = (uint32_t)classes.size();
            cand.push_back(cls);
            classes.push_back({cost, {}});
        }
        classes[cls].members.push_back(k);
    }
    std::vector<uint32_t> by_cost(classes.size());
    for (uint32_t c = 0; c < classes.size(); c++) by_cost[c] = c;
    std::stable_sort(by_cost.begin(), by_cost.end(), [&](uint32_t a, uint32_t b) { return classes[a].cost > classes[b].cost; });
    for (int k = 0; k < 5; k++) {
        const uint32_t G = 2u << k;
        off[k] = (uint32_t)out.size();
        uint32_t waves = 0;
        for (uint32_t c : by_cost) {
            const std::vector<uint32_t>& m = classes[c].members;
            for (size_t i = 0; i < m.size(); i += G, waves++)
                for (uint32_t g = 0; g < G; g++) out.push_back(i + g < m.size() ? m[i + g] : ~0u);
        }
        n_waves[k] = waves;
    }
}
static void build_pack_orders(h2e_program* p) {
    const h2e::Recorder& r = *p->rec;
    p->h_pk_order.clear();
    p->seg_pk_off.assign(r.segments.size(), std::array<uint32_t, 5>{0, 0, 0, 0, 0});
    p->seg_pk_waves.assign(r.segments.size(), std::array<uint32_t, 5>{0, 0, 0, 0, 0});
    for (size_t si = 0; si < r.segments.size(); si++) {
        const h2e::Segment& sg = r.segments[si];
        uint32_t n_sub = p->seg_n_sub[si];
        if (n_sub < 2 || sg.n_strands > 32 || n_sub > (1u << 18)) continue;   // (a packed launch has at most 32 lanes per sub-range)
        pack_orders_of(r, sg, p->h_subs.data() + p->seg_sub_begin[si], n_sub, p->h_pk_order, p->seg_pk_off[si], p->seg_pk_waves[si]);
    }
}

// The order tables only matter to launches with n_strands x n_instances <= 32 lanes per sub-range (engine.hip: the packed
// expansion).  BASELINE-sized batches, the MSM's segments and most of an operator-API context's cached op programs never take
// that path, so the tables (an op-by-op classification of the tape, five tables per segment, a device allocation) are made
// by the first run that does.
static int ensure_pack_orders(h2e_program* p, uint32_t n_instances) {
    if (p->pk_built) return 0;
    const h2e::Recorder& r = *p->rec;
    bool need = false;
    for (size_t si = 0; si < r.segments.size() && si < p->seg_n_sub.size(); si++)
        need = need || (p->seg_n_sub[si] >= 2 && (uint64_t)r.segments[si].n_strands * n_instances <= 32);
    if (!need) return 0;
    build_pack_orders(p);
    if (!p->h_pk_order.empty()) {
        HIP_TRY(hipMalloc((void**)&p->d_pk_order, p->h_pk_order.size() * 4));
        HIP_TRY(hipMemcpy(p->d_pk_order, p->h_pk_order.data(), p->h_pk_order.size() * 4, hipMemcpyHostToDevice));
    }
    p->pk_built = true;
    return 0;
}

static int ensure_device_program(h2e_ctx* ctx, h2e_program* p) {
    if (p->device == ctx->device) return 0;
    if (p->device >= 0) return fail(H2E_ERR_INVALID, "program already bound to another device");
    h2e::Recorder& r = *p->rec;
    HIP_TRY(hipSetDevice(ctx->device));
    auto up = [&](void** d, const void* h, size_t bytes) -> hipError_t {
        if (bytes == 0) bytes = 16;
        hipError_t e = hipMalloc(d, bytes);
        if (e != hipSuccess) return e;
        if (h) return hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
        return hipSuccess;
    };
    HIP_TRY(up((void**)&p->d_tape, r.tape.empty() ? nullptr : r.tape.data(), r.tape.size() * sizeof(H2EOp)));
    HIP_TRY(up((void**)&p->d_aux, r.aux.empty() ? nullptr : r.aux.data(), r.aux.size() * 4));
    HIP_TRY(up((void**)&p->d_pool, r.pool.empty() ? nullptr : r.pool.data(), r.pool.size() * 8));
    HIP_TRY(up((void**)&p->d_params, r.params.empty() ? nullptr : r.params.data(), r.params.size() * 4));
    HIP_TRY(up((void**)&p->d_fixups, r.fixups.empty() ? nullptr : r.fixups.data(), r.fixups.size() * 4));
    HIP_TRY(up((void**)&p->d_pre_args, r.pre_args.empty() ? nullptr : r.pre_args.data(), r.pre_args.size() * 4));
    p->h_subs.clear();
    p->seg_sub_begin.assign(r.segments.size(), ~0u);
    p->seg_n_sub.assign(r.segments.size(), 0);
    for (size_t si = 0; si < r.segments.size(); si++) {
        const h2e::Segment& sg = r.segments[si];
        uint32_t n_ops = sg.tape_end - sg.tape_begin;
        if (sg.n_cuts == 0 || n_ops == 0) continue;
        p->seg_sub_begin[si] = (uint32_t)p->h_subs.size();
        p->h_subs.push_back(0);
        uint32_t n = 0;
        for (uint32_t k = 0; k < sg.n_cuts; k++) {
            uint32_t at = r.cuts[sg.cuts_begin + k];
            if (at > p->h_subs.back() && at < n_ops) {
                p->h_subs.push_back(at);
                n++;
            }
        }
        p->h_subs.push_back(n_ops);
        p->seg_n_sub[si] = n + 1;
    }
    HIP_TRY(up((void**)&p->d_subs, p->h_subs.empty() ? nullptr : p->h_subs.data(), p->h_subs.size() * 4));
    // (the packed expansion's order tables are built by the first run that takes the packed path: ensure_pack_orders)
    HIP_TRY(up((void**)&p->d_vtape, p->h_vtape.empty() ? nullptr : p->h_vtape.data(), p->h_vtape.size() * sizeof(H2EVRec)));
    HIP_TRY(up((void**)&p->d_vpieces, p->h_vpieces.empty() ? nullptr : p->h_vpieces.data(), p->h_vpieces.size() * 4));
    HIP_TRY(up((void**)&p->d_lrecs, p->h_lrecs.empty() ? nullptr : p->h_lrecs.data(), p->h_lrecs.size() * sizeof(H2EVRec)));
    HIP_TRY(up((void**)&p->d_lrefs, p->h_lrefs.empty() ? nullptr : p->h_lrefs.data(), p->h_lrefs.size() * 4));
    HIP_TRY(up((void**)&p->d_lrounds, p->h_lrounds.empty() ? nullptr : p->h_lrounds.data(), p->h_lrounds.size() * 4));
    HIP_TRY(up((void**)&p->d_swords, p->h_swords.empty() ? nullptr : p->h_swords.data(), p->h_swords.size() * 4));
    HIP_TRY(up((void**)&p->d_soffsets, p->h_soffsets.empty() ? nullptr : p->h_soffsets.data(), p->h_soffsets.size() * 4));
    HIP_TRY(up((void**)&p->d_sktab, p->h_sktab.empty() ? nullptr : p->h_sktab.data(), p->h_sktab.size() * 8));
    HIP_TRY(up((void**)&p->d_sext, p->h_sext.empty() ? nullptr : p->h_sext.data(), p->h_sext.size() * 4));
    p->device = ctx->device;
    return 0;
}

// One run.  `join` = true: the caller's stream completes when every stream of the run has (h2e_run); false: the
// caller's stream only carries the value chain and `slot.done` is recorded on the fix-up stream when the run is
// complete (h2e_submit / h2e_wait).
// kind: 0 expansion, 1 value chain / side, 2 fix-up
// (CU masks for the value-chain streams - H2E_CU_RESERVE, rounds 3 and 4 - were a measured loser and are gone: a CU-masked expansion
// stream is slow in itself, 21.5 ms per MSM step with 8 CUs set aside, and masked streams of the pairing batches did not overlap at all)
static hipError_t make_stream(h2e_ctx* ctx, hipStream_t* out, int prio, int kind) {
    (void)ctx;
    (void)kind;
    return hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio);
}

static int run_impl(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
                    void* d_select, void* d_status, hipStream_t stream, bool join, int* slot_out, void* d_digests = nullptr) {
    if (!ctx || !p) return fail(H2E_ERR_INVALID, "null ctx/program");
    if (!d_inputs || !d_base || !d_range || !d_select || !d_status) return fail(H2E_ERR_INVALID, "null device pointer");
    std::lock_guard<std::mutex> guard(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_device_program(ctx, p);
    if (rc) return rc;
    if ((rc = ensure_pack_orders(p, n_instances))) return rc;
    int fp = p->field_pair;
    h2e::Recorder& r = *p->rec;
    {   // constants of every W field the program's segments work in (a GeneralScalarEccContext has two)
        bool need[3] = {false, false, false};
        need[fp] = true;
        for (auto& sg : r.segments) need[sg.field_pair] = true;
        for (int f = 0; f < 3; f++)
            if (need[f] && !ctx->d_fc[f]) {
                HIP_TRY(hipMalloc((void**)&ctx->d_fc[f], sizeof(H2EFieldConsts)));
                HIP_TRY(hipMemcpy(ctx->d_fc[f], &field_pair(f).fc, sizeof(H2EFieldConsts), hipMemcpyHostToDevice));
                HIP_TRY((hipError_t)h2e_engine_set_consts(f, &field_pair(f).fc));
            }
    }
    if (!ctx->expand_stream) HIP_TRY(make_stream(ctx, &ctx->expand_stream, ctx->prio_expand, 0));
    if (!ctx->fixup_stream) HIP_TRY(make_stream(ctx, &ctx->fixup_stream, ctx->prio_fixup, 2));
    // Streams.  The *value chain* (predictor kernels + values-only replay, or the plain tape for segments without cuts)
    // runs on the caller's stream: it is what later segments depend on.  The full expansion of a cut segment only needs
    // the value chain up to that segment, so it runs on a second stream and overlaps the value chain of the following
    // segments (and, with h2e_submit, of the following run); predictors that only depend on earlier predictors, fork
    // segments outside the chain and the program's serial tail on the slot's side stream; inverse fix-ups on a fourth.
    // (The runtime maps a process's streams onto GPU_MAX_HW_QUEUES = 4 hardware queues by default; streams that share a
    // queue serialise.  A host that pipelines runs should raise it to 8 before HIP initialises - bench.py does.)
    int slot_index = (int)(ctx->n_runs % ctx->depth);
    JobSlot& J = ctx->slots[slot_index];
    {   // the side stream only exists for programs that use it (every stream takes one of the process's hardware queues)
        bool need_side = p->tail_from >= 0;
        for (auto& pk : r.pre_kernels) need_side = need_side || pk.early_after_segment >= 0;
        for (size_t si = 0; si < r.segments.size() && si < p->seg_side_dep.size(); si++) need_side = need_side || p->seg_side_dep[si] != -2;
        if (need_side && !J.side_stream) HIP_TRY(make_stream(ctx, &J.side_stream, ctx->prio_side, 1));
    }
    if (!join && !J.chain_stream) HIP_TRY(make_stream(ctx, &J.chain_stream, ctx->prio_side, 1));
    if (!join) {   // the chain stream takes over from the caller's stream at this point
        if (!J.order_ev) HIP_TRY(hipEventCreateWithFlags(&J.order_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(J.order_ev, stream));
        HIP_TRY(hipStreamWaitEvent(J.chain_stream, J.order_ev, 0));
    }
    const hipStream_t sa_main = join ? stream : J.chain_stream;
    hipStream_t sa = sa_main, sb = ctx->expand_stream, sc = J.side_stream, sd = ctx->fixup_stream;
#ifdef H2E_DEBUG_HOOKS
    if (FILE* f = dbg_log_file()) {
        fprintf(f, "run %llu begin slot %d join %d streams chain %p expand %p side %p fixup %p small %p\n", (unsigned long long)ctx->n_runs + 1, slot_index, join ? 1 : 0,
                (void*)sa, (void*)sb, (void*)sc, (void*)sd, (void*)ctx->small_stream);
        fflush(f);
    }
#endif
    ctx->n_runs++;
    ctx->last_slot = slot_index;
    if (slot_out) *slot_out = slot_index;
    if (!J.done) HIP_TRY(hipEventCreateWithFlags(&J.done, hipEventDisableTiming));
    // the slot's previous run (two submissions ago) must be complete before its workspace is overwritten
    if (J.used) HIP_TRY(hipStreamWaitEvent(sa, J.done, 0));
    J.used = true;
    // instance descriptors
    if (J.inst_cap < n_instances) {
        if (J.d_inst) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipFree(J.d_inst));
            J.d_inst = nullptr;
        }
        HIP_TRY(hipMalloc((void**)&J.d_inst, (size_t)n_instances * sizeof(InstanceDescHost)));
        if (J.h_inst) HIP_TRY(hipHostFree(J.h_inst));
        J.h_inst = nullptr;
        HIP_TRY(hipHostMalloc((void**)&J.h_inst, (size_t)n_instances * sizeof(InstanceDescHost), hipHostMallocDefault));
        J.inst_cap = n_instances;
    }
    // the table is uploaded from pinned memory by an asynchronous copy: the copy of the slot's previous run (ring depth
    // submissions ago) must have read it before the host writes the new one - a host wait that never waits in a pipeline
    if (!J.upload_ev) HIP_TRY(hipEventCreateWithFlags(&J.upload_ev, hipEventDisableTiming));
    else HIP_TRY(hipEventSynchronize(J.upload_ev));
    size_t slot_words = r.fp.w_words;
    // workspace
    auto grow = [&](uint64_t** buf, size_t* have, size_t need) -> hipError_t {
        if (need <= *have) return hipSuccess;
        if (*buf) {
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) return e;
            (void)hipFree(*buf);
            *buf = nullptr;
        }
        hipError_t e = hipMalloc((void**)buf, need * 8);
        if (e == hipSuccess) *have = need;
        return e;
    };
    // the value chain's workspace is instance-minor ([slot][instance][w words], engine.hip InstanceDesc); a slot holds a
    // value of the widest W field the program works in
    size_t wsw = slot_words;
    for (auto& sg : r.segments) wsw = std::max<size_t>(wsw, (size_t)field_pair(sg.field_pair).w_words);
    size_t hint_words = ((size_t)r.n_hint_slots + H2E_ECC_HINT_SLOTS) * wsw,  // spare: the replay prefetches slot + 8
           nd_words = hint_words * 2,
           jac_words = (size_t)r.n_jac_slots * 3 * wsw,
           sel_words = (size_t)r.n_sel_slots * H2E_SEL_SLOTS * wsw;
    HIP_TRY(grow(&J.ws_hints, &J.ws_hints_words, std::max<size_t>(1, hint_words * n_instances)));
    HIP_TRY(grow(&J.ws_nd, &J.ws_nd_words, std::max<size_t>(1, nd_words * n_instances)));
    HIP_TRY(grow(&J.ws_jac, &J.ws_jac_words, std::max<size_t>(1, jac_words * n_instances)));
    HIP_TRY(grow(&J.ws_sel, &J.ws_sel_words, std::max<size_t>(1, sel_words * n_instances)));
    for (uint32_t i = 0; i < n_instances; i++) {
        InstanceDescHost& d = J.h_inst[i];
        // batch-interleaved advice arrays [row][col][half][instance][2 words]: instance i starts 2 words in
        d.base = (uint64_t*)d_base + (size_t)i * 2;
        d.range = (uint64_t*)d_range + (size_t)i * 2;
        d.select = (uint64_t*)d_select + (size_t)i * 2;
        d.inputs = (const uint64_t*)d_inputs + (size_t)i * r.n_input_slots * slot_words;
        d.status = (uint32_t*)d_status + i;
        d.hints = J.ws_hints + (size_t)i * wsw;
        d.nd = J.ws_nd + (size_t)i * wsw;
        d.jac = J.ws_jac + (size_t)i * wsw;
        d.sel = J.ws_sel + (size_t)i * wsw;
        d.ws = (uint32_t)(n_instances * wsw);
        d.pad_ = 0;
    }
    HIP_TRY(hipMemcpyAsync(J.d_inst, J.h_inst, (size_t)n_instances * sizeof(InstanceDescHost), hipMemcpyHostToDevice, sa));
    HIP_TRY(hipEventRecord(J.upload_ev, sa));
    // stream digest: the expansion and fix-up kernels of this run add to it (every other stream starts behind this point)
    if (d_digests) {
        if (J.dg_cap < n_instances) {
            if (J.dg_shards) {
                HIP_TRY(hipDeviceSynchronize());
                HIP_TRY(hipFree(J.dg_shards));
                J.dg_shards = nullptr;
            }
            HIP_TRY(hipMalloc((void**)&J.dg_shards, (size_t)H2E_DG_SHARDS * 3 * n_instances * 4 * sizeof(uint64_t)));
            J.dg_cap = n_instances;
        }
        HIP_TRY(hipMemsetAsync(J.dg_shards, 0, (size_t)H2E_DG_SHARDS * 3 * n_instances * 4 * sizeof(uint64_t), sa));
    }
    bool used_sd = false;
    std::vector<hipEvent_t> early_done(r.pre_kernels.size(), nullptr);
    size_t n_sync = 0;
    auto sync_event = [&]() -> hipEvent_t {
        if (n_sync == J.sync_ev.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            J.sync_ev.push_back(e);
        }
        return J.sync_ev[n_sync++];
    };
    auto prof_event = [&](uint32_t k) -> hipEvent_t {
        while (J.ev.size() <= k) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            J.ev.push_back(e);
        }
        return J.ev[k];
    };
    const bool profiling = ctx->profiling;
    J.profiled = profiling;
    {   // the engine's streams start after everything already queued on the caller's stream
        hipEvent_t e = sync_event();
        HIP_TRY(hipEventRecord(e, sa));
        HIP_TRY(hipStreamWaitEvent(sb, e, 0));
    }
    J.n_launches = 0;
    J.x_kernels.clear();
    ctx->last_split_segments = 0;
    hipStream_t se = sc;
    std::vector<hipEvent_t> seg_ev(r.segments.size(), nullptr), side_done(r.segments.size(), nullptr);
    hipEvent_t run_begin = sync_event();
    HIP_TRY(hipEventRecord(run_begin, sa));
    bool used_se = false, used_small = false;
    H2ELaunch pending_L;
    uint32_t pending_li = 0;
    bool have_pending = false;
    // a held-back expansion goes out once the next long predictor chain (a segment with pre-selected points: the MSM
    // windows) is queued - next to that chain it costs 1 ms less than next to the select kernel before it - or, if there
    // is no such segment, right behind the next value chain
    bool hold_longer = false;
    auto flush_pending = [&]() -> int {   // launch an expansion that was held back behind a later value chain
        hipStream_t sp = sb;
        if (!join && (ctx->sched & 8u)) {   // pipelined: beside the previous run's big expansions, not between them
            if (!ctx->small_stream) HIP_TRY(make_stream(ctx, &ctx->small_stream, ctx->prio_expand, 0));
            sp = ctx->small_stream;
            used_small = true;
        }
        hipEvent_t e0 = sync_event();
        HIP_TRY(hipEventRecord(e0, sa));
        HIP_TRY(hipStreamWaitEvent(sp, e0, 0));
        if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * pending_li + 2), sp));
        int prc2 = H2E_LAUNCH((int)pending_L.field_pair, 2, &pending_L, J.d_inst, n_instances, ctx->d_fc[pending_L.field_pair], sp);
        if (prc2 != 0) return fail(H2E_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString((hipError_t)prc2));
        if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * pending_li + 3), sp));
        if (pending_L.n_fixups) {
            hipEvent_t e1 = sync_event();
            HIP_TRY(hipEventRecord(e1, sp));
            HIP_TRY(hipStreamWaitEvent(sd, e1, 0));
            used_sd = true;
            prc2 = H2E_LAUNCH((int)pending_L.field_pair, 4, &pending_L, J.d_inst, n_instances, ctx->d_fc[pending_L.field_pair], sd);
            if (prc2 != 0) return fail(H2E_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString((hipError_t)prc2));
        }
        have_pending = false;
        return 0;
    };
    // (the field chain reads the constant pool where the MSM predictors read strand parameters)
    auto pk_params = [&](const h2e::PreKernel& pk) -> const uint32_t* {
        return pk.k.kind == H2E_PRE_FIELD_CHAIN ? (const uint32_t*)p->d_pool : p->d_params;
    };
    // the digit chain in front of the next launched segment, if it has one (-1: none): it becomes ready together with an expansion
    // launched now, which then gets a gate (engine.hip h2e_gate; H2E_SCHED bit 32 switches the gates off, A/B)
    auto next_digit_chain = [&](size_t si) -> int {
        // (a pipelined run's expansions queue on streams they share with other runs' - a gate there would hold those up - and its
        // chains start next to other runs' expansions whatever it does: h2e_run only)
        if ((ctx->sched & 32u) || !join) return -1;
        size_t sj = si + 1;
        while (sj < r.segments.size() && r.segments[sj].tape_end <= r.segments[sj].tape_begin) sj++;
        if (sj >= r.segments.size()) return -1;
        for (size_t pi = 0; pi < r.pre_kernels.size(); pi++) {
            const h2e::PreKernel& pk = r.pre_kernels[pi];
            if (pk.before_segment == sj && pk.k.kind == H2E_PRE_FIELD_CHAIN && pk.k.f_mode == 1 && pk.early_after_segment < 0) return (int)pi;
        }
        return -1;
    };
    int gate_for = -1;   // the pre-kernel a gate launched in this run is waiting for
    // a gate counts on a chain that is launched later in this function: on EVERY exit on which that chain was not launched (an
    // error return in between, or no such chain) the slot's target goes back, or every later gate of the slot would spin its
    // whole timeout (the gate itself just times out)
    struct GateGuard {
        JobSlot& J;
        const int& gate_for;
        uint32_t n;
        ~GateGuard() { if (gate_for >= 0) J.gate_total -= n; }
    } gate_guard{J, gate_for, n_instances};
    bool run_has_big_x = false;
    for (size_t si = 0; si < r.segments.size(); si++)
        run_has_big_x = run_has_big_x || (p->seg_n_sub[si] > 1 && (uint64_t)p->seg_n_sub[si] * r.segments[si].n_strands * n_instances >= ctx->small_x_lanes);
    for (size_t si = 0; si < r.segments.size(); si++) {
        const h2e::Segment& s = r.segments[si];
        if (s.tape_end <= s.tape_begin) continue;
#ifdef H2E_DEBUG_HOOKS
        g_dbg_run = ctx->n_runs;
        g_dbg_si = (int)si;
#endif
        if ((int64_t)si == p->tail_from && sa == sa_main) {
            // the serial tail of the program: from here on the value chain continues on the slot's side stream, and the
            // caller's stream is free for the next run's value chain
            hipEvent_t e = sync_event();
            HIP_TRY(hipEventRecord(e, sa));
            HIP_TRY(hipStreamWaitEvent(sc, e, 0));
            sa = sc;
            used_se = true;
        }
        // side segments this one reads must be done
        for (size_t sj = 0; sj < si; sj++)
            if (side_done[sj] && p->seg_first_reader[sj] <= si) {
                HIP_TRY(hipStreamWaitEvent(sa, side_done[sj], 0));
                side_done[sj] = nullptr;
            }
        uint32_t li = J.n_launches;
        if (J.x_kernels.size() <= li) J.x_kernels.resize(li + 1, 1);
        if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 0), sa));
        // this segment's predictors: chains first, then (after the early starters below) their finalize kernels
        for (size_t pi = 0; pi < r.pre_kernels.size(); pi++) {
            const h2e::PreKernel& pk = r.pre_kernels[pi];
            if (pk.before_segment != si) continue;
            if (early_done[pi]) {  // already running on the side stream: just wait for it
                HIP_TRY(hipStreamWaitEvent(sa, early_done[pi], 0));
                continue;
            }
            H2EPreKernel k1 = pk.k;
            if ((int)pi == gate_for) {   // a gate is waiting for this chain's workgroups
                k1.f_started = J.d_gate;
                gate_for = -1;
            }
            int prc = H2E_PREDICT(fp, 1, &k1, p->d_pre_args, pk_params(pk), p->d_aux, J.d_inst, n_instances, ctx->d_fc[fp], sa);
            if (prc != 0) return fail(H2E_ERR_HIP, std::string("predictor launch failed: ") + hipGetErrorString((hipError_t)prc));
        }
        if (have_pending && hold_longer && s.sel_stride) {
            int frc = flush_pending();
            if (frc) return frc;
        }
        // predictors of later segments that only depend on this segment's predictor chains start now, on the side stream
        for (size_t pi = 0; pi < r.pre_kernels.size(); pi++) {
            const h2e::PreKernel& pk = r.pre_kernels[pi];
            if (pk.early_after_segment != (int32_t)si || pk.before_segment <= si) continue;
            hipEvent_t e0 = sync_event();
            HIP_TRY(hipEventRecord(e0, sa));
            HIP_TRY(hipStreamWaitEvent(sc, e0, 0));
            int prc = H2E_PREDICT(fp, 3, &pk.k, p->d_pre_args, pk_params(pk), p->d_aux, J.d_inst, n_instances, ctx->d_fc[fp], sc);
            if (prc != 0) return fail(H2E_ERR_HIP, std::string("predictor launch failed: ") + hipGetErrorString((hipError_t)prc));
            hipEvent_t e1 = sync_event();
            HIP_TRY(hipEventRecord(e1, sc));
            early_done[pi] = e1;
            used_se = true;
        }
        for (size_t pi = 0; pi < r.pre_kernels.size(); pi++) {
            const h2e::PreKernel& pk = r.pre_kernels[pi];
            if (pk.before_segment != si || early_done[pi]) continue;
            int prc = H2E_PREDICT(fp, 2, &pk.k, p->d_pre_args, pk_params(pk), p->d_aux, J.d_inst, n_instances, ctx->d_fc[fp], sa);
            if (prc != 0) return fail(H2E_ERR_HIP, std::string("predictor launch failed: ") + hipGetErrorString((hipError_t)prc));
        }
        H2ELaunch L;
        L.tape = p->d_tape + s.tape_begin;
        L.n_ops = s.tape_end - s.tape_begin;
        L.n_strands = s.n_strands;
        L.strand_base0 = s.base0;
        L.strand_range0 = s.range0;
        L.strand_select0 = s.select0;
        L.delta_base = s.dbase;
        L.delta_range = s.drange;
        L.delta_select = s.dselect;
        L.input_stride = s.input_stride;
        L.n_params = s.n_params;
        L.params = p->d_params + s.params_begin;
        L.aux = p->d_aux;
        L.const_pool = p->d_pool;
        L.hint_stride = s.hint_stride;
        L.n_fixups = s.n_fixups;
        L.fixups = p->d_fixups + s.fixups_begin;
        L.rel_refs = s.is_fork ? 1 : 0;
        L.field_pair = (uint32_t)s.field_pair;
        L.slot_words = (uint32_t)slot_words;
        L.n_sub = p->seg_n_sub[si];
        L.sub = L.n_sub > 1 ? p->d_subs + p->seg_sub_begin[si] : nullptr;
        L.pk_order = nullptr;
        L.pk_n_sub = 0;
        if (L.n_sub > 1 && si < p->seg_pk_waves.size() && p->d_pk_order) {
            L.pk_order = p->d_pk_order;
            L.pk_n_sub = L.n_sub;
            for (int k = 0; k < 5; k++) {
                L.pk_off[k] = p->seg_pk_off[si][k];
                L.pk_waves[k] = p->seg_pk_waves[si][k];
            }
        }
        bool compiled = si < p->seg_n_pieces.size() && p->seg_n_pieces[si] > 0;
        L.vtape = compiled ? p->d_vtape : nullptr;
        L.vpieces = compiled ? p->d_vpieces + 2 * (size_t)p->seg_piece_begin[si] : nullptr;
        L.n_vpieces = compiled ? p->seg_n_pieces[si] : 0;
        L.v_int_slots = compiled ? p->seg_v_slots[si] : 0;
        L.v_units = compiled ? p->seg_v_units[si] : 0;
        L.sel_stride = s.sel_stride;
        bool levels = compiled && si < p->seg_l_steps.size() && p->seg_l_steps[si] > 0;
        L.lrecs = levels ? p->d_lrecs + p->seg_l_begin[si] : nullptr;
        L.lrefs = p->d_lrefs;
        L.lrounds = levels && p->seg_l_pair[si] == 2 ? p->d_lrounds + p->seg_lr_begin[si] : nullptr;
        L.l_recs = levels ? p->seg_l_recs[si] : 0;
        bool hstore = si < p->seg_n_sops.size() && p->seg_n_sops[si] > 0;
        L.s_words = hstore ? p->d_swords + p->seg_s_begin[si] : nullptr;
        L.s_offsets = hstore ? p->d_soffsets + p->seg_so_begin[si] : nullptr;
        L.s_ktab = hstore ? p->d_sktab + p->seg_sk_begin[si] : nullptr;
        L.n_sops = hstore ? p->seg_n_sops[si] : 0;
        L.s_ext = hstore && si < p->seg_sx_begin.size() ? p->d_sext + p->seg_sx_begin[si] : nullptr;
        L.dg_out = d_digests ? J.dg_shards : nullptr;
        L.dg_shards = H2E_DG_SHARDS;
        L.l_steps = levels ? p->seg_l_steps[si] : 0;
        L.l_slots = levels ? p->seg_l_slots[si] : 0;
        L.l_pair = levels ? p->seg_l_pair[si] : 0;
        int lrc;
        auto launch_one = [&](int mode, const H2ELaunch& l, hipStream_t st) -> int {
            int rc2 = H2E_LAUNCH((int)l.field_pair, mode, &l, J.d_inst, n_instances, ctx->d_fc[l.field_pair], st);
            if (rc2 != 0) return fail(H2E_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString((hipError_t)rc2));
            return 0;
        };
        auto launch = [&](int mode, hipStream_t st) -> int { return launch_one(mode, L, st); };
        // A big expansion (the MSM windows) goes out as two back-to-back launches over the first x_split_pct percent / the
        // rest of its sub-ranges: while the first one drains, the value chain that became ready meanwhile (the MSM tail's
        // replay wants most of a CU's LDS per workgroup) gets its CUs instead of waiting for the whole expansion, and the
        // inverse fix-up of the first part runs under the second.
        // (x_parts > 2: the second part again in equal launches - the last inverse fix-up, which nothing can run under, shrinks with it)
        uint32_t split_sub = 0, split_fix = 0;
        std::vector<uint32_t> part_sub, part_fix;   // boundaries: sub-range index / fix-up index each launch starts at (+ the end)
        {
            uint32_t pct = ctx->x_split_pct;
            if (pct > 0 && pct < 100 && L.n_sub >= 4 && (uint64_t)L.n_sub * L.n_strands * n_instances >= ctx->x_split_min_lanes) {
                split_sub = std::min<uint32_t>(std::max<uint32_t>(2, (uint32_t)((uint64_t)L.n_sub * pct / 100)), L.n_sub - 2);
                auto fb = r.fixups.begin() + s.fixups_begin;
                const bool sorted_fix = std::is_sorted(fb, fb + s.n_fixups);
                // fix-up rows are recorded in tape order: those below the first row of a part belong to the parts before it
                auto fix_at = [&](uint32_t sub) -> uint32_t {
                    uint32_t row = r.tape[s.tape_begin + p->h_subs[p->seg_sub_begin[si] + sub]].base_row;
                    return sorted_fix ? (uint32_t)(std::lower_bound(fb, fb + s.n_fixups, row) - fb) : 0;
                };
                split_fix = fix_at(split_sub);
                part_sub = {0, split_sub};
                part_fix = {0, split_fix};
                uint32_t extra = std::min<uint32_t>(ctx->x_parts > 2 ? ctx->x_parts - 2 : 0, (L.n_sub - split_sub) / 2);
                for (uint32_t q = 1; q <= extra; q++) {
                    uint32_t at = split_sub + (uint32_t)((uint64_t)(L.n_sub - split_sub) * q / (extra + 1));
                    if (at > part_sub.back() && at < L.n_sub) {
                        part_sub.push_back(at);
                        part_fix.push_back(fix_at(at));
                    }
                }
                part_sub.push_back(L.n_sub);
                part_fix.push_back(s.n_fixups);
                ctx->last_split_segments++;
            }
        }
        // the inverse fix-up of a segment only touches cells nothing else reads or writes: own stream, after the expansion
        // (a small expansion keeps its fix-up in its own stream)
        bool fixup_in_stream = false;
        auto fixup_part = [&](hipStream_t st, uint32_t lo, uint32_t hi) -> int {
            if (hi <= lo) return 0;
            H2ELaunch f = L;
            f.fixups = L.fixups + lo;
            f.n_fixups = hi - lo;
            if (fixup_in_stream && !join && sc && (ctx->sched & 1u) && st != sc) {   // pipelined: keep the shared expansion stream free
                hipEvent_t e = sync_event();
                HIP_TRY(hipEventRecord(e, st));
                HIP_TRY(hipStreamWaitEvent(sc, e, 0));
                used_se = true;
                return launch_one(4, f, sc);
            }
            // A pipelined run WITHOUT a big expansion (a pairing batch smaller than half a wave) sends its small fix-ups to the fix-up
            // stream as well: its expansions share one stream with those of the other runs in flight, and a 0.1 ms one-workgroup
            // inversion behind every one of them made that stream the step (16 x bls12_381 at four runs in flight 1.81 -> 1.58 ms, 8 x
            // bn256 1.40 -> 1.10, 2 x bls12_381 1.22 -> 0.95; the MSM, whose small expansions run beside big ones: no difference, left
            // as it was).  H2E_SCHED & 64: in their stream as before (A/B)
            if (fixup_in_stream && (join || (run_has_big_x && !(ctx->sched & 128u)) || (ctx->sched & 64u))) return launch_one(4, f, st);   // (128: to the fix-up stream whatever the run holds - experiment)
            hipEvent_t e = sync_event();
            HIP_TRY(hipEventRecord(e, st));
            HIP_TRY(hipStreamWaitEvent(sd, e, 0));
            used_sd = true;
            return launch_one(4, f, sd);
        };
        auto fixup_after = [&](hipStream_t st) -> int { return fixup_part(st, 0, s.n_fixups); };
        // full expansion + fix-up of this segment on stream st
        auto expand = [&](hipStream_t st) -> int {
            int xrc;
            if (!split_sub) {
                if ((xrc = launch(2, st))) return xrc;
                return 0;
            }
            const size_t n_parts = part_sub.size() - 1;
            J.x_kernels[li] = (uint32_t)n_parts;
            for (size_t q = 0; q < n_parts; q++) {
                H2ELaunch a = L;
                a.n_sub = part_sub[q + 1] - part_sub[q];
                a.sub = L.sub + part_sub[q];
                if ((xrc = launch_one(2, a, st))) return xrc;
                if (q + 1 < n_parts && (xrc = fixup_part(st, part_fix[q], part_fix[q + 1]))) return xrc;   // (the last part's: expand_fixup)
            }
            return 0;
        };
        auto expand_fixup = [&](hipStream_t st) -> int { return fixup_part(st, split_sub ? part_fix[part_fix.size() - 2] : 0, s.n_fixups); };
        if (L.n_sub > 1) {
            if ((lrc = launch(1, sa))) return lrc;
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 1), sa));
            if (have_pending && !hold_longer && (lrc = flush_pending())) return lrc;
            // (H2E_SCHED & 16: pipelined runs do not hold the expansion back - measured 0.35 ms per step worse)
            if (s.expand_after_next && si + 1 < r.segments.size() && p->seg_n_sub[si + 1] > 1 && (join || !(ctx->sched & 16u))) {
                if (have_pending && (lrc = flush_pending())) return lrc;   // single slot: never overwrite a held expansion
                pending_L = L;
                pending_li = li;
                have_pending = true;
                hold_longer = false;
                for (size_t sj = si + 1; sj < r.segments.size(); sj++) hold_longer = hold_longer || r.segments[sj].sel_stride != 0;
                J.n_launches++;
                seg_ev[si] = sync_event();
                HIP_TRY(hipEventRecord(seg_ev[si], sa));
                continue;
            }
            // a small expansion (the MSM tail: 763 waves) queues behind the big one on the expansion stream: beside it on
            // the side stream it and its fix-up slow the big one down by more than they take alone; its fix-up follows it
            // in its stream: the fix-up stream still holds the big expansion's second fix-up
            bool small_x = (uint64_t)L.n_sub * L.n_strands * n_instances < ctx->small_x_lanes;
            hipStream_t sx = (small_x && !join && sc && (ctx->sched & 2u)) ? sc : sb;
            if (sx == sc) used_se = true;
            if (small_x && !join && (ctx->sched & 4u)) {
                // pipelined: the shared expansion stream only carries the big expansions - the small ones (latency-bound: a few
                // hundred waves and their inverse fix-ups) run beside them on their own stream instead of between them
                if (!ctx->small_stream) HIP_TRY(make_stream(ctx, &ctx->small_stream, ctx->prio_expand, 0));
                sx = ctx->small_stream;
                used_small = true;
            }
            fixup_in_stream = small_x;
            hipEvent_t e = sync_event();
            HIP_TRY(hipEventRecord(e, sa));
            HIP_TRY(hipStreamWaitEvent(sx, e, 0));
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 2), sx));
            // test hook (H2E_OPT_TEST_SKIP_EXPANSION): leave out the expansion of cut segment <si> (or, with -1, of every cut
            // segment but the last): whatever the value chain reads must have been stored by the value chain itself.  The
            // run's status words get H2E_ST_TEST_HOOK, so its arrays cannot be mistaken for a witness.
            bool skip_x = false;
            if (ctx->test_skip_expansion != INT64_MIN) {
                bool later_cut = false;
                for (size_t sj = si + 1; sj < r.segments.size(); sj++) later_cut = later_cut || p->seg_n_sub[sj] > 1;
                skip_x = ctx->test_skip_expansion == (int64_t)si || (ctx->test_skip_expansion == -1 && later_cut);
            }
            if (!skip_x && gate_for < 0) {
                int gpi = next_digit_chain(si);
                if (gpi >= 0) {
                    if (!J.d_gate) {
                        HIP_TRY(hipMalloc((void**)&J.d_gate, 4));
                        HIP_TRY(hipMemset(J.d_gate, 0, 4));
                        J.gate_total = 0;
                    }
                    J.gate_total += n_instances;   // one workgroup per instance
                    gate_for = gpi;
                    int grc = h2e_engine_gate(J.d_gate, J.gate_total, sx);
                    if (grc != 0) return fail(H2E_ERR_HIP, std::string("gate launch failed: ") + hipGetErrorString((hipError_t)grc));
                }
            }
            if (!skip_x && (lrc = expand(sx))) return lrc;
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 3), sx));
            if (!skip_x && (lrc = expand_fixup(sx))) return lrc;
        } else if (p->seg_side_dep[si] != -2) {
            // runs beside the value chain: after the last segment it reads, before the first segment that reads it
            int32_t depi = p->seg_side_dep[si];
            HIP_TRY(hipStreamWaitEvent(se, depi >= 0 && seg_ev[depi] ? seg_ev[depi] : run_begin, 0));
            used_se = true;
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 1), sa));
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 2), se));
            if ((lrc = launch(2, se))) return lrc;
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 3), se));
            if ((lrc = fixup_after(se))) return lrc;
            side_done[si] = sync_event();
            HIP_TRY(hipEventRecord(side_done[si], se));
        } else if (p->seg_deferrable[si]) {
            // nothing later reads this segment's cells: off the critical stream
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 1), sa));
            hipEvent_t e = sync_event();
            HIP_TRY(hipEventRecord(e, sa));
            HIP_TRY(hipStreamWaitEvent(sb, e, 0));
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 2), sb));
            if ((lrc = launch(2, sb))) return lrc;
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 3), sb));
            if ((lrc = fixup_after(sb))) return lrc;
        } else {
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 1), sa));
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 2), sa));
            if ((lrc = launch(2, sa))) return lrc;
            if (profiling) HIP_TRY(hipEventRecord(prof_event(4 * li + 3), sa));
            if ((lrc = fixup_after(sa))) return lrc;
        }
        seg_ev[si] = sync_event();
        HIP_TRY(hipEventRecord(seg_ev[si], sa));
        J.n_launches++;
    }
    if (have_pending) {   // (no later segment took it with it)
        int frc = flush_pending();
        if (frc) return frc;
    }
    if (ctx->test_skip_expansion != INT64_MIN) {
        int orc = h2e_engine_or_status(J.d_inst, n_instances, H2E_ST_TEST_HOOK, sa);
        if (orc != 0) return fail(H2E_ERR_HIP, std::string("status kernel launch failed: ") + hipGetErrorString((hipError_t)orc));
    }
    // completion: the fix-up stream collects the other streams and records the slot's `done` event; h2e_run then makes
    // the caller's stream wait for it, h2e_submit leaves that to h2e_wait
    {
        hipEvent_t ea = sync_event();
        HIP_TRY(hipEventRecord(ea, sa_main));
        HIP_TRY(hipStreamWaitEvent(sd, ea, 0));
        hipEvent_t eb = sync_event();
        HIP_TRY(hipEventRecord(eb, sb));
        HIP_TRY(hipStreamWaitEvent(sd, eb, 0));
        if (used_se) {
            hipEvent_t e3 = sync_event();
            HIP_TRY(hipEventRecord(e3, se));
            HIP_TRY(hipStreamWaitEvent(sd, e3, 0));
        }
        if (used_small) {
            hipEvent_t e4 = sync_event();
            HIP_TRY(hipEventRecord(e4, ctx->small_stream));
            HIP_TRY(hipStreamWaitEvent(sd, e4, 0));
        }
        (void)used_sd;
        if (d_digests) {   // every kernel that adds to the digest shards has finished here
            int drc = h2e_engine_digest_reduce(J.dg_shards, H2E_DG_SHARDS, 3 * n_instances * 4, d_digests, sd);
            if (drc != 0) return fail(H2E_ERR_HIP, std::string("digest kernel launch failed: ") + hipGetErrorString((hipError_t)drc));
        }
        HIP_TRY(hipEventRecord(J.done, sd));
        if (join) HIP_TRY(hipStreamWaitEvent(sa_main, J.done, 0));
    }
    return 0;
}

int h2e_run(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
            void* d_select, void* d_status, void* stream_) {
    if (n_instances == 0) return 0;
    return run_impl(ctx, p, n_instances, d_inputs, d_base, d_range, d_select, d_status, (hipStream_t)stream_, true, nullptr);
}

int h2e_submit(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
               void* d_select, void* d_status, void* stream_, int* job) {
    if (!job) return fail(H2E_ERR_INVALID, "job is null");
    *job = -1;
    if (n_instances == 0) return fail(H2E_ERR_INVALID, "n_instances must be > 0");
    return run_impl(ctx, p, n_instances, d_inputs, d_base, d_range, d_select, d_status, (hipStream_t)stream_, false, job);
}

int h2e_run_digest(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
                   void* d_select, void* d_status, void* d_digests, void* stream_) {
    if (n_instances == 0) return 0;
    if (!d_digests) return fail(H2E_ERR_INVALID, "d_digests is null");
    return run_impl(ctx, p, n_instances, d_inputs, d_base, d_range, d_select, d_status, (hipStream_t)stream_, true, nullptr, d_digests);
}
int h2e_submit_digest(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
                      void* d_select, void* d_status, void* d_digests, void* stream_, int* job) {
    if (!job) return fail(H2E_ERR_INVALID, "job is null");
    *job = -1;
    if (n_instances == 0) return fail(H2E_ERR_INVALID, "n_instances must be > 0");
    if (!d_digests) return fail(H2E_ERR_INVALID, "d_digests is null");
    return run_impl(ctx, p, n_instances, d_inputs, d_base, d_range, d_select, d_status, (hipStream_t)stream_, false, job, d_digests);
}

int h2e_wait(h2e_ctx* ctx, int job, void* stream_) {
    if (!ctx) return fail(H2E_ERR_INVALID, "null ctx");
    std::lock_guard<std::mutex> guard(ctx->mu);   // (the slot's event is created under this lock by run_impl)
    if (job < 0 || job >= (int)ctx->depth || !ctx->slots[job].done) return fail(H2E_ERR_INVALID, "bad job");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream_, ctx->slots[job].done, 0));
    return 0;
}

int h2e_ctx_set_option(h2e_ctx* ctx, int option, int64_t value) {
    if (!ctx) return fail(H2E_ERR_INVALID, "null ctx");
    std::lock_guard<std::mutex> guard(ctx->mu);
    switch (option) {
        case H2E_OPT_X_SPLIT_PCT: ctx->x_split_pct = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(100, value)); return 0;
        case H2E_OPT_X_SPLIT_MIN_LANES: ctx->x_split_min_lanes = (uint64_t)std::max<int64_t>(0, value); return 0;
        case H2E_OPT_TEST_SKIP_EXPANSION: ctx->test_skip_expansion = value; return 0;
        case H2E_OPT_PIPELINE_DEPTH:
            if (value < 1 || value > h2e_ctx::N_SLOTS) return fail(H2E_ERR_INVALID, "pipeline depth out of range");
            HIP_TRY(hipSetDevice(ctx->device));
            HIP_TRY(hipDeviceSynchronize());   // no run may be in flight while the slots are renumbered
            ctx->depth = (uint32_t)value;
            ctx->n_runs = 0;
            return 0;
        case H2E_OPT_TEST_SCAN_FALLBACK:
            HIP_TRY(hipSetDevice(ctx->device));
            HIP_TRY(hipDeviceSynchronize());
            h2e_engine_set_tuning(3, (int)value);
            return 0;
        case H2E_OPT_PREFAULT_HBM: {
            // The first process that touches the HBM of a freshly booted device pays for it: kernels that stream into memory
            // nobody has written since boot run at half their rate (a 2^16-point MSM step 47 instead of 24 ms; any later
            // process - or this one, after the first pass over the memory - is unaffected).  One throw-away allocate / fill /
            // free of `value` percent of the free memory (0.3 s for 270 GB) takes that out of the caller's first runs.
            if (value <= 0) return 0;
            HIP_TRY(hipSetDevice(ctx->device));
            size_t free_b = 0, total_b = 0;
            HIP_TRY(hipMemGetInfo(&free_b, &total_b));
            size_t want = (size_t)((double)free_b * (double)std::min<int64_t>(value, 98) / 100.0) & ~(size_t)0xfffff;
            if (want == 0) return 0;
            void* scratch = nullptr;
            HIP_TRY(hipMalloc(&scratch, want));
            hipError_t e = hipMemset(scratch, 0xff, want);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            (void)hipFree(scratch);
            if (e != hipSuccess) return fail(H2E_ERR_HIP, std::string("prefault: ") + hipGetErrorString(e));
            return 0;
        }
        case H2E_OPT_OP_CACHE_CAP: {
            if (value < 1) return fail(H2E_ERR_INVALID, "op cache capacity must be at least 1");
            std::lock_guard<std::mutex> g2(ctx->op_mu);
            ctx->op_cache_cap = (size_t)value;
            ctx->op_cache_trim();
            return 0;
        }
        default: return fail(H2E_ERR_INVALID, "unknown option");
    }
}
int64_t h2e_ctx_get_stat(h2e_ctx* ctx, int stat) {
    if (!ctx) return -1;
    std::lock_guard<std::mutex> guard(ctx->mu);
    switch (stat) {
        case H2E_STAT_LAST_SPLIT_SEGMENTS: return ctx->last_split_segments;
        case H2E_STAT_RUNS: return (int64_t)ctx->n_runs;
        case H2E_STAT_SCAN_FALLBACKS:
            if (hipSetDevice(ctx->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
            return (int64_t)h2e_engine_scan_fallbacks();
        case H2E_STAT_PIPELINE_DEPTH: return ctx->depth;
        case H2E_STAT_MAX_PIPELINE_DEPTH: return h2e_ctx::N_SLOTS;
        case H2E_STAT_OP_CACHE_HITS: {
            std::lock_guard<std::mutex> g2(ctx->op_mu);
            return (int64_t)ctx->op_hits;
        }
        case H2E_STAT_OP_CACHE_MISSES: {
            std::lock_guard<std::mutex> g2(ctx->op_mu);
            return (int64_t)ctx->op_misses;
        }
        case H2E_STAT_OP_CACHE_EVICTIONS: {
            std::lock_guard<std::mutex> g2(ctx->op_mu);
            return (int64_t)ctx->op_evictions;
        }
        case H2E_STAT_OP_CACHE_SIZE: {
            std::lock_guard<std::mutex> g2(ctx->op_mu);
            return (int64_t)ctx->op_cache.size();
        }
        default: return -1;
    }
}

int h2e_program_outputs(const h2e_program* p, uint32_t* refs, uint32_t cap) {
    if (!p) return fail(H2E_ERR_INVALID, "null program");
    const h2e::Recorder& r = *p->rec;
    for (uint32_t i = 0; i < r.outputs.size() && i < cap; i++) refs[i] = r.outputs[i];
    return (int)r.outputs.size();
}
int h2e_program_launches(const h2e_program* p, uint64_t* out, uint32_t cap) {
    if (!p) return fail(H2E_ERR_INVALID, "null program");
    const h2e::Recorder& r = *p->rec;
    uint32_t k = 0;
    for (auto& s : r.segments) {
        if (s.tape_end <= s.tape_begin) continue;
        if (k < cap) {
            uint64_t* o = out + (size_t)k * 8;
            o[0] = s.n_strands;
            o[1] = s.tape_end - s.tape_begin;
            o[2] = s.cells;
            o[3] = s.dbase;
            o[4] = s.drange;
            o[5] = s.dselect;
            o[6] = s.n_params;
            o[7] = s.is_fork ? s.base0 : r.tape[s.tape_begin].base_row;   // (main context: the base row its first op starts at)
        }
        k++;
    }
    return (int)k;
}

// First row of each advice array the k-th launch writes (a fork: of its strand 0): out[0..2] = base, range, select row.
int h2e_program_launch_rows(const h2e_program* p, uint32_t launch, uint64_t* out) {
    if (!p || !out) return fail(H2E_ERR_INVALID, "null argument");
    const h2e::Recorder& r = *p->rec;
    uint32_t k = 0;
    for (auto& s : r.segments) {
        if (s.tape_end <= s.tape_begin) continue;
        if (k++ != launch) continue;
        const auto& op = r.tape[s.tape_begin];
        out[0] = s.is_fork ? s.base0 : op.base_row;
        out[1] = s.is_fork ? s.range0 : op.range_row;
        out[2] = s.is_fork ? s.select0 : op.select_row;
        return 0;
    }
    return fail(H2E_ERR_INVALID, "no such launch");
}

// Diagnostics: the opcodes of one launch's tape (the k-th segment h2e_program_launches lists) and the op indices its expansion's
// sub-ranges start at - what exp/pack_sim.py replays on the host to price a packing of sub-ranges into waves.  Returns the
// number of ops; *n_subs = sub-range bounds written (first = 0, last = the number of ops).
int h2e_program_tape_opcodes(const h2e_program* p, uint32_t launch, uint16_t* opcodes, uint32_t cap, uint32_t* subs, uint32_t subs_cap,
                             uint32_t* n_subs) {
    if (!p) return fail(H2E_ERR_INVALID, "null program");
    const h2e::Recorder& r = *p->rec;
    uint32_t k = 0;
    for (auto& s : r.segments) {
        if (s.tape_end <= s.tape_begin) continue;
        if (k++ != launch) continue;
        uint32_t n_ops = s.tape_end - s.tape_begin;
        for (uint32_t i = 0; i < n_ops && i < cap; i++) opcodes[i] = r.tape[s.tape_begin + i].opcode;
        std::vector<uint32_t> b{0};
        for (uint32_t c = 0; c < s.n_cuts; c++) {
            uint32_t at = r.cuts[s.cuts_begin + c];
            if (at > b.back() && at < n_ops) b.push_back(at);
        }
        b.push_back(n_ops);
        if (n_subs) *n_subs = (uint32_t)b.size();
        for (size_t i = 0; i < b.size() && i < subs_cap; i++) subs[i] = b[i];
        return (int)n_ops;
    }
    return fail(H2E_ERR_INVALID, "no such launch");
}

// Diagnostics: the packed expansion's order table of the k-th launch for 2 << groups_log2m1 groups per wave (the table
// ensure_device_program uploads): returns its entries (waves x groups; ~0u = empty slot), copies at most `cap` of them
int h2e_program_pack_order(const h2e_program* p, uint32_t launch, uint32_t groups_log2m1, uint32_t* out, uint32_t cap) {
    if (!p || groups_log2m1 > 4) return fail(H2E_ERR_INVALID, "bad argument");
    const h2e::Recorder& r = *p->rec;
    uint32_t k = 0;
    for (auto& s : r.segments) {
        if (s.tape_end <= s.tape_begin) continue;
        if (k++ != launch) continue;
        uint32_t n_ops = s.tape_end - s.tape_begin;
        std::vector<uint32_t> b{0};
        for (uint32_t c = 0; c < s.n_cuts; c++) {
            uint32_t at = r.cuts[s.cuts_begin + c];
            if (at > b.back() && at < n_ops) b.push_back(at);
        }
        b.push_back(n_ops);
        if (b.size() < 3) return 0;
        std::vector<uint32_t> tab;
        std::array<uint32_t, 5> off{}, waves{};
        pack_orders_of(r, s, b.data(), (uint32_t)b.size() - 1, tab, off, waves);
        uint32_t n = waves[groups_log2m1] * (2u << groups_log2m1);
        for (uint32_t i = 0; i < n && i < cap; i++) out[i] = tab[off[groups_log2m1] + i];
        return (int)n;
    }
    return fail(H2E_ERR_INVALID, "no such launch");
}

// Diagnostics: how the value chain of the k-th launch puts its escaping values in place: out[0] = store ops per strand of a hint
// store (0: none), out[1] = pieces of a compiled replay (0: none), out[2] = 1 if the segment has a field chain
int h2e_program_value_chain_kind(const h2e_program* p, uint32_t launch, uint32_t* out3) {
    if (!p || !out3) return fail(H2E_ERR_INVALID, "null argument");
    const h2e::Recorder& r = *p->rec;
    uint32_t k = 0;
    for (size_t si = 0; si < r.segments.size(); si++) {
        const h2e::Segment& s = r.segments[si];
        if (s.tape_end <= s.tape_begin) continue;
        if (k++ != launch) continue;
        out3[0] = si < p->seg_n_sops.size() ? p->seg_n_sops[si] : 0;
        out3[1] = si < p->seg_n_pieces.size() ? p->seg_n_pieces[si] : 0;
        out3[2] = s.field_hints ? 1 : 0;
        return 0;
    }
    return fail(H2E_ERR_INVALID, "no such launch");
}

// the program's assigned / permute bytes of one region on the device (nullptr for programs recorded without their shape)
static int device_flags(h2e_program* p, int region, const uint8_t** out) {
    const h2e::Recorder& r = *p->rec;
    *out = nullptr;
    if (!r.emit_shape) return 0;
    const uint32_t cols = region == 0 ? 5 : region == 1 ? 3 : 2;
    const uint64_t rows = region == 0 ? p->base_rows : region == 1 ? p->range_rows : p->select_rows;
    const std::vector<uint8_t>& hf = region == 0 ? r.base_flags : region == 1 ? r.range_flags : r.select_flags;
    if (hf.size() < rows * cols) return fail(H2E_ERR_SHAPE, "internal: flag array shorter than the advice array");
    if (!p->d_flags[region]) {
        HIP_TRY(hipMalloc((void**)&p->d_flags[region], std::max<size_t>(16, rows * cols)));
        HIP_TRY(hipMemcpy(p->d_flags[region], hf.data(), rows * cols, hipMemcpyHostToDevice));
    }
    *out = p->d_flags[region];
    return 0;
}

int h2e_digest(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, int region, const void* d_batch, void* d_digests, void* stream) {
    if (!ctx || !p || !d_batch || !d_digests) return fail(H2E_ERR_INVALID, "null argument");
    if (region < 0 || region > 2) return fail(H2E_ERR_INVALID, "region must be 0 (base), 1 (range) or 2 (select)");
    if (n_instances == 0) return 0;
    std::lock_guard<std::mutex> guard(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    if (p->device >= 0 && p->device != ctx->device) return fail(H2E_ERR_INVALID, "program bound to another device");
    const uint8_t* d_flags = nullptr;
    int rc = device_flags(p, region, &d_flags);
    if (rc) return rc;
    const uint32_t cols = region == 0 ? 5 : region == 1 ? 3 : 2;
    const uint64_t rows = region == 0 ? p->base_rows : region == 1 ? p->range_rows : p->select_rows;
    rc = h2e_engine_digest(cols, d_batch, d_flags, rows, n_instances, d_digests, (hipStream_t)stream);
    if (rc != 0) return fail(H2E_ERR_HIP, rc < 0 ? "digest: bad geometry" : std::string("digest launch failed: ") + hipGetErrorString((hipError_t)rc));
    return 0;
}

// Words of a unit record: status, Offset (3), the result point's coordinate limbs (2 x limbs cells, 2 words each: limbs are
// < 2^128) and z (1 word), 32-byte digest per advice array - 29 for the 3-limb curves, 33 for bls12_381 tiles.
static uint32_t unit_record_limbs(const h2e_program* p) {
    const h2e::Recorder& r = *p->rec;
    if (r.outputs.size() >= 9 && (r.outputs.size() - 3) % 2 == 0) return (uint32_t)((r.outputs.size() - 3) / 2);
    return 3;   // (a workload without a result point: the bn256 record size)
}
int h2e_unit_record_words(const h2e_program* p) {
    if (!p) return fail(H2E_ERR_INVALID, "null program");
    return (int)(1 + 3 + 4 * unit_record_limbs(p) + 1 + 12);
}
int h2e_unit_records(h2e_ctx* ctx, const h2e_program* p, uint32_t n_instances, const void* d_base, const void* d_status,
                     const void* d_digests, void* d_out, uint32_t out_stride_words, void* stream) {
    if (!ctx || !p || !d_base || !d_status || !d_out) return fail(H2E_ERR_INVALID, "null argument");
    const h2e::Recorder& r = *p->rec;
    const uint32_t limbs = unit_record_limbs(p);
    const uint32_t R = 1 + 3 + 4 * limbs + 1 + 12;
    if (out_stride_words < R) return fail(H2E_ERR_INVALID, "h2e_unit_records: out_stride_words is smaller than h2e_unit_record_words()");
    if (n_instances == 0) return 0;
    const bool has_point = r.outputs.size() == 2 * (size_t)limbs + 3;
    if (!r.outputs.empty() && !has_point) return fail(H2E_ERR_INVALID, "h2e_unit_records: the program's outputs are not a point");
    uint32_t refs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (has_point) {   // x limbs, x native, y limbs, y native, z  (h2e_program_outputs) -> x limbs, y limbs, z
        for (uint32_t i = 0; i < limbs; i++) {
            refs[i] = r.outputs[i];
            refs[limbs + i] = r.outputs[limbs + 1 + i];
        }
        refs[2 * limbs] = r.outputs[2 * limbs + 2];
        for (uint32_t i = 0; i <= 2 * limbs; i++)
            if ((refs[i] >> 30) != 0) return fail(H2E_ERR_INVALID, "h2e_unit_records: a result cell outside the base array");
    }
    const uint64_t offs[3] = {r.base_offset, r.range_offset, r.select_offset};
    // (no context lock: nothing of the context is touched - the kernel reads the caller's arrays on the caller's stream)
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = h2e_engine_unit_records(d_base, d_status, d_digests, d_out, offs, refs, limbs, has_point ? 1 : 0, n_instances, out_stride_words,
                                     (hipStream_t)stream);
    if (rc != 0) return fail(H2E_ERR_HIP, std::string("unit-records launch failed: ") + hipGetErrorString((hipError_t)rc));
    return 0;
}

int h2e_export(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, int region, int layout, int form, const void* d_batch,
               void* d_out, void* stream) {
    if (!ctx || !p || !d_batch || !d_out) return fail(H2E_ERR_INVALID, "null argument");
    if (region < 0 || region > 2) return fail(H2E_ERR_INVALID, "region must be 0 (base), 1 (range) or 2 (select)");
    if (layout != H2E_LAYOUT_ROWS && layout != H2E_LAYOUT_COLUMNS) return fail(H2E_ERR_INVALID, "bad layout");
    if (form != H2E_FORM_CANONICAL && form != H2E_FORM_MONTGOMERY) return fail(H2E_ERR_INVALID, "bad number form");
    if (n_instances == 0) return 0;
    std::lock_guard<std::mutex> guard(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    if (p->device >= 0 && p->device != ctx->device) return fail(H2E_ERR_INVALID, "program bound to another device");
    const uint32_t cols = region == 0 ? 5 : region == 1 ? 3 : 2;
    const uint64_t rows = region == 0 ? p->base_rows : region == 1 ? p->range_rows : p->select_rows;
    const uint8_t* d_flags = nullptr;
    {
        int frc = device_flags(p, region, &d_flags);
        if (frc) return frc;
    }
    int fp = p->field_pair;
    if (!ctx->d_fc[fp]) {
        HIP_TRY(hipMalloc((void**)&ctx->d_fc[fp], sizeof(H2EFieldConsts)));
        HIP_TRY(hipMemcpy(ctx->d_fc[fp], &field_pair(fp).fc, sizeof(H2EFieldConsts), hipMemcpyHostToDevice));
        HIP_TRY((hipError_t)h2e_engine_set_consts(fp, &field_pair(fp).fc));
    }
    int rc = h2e_engine_export(cols, layout == H2E_LAYOUT_COLUMNS, form == H2E_FORM_MONTGOMERY, d_batch, d_out, d_flags, rows,
                               n_instances, ctx->d_fc[fp], (hipStream_t)stream);
    if (rc != 0) return fail(H2E_ERR_HIP, rc < 0 ? "export: bad geometry" : std::string("export launch failed: ") + hipGetErrorString((hipError_t)rc));
    return 0;
}

static int ensure_fc(h2e_ctx* ctx, int fp) {
    if (!ctx->d_fc[fp]) {
        HIP_TRY(hipMalloc((void**)&ctx->d_fc[fp], sizeof(H2EFieldConsts)));
        HIP_TRY(hipMemcpy(ctx->d_fc[fp], &field_pair(fp).fc, sizeof(H2EFieldConsts), hipMemcpyHostToDevice));
        HIP_TRY((hipError_t)h2e_engine_set_consts(fp, &field_pair(fp).fc));
    }
    return 0;
}

int h2e_export_fixed(h2e_ctx* ctx, h2e_program* p, int region, int layout, int form, uint32_t n_instances, const void* d_inputs, void* d_out,
                     void* stream) {
    if (!ctx || !p || !d_out) return fail(H2E_ERR_INVALID, "null argument");
    if (region < 0 || region > 2) return fail(H2E_ERR_INVALID, "region must be 0 (base), 1 (range) or 2 (select)");
    if (layout != H2E_LAYOUT_ROWS && layout != H2E_LAYOUT_COLUMNS) return fail(H2E_ERR_INVALID, "bad layout");
    if (form != H2E_FORM_CANONICAL && form != H2E_FORM_MONTGOMERY) return fail(H2E_ERR_INVALID, "bad number form");
    h2e::Recorder& r = *p->rec;
    if (!r.emit_shape) return fail(H2E_ERR_INVALID, "the program was recorded without its shape (emit_shape = 0)");
    if (n_instances == 0) return fail(H2E_ERR_INVALID, "n_instances must be > 0");
    if (region == 0 && !r.fixed_patches.empty() && !d_inputs)
        return fail(H2E_ERR_INVALID, "this program has fixed cells made from instance inputs: d_inputs is needed");
    std::lock_guard<std::mutex> guard(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_fc(ctx, p->field_pair);
    if (rc) return rc;
    const uint32_t cols = region == 0 ? 9 : 2;
    const uint64_t rows = region == 0 ? p->base_rows : region == 1 ? p->range_rows : p->select_rows;
    const std::vector<uint32_t>& ids = region == 0 ? r.base_fix : region == 1 ? r.range_fix : r.select_fix;
    if (ids.size() < rows * cols) return fail(H2E_ERR_SHAPE, "internal: fixed array shorter than the region");
    if (!p->d_fix[region]) {
        HIP_TRY(hipMalloc((void**)&p->d_fix[region], std::max<size_t>(16, rows * cols * 4)));
        HIP_TRY(hipMemcpy(p->d_fix[region], ids.data(), rows * cols * 4, hipMemcpyHostToDevice));
    }
    if (!p->d_dict) {
        HIP_TRY(hipMalloc((void**)&p->d_dict, r.dict.size() * 32));
        HIP_TRY(hipMemcpy(p->d_dict, r.dict.data(), r.dict.size() * 32, hipMemcpyHostToDevice));
    }
    uint32_t n_patches = region == 0 ? (uint32_t)r.fixed_patches.size() : 0;
    if (n_patches && !p->d_patches) {
        HIP_TRY(hipMalloc((void**)&p->d_patches, (size_t)n_patches * 16));
        HIP_TRY(hipMemcpy(p->d_patches, p->patch_flat.data(), (size_t)n_patches * 16, hipMemcpyHostToDevice));
    }
    rc = h2e_engine_fixed(p->field_pair, p->d_fix[region], p->d_dict, rows, cols, layout == H2E_LAYOUT_COLUMNS, form == H2E_FORM_MONTGOMERY,
                          p->d_patches, n_patches, (const uint64_t*)d_inputs, r.n_input_slots, (uint32_t)r.fp.w_words, n_instances,
                          ctx->d_fc[p->field_pair], d_out, (hipStream_t)stream);
    if (rc != 0) return fail(H2E_ERR_HIP, rc < 0 ? "export_fixed: bad geometry" : std::string("export_fixed launch failed: ") + hipGetErrorString((hipError_t)rc));
    return 0;
}

int h2e_range_table(h2e_ctx* ctx, int form, void* d_out, void* stream) {
    if (!ctx || !d_out) return fail(H2E_ERR_INVALID, "null argument");
    if (form != H2E_FORM_CANONICAL && form != H2E_FORM_MONTGOMERY) return fail(H2E_ERR_INVALID, "bad number form");
    std::lock_guard<std::mutex> guard(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_fc(ctx, 0);
    if (rc) return rc;
    rc = h2e_engine_range_table(form == H2E_FORM_MONTGOMERY, ctx->d_fc[0], d_out, (hipStream_t)stream);
    if (rc != 0) return fail(H2E_ERR_HIP, std::string("range table launch failed: ") + hipGetErrorString((hipError_t)rc));
    return 0;
}

int h2e_export_copy_constraints(h2e_ctx* ctx, h2e_program* p, void* d_out, void* stream) {
    if (!ctx || !p || !d_out) return fail(H2E_ERR_INVALID, "null argument");
    h2e::Recorder& r = *p->rec;
    if (!r.emit_shape) return fail(H2E_ERR_INVALID, "the program was recorded without its shape (emit_shape = 0)");
    std::lock_guard<std::mutex> guard(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    size_t n = r.permutations.size();
    if (n && !p->d_perms) {
        HIP_TRY(hipMalloc((void**)&p->d_perms, n * 8));
        HIP_TRY(hipMemcpy(p->d_perms, p->perm_flat.data(), n * 8, hipMemcpyHostToDevice));
    }
    int rc = h2e_engine_copy_constraints(p->d_perms, n, d_out, (hipStream_t)stream);
    if (rc != 0) return fail(H2E_ERR_HIP, std::string("copy-constraint launch failed: ") + hipGetErrorString((hipError_t)rc));
    return 0;
}

// Device-side constraint check (include/h2e.h): the shape artefacts the kernels of checker.hip read are uploaded once per program.
static int ensure_check_artefacts(h2e_ctx* ctx, h2e_program* p) {
    if (p->check_ready) return 0;
    h2e::Recorder& r = *p->rec;
    const uint64_t rows[3] = {p->base_rows, p->range_rows, p->select_rows};
    const uint32_t fcols[3] = {9, 2, 2};
    const std::vector<uint32_t>* ids[3] = {&r.base_fix, &r.range_fix, &r.select_fix};
    for (int reg = 0; reg < 3; reg++) {
        if (ids[reg]->size() < rows[reg] * fcols[reg]) return fail(H2E_ERR_SHAPE, "internal: fixed array shorter than the region");
        const uint8_t* fl = nullptr;
        int rc = device_flags(p, reg, &fl);
        if (rc) return rc;
        if (reg > 0 && !p->d_fix[reg]) {
            HIP_TRY(hipMalloc((void**)&p->d_fix[reg], std::max<size_t>(16, rows[reg] * fcols[reg] * 4)));
            HIP_TRY(hipMemcpy(p->d_fix[reg], ids[reg]->data(), rows[reg] * fcols[reg] * 4, hipMemcpyHostToDevice));
        }
    }
    {   // base ids, cells made from instance inputs marked with their patch index
        std::vector<uint32_t> ck(r.base_fix.begin(), r.base_fix.begin() + rows[0] * 9);
        for (size_t k = 0; k < r.fixed_patches.size(); k++) {
            const h2e::FixedPatch& f = r.fixed_patches[k];
            if (f.col != 8 || f.row >= rows[0]) return fail(H2E_ERR_SHAPE, "internal: a fixed patch outside the constant column");
            ck[(size_t)f.row * 9 + 8] = 0x80000000u | (uint32_t)k;
        }
        if (!p->d_fix_ck) HIP_TRY(hipMalloc((void**)&p->d_fix_ck, std::max<size_t>(16, ck.size() * 4)));   // (a retry after a failed call keeps what it has)
        HIP_TRY(hipMemcpy(p->d_fix_ck, ck.data(), ck.size() * 4, hipMemcpyHostToDevice));
    }
    if (!p->d_dict) {
        HIP_TRY(hipMalloc((void**)&p->d_dict, r.dict.size() * 32));
        HIP_TRY(hipMemcpy(p->d_dict, r.dict.data(), r.dict.size() * 32, hipMemcpyHostToDevice));
    }
    uint32_t n_patches = (uint32_t)r.fixed_patches.size();
    if (n_patches && !p->d_patches) {
        HIP_TRY(hipMalloc((void**)&p->d_patches, (size_t)n_patches * 16));
        HIP_TRY(hipMemcpy(p->d_patches, p->patch_flat.data(), (size_t)n_patches * 16, hipMemcpyHostToDevice));
    }
    size_t n_perm = r.permutations.size();
    if (n_perm && !p->d_perms) {
        HIP_TRY(hipMalloc((void**)&p->d_perms, n_perm * 8));
        HIP_TRY(hipMemcpy(p->d_perms, p->perm_flat.data(), n_perm * 8, hipMemcpyHostToDevice));
    }
    {   // Fr constants of the check: once per process is enough, once per program is simpler and costs nothing
        const H2EFieldConsts& fc = field_pair(0).fc;
        HIP_TRY((hipError_t)h2e_engine_check_consts(fc.n, fc.n_minv, fc.n_r2));
    }
    if (!p->d_dict_m) HIP_TRY(hipMalloc((void**)&p->d_dict_m, r.dict.size() * 32));
    HIP_TRY((hipError_t)h2e_engine_check_to_mont(p->d_dict, p->d_dict_m, r.dict.size(), nullptr));
    {   // 2^(18 k), k = 0..5: the range gates' shifts (range_chip.rs:160-218)
        uint64_t sh[6][4];
        std::memset(sh, 0, sizeof(sh));
        for (int k = 0; k < 6; k++) sh[k][(18 * k) / 64] = 1ull << ((18 * k) % 64);
        if (!p->d_shifts_m) HIP_TRY(hipMalloc((void**)&p->d_shifts_m, sizeof(sh)));
        HIP_TRY(hipMemcpy(p->d_shifts_m, sh, sizeof(sh), hipMemcpyHostToDevice));
        HIP_TRY((hipError_t)h2e_engine_check_to_mont(p->d_shifts_m, p->d_shifts_m, 6, nullptr));
    }
    {   // the select chip's table: rows whose is_lookup cell is zero, keyed by their (fixed) encode cell; the all-zero rows of
        // the unused part of the circuit are one entry (row 0xffffffff)
        struct Ent {
            h2e::FrVal key;
            uint32_t row;
        };
        std::vector<Ent> tab;
        h2e::FrVal zero{};
        for (int i = 0; i < 4; i++) zero[i] = 0;
        tab.push_back(Ent{zero, 0xffffffffu});
        const uint64_t height = std::min<uint64_t>(rows[2], (uint64_t)r.select_height + 1);
        for (uint64_t row = 0; row < height; row++) {
            uint32_t enc = r.select_fix[row * 2], look = r.select_fix[row * 2 + 1];
            bool is_lookup = false;
            if (look) {
                const h2e::FrVal& lv = r.dict[look];
                is_lookup = (lv[0] | lv[1] | lv[2] | lv[3]) != 0;
            }
            if (is_lookup) continue;
            tab.push_back(Ent{enc ? r.dict[enc] : zero, (uint32_t)row});
        }
        std::sort(tab.begin(), tab.end(), [](const Ent& a, const Ent& b) {
            for (int i = 3; i >= 0; i--)
                if (a.key[i] != b.key[i]) return a.key[i] < b.key[i];
            return a.row < b.row;
        });
        std::vector<uint64_t> keys(tab.size() * 4);
        std::vector<uint32_t> krows(tab.size());
        for (size_t k = 0; k < tab.size(); k++) {
            for (int i = 0; i < 4; i++) keys[4 * k + i] = tab[k].key[i];
            krows[k] = tab[k].row;
        }
        p->n_sel_keys = (uint32_t)tab.size();
        if (!p->d_sel_keys) HIP_TRY(hipMalloc((void**)&p->d_sel_keys, keys.size() * 8));
        HIP_TRY(hipMemcpy(p->d_sel_keys, keys.data(), keys.size() * 8, hipMemcpyHostToDevice));
        if (!p->d_sel_key_rows) HIP_TRY(hipMalloc((void**)&p->d_sel_key_rows, krows.size() * 4));
        HIP_TRY(hipMemcpy(p->d_sel_key_rows, krows.data(), krows.size() * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipDeviceSynchronize());
    p->check_ready = true;
    (void)ctx;
    return 0;
}

int h2e_check(h2e_ctx* ctx, h2e_program* p, uint32_t n_instances, const void* d_inputs, const void* d_base, const void* d_range,
              const void* d_select, uint32_t classes, void* d_fail, void* stream) {
    if (!ctx || !p || !d_base || !d_range || !d_select || !d_fail) return fail(H2E_ERR_INVALID, "null argument");
    if (n_instances == 0) return fail(H2E_ERR_INVALID, "n_instances must be > 0");
    h2e::Recorder& r = *p->rec;
    if (!r.emit_shape) return fail(H2E_ERR_INVALID, "the program was recorded without its shape (emit_shape = 0): nothing to check the cells against");
    if (!r.fixed_patches.empty() && !d_inputs) return fail(H2E_ERR_INVALID, "this program has fixed cells made from instance inputs: d_inputs is needed");
    if (classes == 0) classes = (1u << H2E_CHECK_CLASSES) - 1u;
    std::lock_guard<std::mutex> guard(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_fc(ctx, p->field_pair);
    if (rc) return rc;
    rc = ensure_check_artefacts(ctx, p);
    if (rc) return rc;
    uint32_t n_patches = (uint32_t)r.fixed_patches.size();
    if (n_patches) {
        size_t need = (size_t)n_patches * n_instances * 32;
        if (need > p->patch_vals_cap) {
            if (p->d_patch_vals) {
                HIP_TRY(hipDeviceSynchronize());
                (void)hipFree(p->d_patch_vals);
                p->d_patch_vals = nullptr;
            }
            HIP_TRY(hipMalloc((void**)&p->d_patch_vals, need));
            p->patch_vals_cap = need;
        }
        rc = h2e_engine_check_patch_values(p->d_patches, n_patches, (const uint64_t*)d_inputs, r.n_input_slots, (uint32_t)r.fp.w_words, n_instances,
                                           p->d_patch_vals, (hipStream_t)stream);
        if (rc != 0) return fail(H2E_ERR_HIP, "check: patch values launch failed");
    }
    H2ECheckRegion regs[3];
    const void* adv[3] = {d_base, d_range, d_select};
    const uint64_t rows[3] = {p->base_rows, p->range_rows, p->select_rows};
    // gates run over the rows MockProver sees cells in: [0, height) for the base gate, one more row for the range and select
    // chips (their heights are "last used row + 1" with the quirks of context.rs:716-720; the oracle's checker does the same)
    const uint64_t heights[3] = {std::min<uint64_t>(rows[0], r.base_height), std::min<uint64_t>(rows[1], (uint64_t)r.range_height + 1),
                                 std::min<uint64_t>(rows[2], (uint64_t)r.select_height + 1)};
    for (int reg = 0; reg < 3; reg++) {
        regs[reg].adv = adv[reg];
        regs[reg].flags = p->d_flags[reg];
        regs[reg].fix = reg == 0 ? p->d_fix_ck : p->d_fix[reg];
        regs[reg].rows = rows[reg];
        regs[reg].height = heights[reg];
    }
    rc = h2e_engine_check(regs, p->d_dict, p->d_dict_m, p->d_shifts_m, p->d_patch_vals, n_patches, p->d_sel_keys, p->d_sel_key_rows, p->n_sel_keys,
                          p->d_perms, r.permutations.size(), n_instances, classes, (uint64_t*)d_fail, (hipStream_t)stream);
    if (rc != 0) return fail(H2E_ERR_HIP, rc < 0 ? "check: bad geometry" : std::string("check launch failed: ") + hipGetErrorString((hipError_t)rc));
    return 0;
}

int h2e_set_profiling(h2e_ctx* ctx, int enable) {
    if (!ctx) return fail(H2E_ERR_INVALID, "null ctx");
    std::lock_guard<std::mutex> guard(ctx->mu);
    ctx->profiling = enable != 0;
    return 0;
}
static int job_launch_ms(h2e_ctx* ctx, int job, float* ms, uint32_t cap) {
    JobSlot& J = ctx->slots[job];
    if (!J.profiled || !J.done) return 0;   // that run recorded no events (profiling was off when it was queued)
    HIP_TRY(hipEventSynchronize(J.done));
    // two numbers per launched segment: value chain (predictors + values-only replay), expansion (+ fix-up)
    uint32_t n = std::min<uint32_t>(J.n_launches, (uint32_t)(J.ev.size() / 4));
    for (uint32_t i = 0; i < n && 2 * i + 1 < cap; i++) {
        float t0 = 0, t1 = 0;
        hipError_t e = hipEventElapsedTime(&t0, J.ev[4 * i], J.ev[4 * i + 1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&t1, J.ev[4 * i + 2], J.ev[4 * i + 3]);
        if (e != hipSuccess) return fail(H2E_ERR_HIP, hipGetErrorString(e));
        ms[2 * i] = t0;
        ms[2 * i + 1] = t1;
    }
    return (int)n;
}
int h2e_last_run_launch_ms(h2e_ctx* ctx, float* ms, uint32_t cap) {
    if (!ctx || !ms) return fail(H2E_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(ctx->mu);
    if (ctx->last_slot < 0) return 0;
    return job_launch_ms(ctx, ctx->last_slot, ms, cap);
}
int h2e_job_launch_ms(h2e_ctx* ctx, int job, float* ms, uint32_t cap) {
    if (!ctx || !ms || job < 0 || job >= h2e_ctx::N_SLOTS) return fail(H2E_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> guard(ctx->mu);
    return job_launch_ms(ctx, job, ms, cap);
}

int h2e_last_run_expansion_launches(h2e_ctx* ctx, uint32_t* counts, uint32_t cap) {
    if (!ctx || !counts) return fail(H2E_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(ctx->mu);
    if (ctx->last_slot < 0) return 0;
    JobSlot& J = ctx->slots[ctx->last_slot];
    for (uint32_t i = 0; i < J.n_launches && i < cap; i++) counts[i] = i < J.x_kernels.size() ? J.x_kernels[i] : 1;
    return (int)J.n_launches;
}

static int cached_run(h2e_ctx* ctx, const std::string& key, std::function<int(h2e_program**)> make, uint32_t n_instances,
                      const void* d_inputs, void* d_base, void* d_range, void* d_select, void* d_status, void* stream) {
    if (!ctx) return fail(H2E_ERR_INVALID, "null ctx");
    auto it = ctx->cache.find(key);
    if (it == ctx->cache.end()) {
        h2e_program* p = nullptr;
        int rc = make(&p);
        if (rc) return rc;
        it = ctx->cache.emplace(key, p).first;
    }
    return h2e_run(ctx, it->second, n_instances, d_inputs, d_base, d_range, d_select, d_status, stream);
}

int h2e_int_mul_batch(h2e_ctx* ctx, int fp, uint32_t n, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
                      void* d_select, void* d_status, void* stream) {
    return cached_run(ctx, "int_mul_batch/" + std::to_string(fp) + "/" + std::to_string(n),
                      [&](h2e_program** p) { return h2e_program_int_mul_batch(fp, n, 0, p); }, n_instances, d_inputs, d_base,
                      d_range, d_select, d_status, stream);
}
int h2e_msm_bn256_tile(h2e_ctx* ctx, uint32_t n_points, uint32_t n_tiles, const void* d_inputs, void* d_base, void* d_range,
                       void* d_select, void* d_status, void* stream) {
    return cached_run(ctx, "msm_bn256_tile/" + std::to_string(n_points),
                      [&](h2e_program** p) { return h2e_program_msm_bn256_tile(n_points, 0, p); }, n_tiles, d_inputs, d_base,
                      d_range, d_select, d_status, stream);
}
int h2e_pairing_check_bn256(h2e_ctx* ctx, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
                            void* d_select, void* d_status, void* stream) {
    return cached_run(ctx, "pairing_check_bn256", [&](h2e_program** p) { return h2e_program_pairing_check_bn256(0, p); },
                      n_instances, d_inputs, d_base, d_range, d_select, d_status, stream);
}
int h2e_pairing_check_bls12_381(h2e_ctx* ctx, uint32_t n_instances, const void* d_inputs, void* d_base, void* d_range,
                                void* d_select, void* d_status, void* stream) {
    return cached_run(ctx, "pairing_check_bls12_381",
                      [&](h2e_program** p) { return h2e_program_pairing_check_bls12_381(0, p); }, n_instances, d_inputs,
                      d_base, d_range, d_select, d_status, stream);
}

// =================================================================================================================
// Operator API: a device-resident Context (include/h2e.h "operator API").  The reference's operator surface is a
// Context you call chip ops on; its own seam for running part of the work elsewhere is fork-at-offset / merge
// (ParallelClone, src/circuit/ecc_chip.rs:64-77).  h2e_records is that Context for a batch of instances: advice arrays
// in HBM, cursors / heights / msm prefix / shape artefacts on the host.  Every op records one program that starts at the
// current offsets with its operands as handles to earlier rows, runs it on the shared arrays and advances the cursors.
struct h2e_records {
    h2e_ctx* ctx = nullptr;
    int field_pair = 0, scalar_field = -1;
    uint32_t n_instances = 0;
    bool emit_shape = true, own_arrays = false;
    uint64_t cap[3] = {0, 0, 0};
    void* d_arr[3] = {nullptr, nullptr, nullptr};
    uint32_t* d_status = nullptr;
    // Context (src/context.rs:40-46) + Records heights (:297-299) + NativeScalarEccContext.1 / msm_prefix
    uint64_t off[3] = {0, 0, 0}, height[3] = {0, 0, 0};
    size_t msm_prefix = 0;
    uint64_t n_advice_cells = 0, n_ops = 0;
    // accumulated shape artefacts
    std::vector<h2e::FrVal> dict;
    std::unordered_map<h2e::FrVal, uint32_t, h2e::FrValHash> dict_map;
    std::vector<uint32_t> fix[3];
    std::vector<uint8_t> flags[3];
    std::vector<uint32_t> perm_flat;
    std::vector<uint32_t> patch_flat;   // [row, fixed col, op index << 16 | input slot, limb]
    void* d_dummy = nullptr;            // input vector of ops that take none (the engine wants a valid pointer)
    h2e_records() { dict.push_back(h2e::FrVal{0, 0, 0, 0}); }
    ~h2e_records() {
        if (ctx) {
            (void)hipSetDevice(ctx->device);
            (void)hipDeviceSynchronize();
        }
        (void)hipFree(d_dummy);
        if (own_arrays) {
            for (int i = 0; i < 3; i++) (void)hipFree(d_arr[i]);
            (void)hipFree(d_status);
        }
    }
    uint32_t intern(const h2e::FrVal& v) {
        auto it = dict_map.find(v);
        if (it != dict_map.end()) return it->second;
        uint32_t id = (uint32_t)dict.size();
        dict.push_back(v);
        dict_map.emplace(v, id);
        return id;
    }
};

namespace {
const int FIXC[3] = {9, 2, 2}, ADVC[3] = {5, 3, 2};
static h2e::AssignedInteger to_int(const h2e_int& a) {
    h2e::AssignedInteger r;
    for (int i = 0; i < H2E_MAX_L; i++) r.limbs_le[i] = a.limbs[i];
    r.native = a.native;
    r.times = a.times;
    return r;
}
h2e_int from_int(const h2e::AssignedInteger& a) {
    h2e_int r;
    for (int i = 0; i < H2E_MAX_L; i++) r.limbs[i] = a.limbs_le[i];
    r.native = a.native;
    r.times = (uint32_t)a.times;
    return r;
}
static h2e::AssignedPoint to_point(const h2e_point& p) { return h2e::AssignedPoint{to_int(p.x), to_int(p.y), h2e::AssignedCondition{h2e::AssignedValue{p.z}}}; }
h2e_point from_point(const h2e::AssignedPoint& p) {
    h2e_point r;
    r.x = from_int(p.x);
    r.y = from_int(p.y);
    r.z = p.z.v.ref;
    return r;
}

// Record one op at the records' current state (or take its program from the context's op cache), run it, merge its shape
// artefacts and advance the Context.  `key` names the op and everything its recording depends on besides the records' state;
// `outs` lists the caller's output handles (filled by `body` when the op is recorded, from the cache otherwise).
struct OpOut {
    void* ptr;
    size_t bytes;
};
static std::string key_of(const char* name, std::initializer_list<std::pair<const void*, size_t>> blobs) {
    std::string k(name);
    for (auto& b : blobs) {
        k.push_back('|');
        if (b.first) k.append((const char*)b.first, b.second);
    }
    return k;
}
int records_op(h2e_records* R, const std::string& key, uint32_t n_slots, const void* d_inputs, void* stream, std::initializer_list<OpOut> outs,
               const std::function<void(h2e::Recorder&, h2e::NativeScalarEccContext&, uint32_t)>& body) {
    if (!R) return fail(H2E_ERR_INVALID, "null records");
    if (n_slots && !d_inputs) return fail(H2E_ERR_INVALID, "the op takes inputs: d_inputs is null");
    if (R->n_ops >= 65535) return fail(H2E_ERR_SHAPE, "records: more than 65535 ops (the fixed-patch list packs the op index in 16 bits)");
    h2e_ctx* ctx = R->ctx;
    std::string full = key;
    {
        uint64_t st[9] = {R->off[0], R->off[1], R->off[2], R->height[0], R->height[1], R->height[2], (uint64_t)R->msm_prefix,
                          (uint64_t)(R->field_pair * 8 + (R->scalar_field + 1)), (uint64_t)(R->emit_shape ? 1 : 0) | ((uint64_t)n_slots << 8)};
        full.push_back('#');
        full.append((const char*)st, sizeof(st));
    }
    h2e_program* p = nullptr;
    size_t prefix_after = R->msm_prefix;   // (applied when the op has run: a failing op leaves the records' state as it was)
    struct Release {   // the entry cannot be evicted while this call runs its program
        h2e_ctx* ctx;
        const std::string* key;
        bool held = false;
        ~Release() {
            if (!held) return;
            std::lock_guard<std::mutex> g(ctx->op_mu);
            auto it = ctx->op_cache.find(*key);
            if (it != ctx->op_cache.end() && it->second.in_use) it->second.in_use--;
        }
    } release{ctx, &full};
    {
        std::lock_guard<std::mutex> g(ctx->op_mu);
        auto it = ctx->op_cache.find(full);
        if (it != ctx->op_cache.end()) {
            p = it->second.prog;
            size_t k = 0;
            for (auto& o : outs) {
                if (o.ptr && k < it->second.outs.size() && it->second.outs[k].size() == o.bytes) std::memcpy(o.ptr, it->second.outs[k].data(), o.bytes);
                k++;
            }
            prefix_after = it->second.msm_prefix_after;
            it->second.last_use = ++ctx->op_tick;
            it->second.in_use++;
            release.held = true;
            ctx->op_hits++;
        }
    }
    if (!p) {
        std::unique_ptr<h2e_program> np(new h2e_program());
        np->field_pair = R->field_pair;
        try {
            np->rec.reset(new h2e::Recorder(field_pair(R->field_pair)));
            h2e::Recorder& r = *np->rec;
            r.emit_shape = R->emit_shape;
            // clone_with_offset of the caller's context (context.rs:145-158): cursors and heights carry over
            r.base_offset = R->off[0];
            r.range_offset = R->off[1];
            r.select_offset = R->off[2];
            r.base_height = R->height[0];
            r.range_height = R->height[1];
            r.select_height = R->height[2];
            h2e::NativeScalarEccContext ecc(r, R->field_pair == H2E_FIELD_BN256_FQ ? h2e::bn256_g1_params() : h2e::bls12_381_g1_params(), R->msm_prefix);
            ecc.scalar_field = R->scalar_field;
            ecc.with_select = ecc.has_select_chip();
            uint32_t s0 = r.alloc_inputs(std::max<uint32_t>(1, n_slots));
            body(r, ecc, s0);
            np->finish();
            prefix_after = ecc.msm_prefix;
        } catch (std::exception& e) {
            return fail(H2E_ERR_SHAPE, e.what());
        }
        if (np->base_rows > R->cap[0] || np->range_rows > R->cap[1] || np->select_rows > R->cap[2])
            return fail(H2E_ERR_SHAPE, "records: the op does not fit the arrays' capacity (like HALO2ECC_S_MAX_ROWS, src/context.rs:36)");
        std::lock_guard<std::mutex> g(ctx->op_mu);
        h2e_ctx::OpEntry& e = ctx->op_cache[full];
        if (!e.prog) {
            e.prog = np.release();
            for (auto& o : outs) e.outs.emplace_back((const uint8_t*)o.ptr, (const uint8_t*)o.ptr + (o.ptr ? o.bytes : 0));
            e.msm_prefix_after = prefix_after;
            ctx->op_misses++;
        }
        e.last_use = ++ctx->op_tick;
        e.in_use++;
        release.held = true;
        p = e.prog;
        ctx->op_cache_trim();
    }
    if (p->base_rows > R->cap[0] || p->range_rows > R->cap[1] || p->select_rows > R->cap[2])
        return fail(H2E_ERR_SHAPE, "records: the op does not fit the arrays' capacity (like HALO2ECC_S_MAX_ROWS, src/context.rs:36)");
    h2e::Recorder& r = *p->rec;
    int rc = 0;
    if (!r.tape.empty()) {
        const void* in = d_inputs;
        if (!in) {   // ops without inputs still get a valid (unused) pointer
            if (!R->d_dummy) {
                HIP_TRY(hipSetDevice(ctx->device));
                HIP_TRY(hipMalloc(&R->d_dummy, (size_t)R->n_instances * 64));
                HIP_TRY(hipMemset(R->d_dummy, 0, (size_t)R->n_instances * 64));
            }
            in = R->d_dummy;
        }
        rc = h2e_run(ctx, p, R->n_instances, in, R->d_arr[0], R->d_arr[1], R->d_arr[2], R->d_status, stream);
    }
    if (rc) return rc;
    R->msm_prefix = prefix_after;
    // merge (ParallelClone::merge + apply_offset_diff)
    uint64_t before[3] = {R->off[0], R->off[1], R->off[2]};
    R->off[0] = r.base_offset;
    R->off[1] = r.range_offset;
    R->off[2] = r.select_offset;
    R->height[0] = r.base_height;
    R->height[1] = r.range_height;
    R->height[2] = r.select_height;
    if (R->emit_shape) {
        const std::vector<uint32_t>* pfix[3] = {&r.base_fix, &r.range_fix, &r.select_fix};
        const std::vector<uint8_t>* pfl[3] = {&r.base_flags, &r.range_flags, &r.select_flags};
        uint64_t rows[3] = {p->base_rows, p->range_rows, p->select_rows};
        std::vector<uint32_t> idmap(r.dict.size(), 0);
        for (size_t i = 1; i < r.dict.size(); i++) idmap[i] = R->intern(r.dict[i]);
        for (int reg = 0; reg < 3; reg++) {
            if (R->fix[reg].size() < rows[reg] * FIXC[reg]) R->fix[reg].resize(rows[reg] * FIXC[reg], 0);
            if (R->flags[reg].size() < rows[reg] * ADVC[reg]) R->flags[reg].resize(rows[reg] * ADVC[reg], 0);
            for (uint64_t row = before[reg]; row < rows[reg]; row++) {
                for (int c = 0; c < FIXC[reg]; c++) {
                    size_t k = row * FIXC[reg] + c;
                    if (k < pfix[reg]->size() && (*pfix[reg])[k]) R->fix[reg][k] = idmap[(*pfix[reg])[k]];
                }
                for (int c = 0; c < ADVC[reg]; c++) {
                    size_t k = row * ADVC[reg] + c;
                    if (k < pfl[reg]->size()) R->flags[reg][k] |= (*pfl[reg])[k];
                }
            }
        }
        auto set_perm = [&](uint32_t cell) {
            uint32_t reg = H2E_REF_REGION(cell);
            size_t k = (size_t)H2E_REF_ROW(cell) * ADVC[reg] + H2E_REF_COL(cell);
            if (R->flags[reg].size() <= k) R->flags[reg].resize(k + 1, 0);
            R->flags[reg][k] |= 2;
        };
        for (auto& pr : r.permutations) {
            R->perm_flat.push_back(pr.first);
            R->perm_flat.push_back(pr.second);
            set_perm(pr.first);
            set_perm(pr.second);
        }
        for (auto& fpch : r.fixed_patches) {
            R->patch_flat.push_back(fpch.row);
            R->patch_flat.push_back(fpch.col);
            R->patch_flat.push_back((uint32_t)(R->n_ops << 16) | fpch.input_slot);
            R->patch_flat.push_back((uint32_t)fpch.limb);
        }
        R->n_advice_cells += r.n_advice_cells;
    }
    R->n_ops++;
    return 0;
}
}  // namespace

int h2e_records_create(h2e_ctx* ctx, int field_pair_id, int scalar_field, uint32_t n_instances, uint64_t base_rows, uint64_t range_rows,
                       uint64_t select_rows, int emit_shape, h2e_records** out) {
    if (!ctx || !out) return fail(H2E_ERR_INVALID, "null argument");
    if (field_pair_id < 0 || field_pair_id > 2 || scalar_field < -1 || scalar_field > 2) return fail(H2E_ERR_INVALID, "bad field pair");
    if (n_instances == 0 || base_rows == 0 || range_rows == 0 || select_rows == 0) return fail(H2E_ERR_INVALID, "empty records");
    // a flag word, not a boolean: a caller's "true" of 2 or -1 must not silently mean "no shape" / "no select chip"
    if (emit_shape & ~(H2E_RECORDS_EMIT_SHAPE | H2E_RECORDS_NO_SELECT_CHIP)) return fail(H2E_ERR_INVALID, "h2e_records_create: unknown bits in the flag word (H2E_RECORDS_*)");
    h2e_records* R = new h2e_records();
    R->ctx = ctx;
    R->field_pair = field_pair_id;
    R->scalar_field = scalar_field;
    R->n_instances = n_instances;
    R->emit_shape = (emit_shape & H2E_RECORDS_EMIT_SHAPE) != 0;
    // NativeScalarEccContext::new_without_select_chip (src/context.rs:201-205): the msm prefix is usize::MAX and msm_unsafe takes
    // the bisection form (src/circuit/ecc_chip.rs:373-408 dispatches on has_select_chip, native_scalar_ecc_chip.rs:27-46)
    if (emit_shape & H2E_RECORDS_NO_SELECT_CHIP) R->msm_prefix = (size_t)-1;
    R->cap[0] = base_rows;
    R->cap[1] = range_rows;
    R->cap[2] = select_rows;
    R->own_arrays = true;
    hipError_t e = hipSetDevice(ctx->device);
    for (int i = 0; i < 3 && e == hipSuccess; i++) {
        size_t bytes = (size_t)R->cap[i] * ADVC[i] * 32 * n_instances;
        e = hipMalloc(&R->d_arr[i], bytes);
        if (e == hipSuccess) e = hipMemset(R->d_arr[i], 0, bytes);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&R->d_status, (size_t)n_instances * 4);
    if (e == hipSuccess) e = hipMemset(R->d_status, 0, (size_t)n_instances * 4);
    if (e != hipSuccess) {
        delete R;
        return fail(H2E_ERR_HIP, std::string("records: ") + hipGetErrorString(e));
    }
    *out = R;
    return 0;
}
int h2e_records_attach(h2e_ctx* ctx, int field_pair_id, int scalar_field, uint32_t n_instances, void* d_base, void* d_range, void* d_select,
                       void* d_status, const uint64_t capacity_rows[3], const uint64_t offset0[3], uint64_t msm_prefix0, int emit_shape,
                       h2e_records** out) {
    if (!ctx || !out || !capacity_rows || !offset0) return fail(H2E_ERR_INVALID, "null argument");
    if (!d_base || !d_range || !d_select || !d_status) return fail(H2E_ERR_INVALID, "null device pointer");
    if (field_pair_id < 0 || field_pair_id > 2 || scalar_field < -1 || scalar_field > 2) return fail(H2E_ERR_INVALID, "bad field pair");
    if (n_instances == 0) return fail(H2E_ERR_INVALID, "empty records");
    for (int i = 0; i < 3; i++)
        if (capacity_rows[i] == 0 || offset0[i] >= capacity_rows[i] || capacity_rows[i] > (1ull << 26))
            return fail(H2E_ERR_INVALID, "offsets must lie inside the arrays (at most 2^26 rows)");
    h2e_records* R = new h2e_records();
    R->ctx = ctx;
    R->field_pair = field_pair_id;
    R->scalar_field = scalar_field;
    R->n_instances = n_instances;
    R->emit_shape = emit_shape != 0;
    R->own_arrays = false;
    R->d_arr[0] = d_base;
    R->d_arr[1] = d_range;
    R->d_arr[2] = d_select;
    R->d_status = (uint32_t*)d_status;
    for (int i = 0; i < 3; i++) {
        R->cap[i] = capacity_rows[i];
        R->off[i] = offset0[i];
        R->height[i] = offset0[i];
    }
    R->msm_prefix = (size_t)msm_prefix0;
    *out = R;
    return 0;
}
void h2e_records_destroy(h2e_records* R) { delete R; }
int h2e_records_arrays(h2e_records* R, void** d_base, void** d_range, void** d_select, void** d_status) {
    if (!R) return fail(H2E_ERR_INVALID, "null records");
    if (d_base) *d_base = R->d_arr[0];
    if (d_range) *d_range = R->d_arr[1];
    if (d_select) *d_select = R->d_arr[2];
    if (d_status) *d_status = R->d_status;
    return 0;
}
int h2e_records_shape(const h2e_records* R, h2e_shape* out) {
    if (!R || !out) return fail(H2E_ERR_INVALID, "null argument");
    std::memset(out, 0, sizeof(*out));
    out->field_pair = R->field_pair;
    out->slot_words = field_pair(R->field_pair).w_words;
    out->base_offset = R->off[0];
    out->range_offset = R->off[1];
    out->select_offset = R->off[2];
    out->base_height = R->height[0];
    out->range_height = R->height[1];
    out->select_height = R->height[2];
    out->base_rows = R->cap[0];
    out->range_rows = R->cap[1];
    out->select_rows = R->cap[2];
    out->n_advice_cells = R->n_advice_cells;
    out->n_permutations = R->perm_flat.size() / 2;
    out->n_dict = R->dict.size();
    out->n_fixed_patches = R->patch_flat.size() / 4;
    out->n_segments = 0;
    out->n_ops = R->n_ops;
    if (R->emit_shape) {
        h2e_records* W = const_cast<h2e_records*>(R);
        for (int reg = 0; reg < 3; reg++) {   // the views cover the arrays' whole capacity
            W->fix[reg].resize(R->cap[reg] * FIXC[reg], 0);
            W->flags[reg].resize(R->cap[reg] * ADVC[reg], 0);
        }
        out->dict = (const uint64_t*)R->dict.data();
        out->base_fix = R->fix[0].data();
        out->range_fix = R->fix[1].data();
        out->select_fix = R->fix[2].data();
        out->base_flags = R->flags[0].data();
        out->range_flags = R->flags[1].data();
        out->select_flags = R->flags[2].data();
        out->permutations = R->perm_flat.data();
        out->fixed_patches = R->patch_flat.data();
    }
    return 0;
}

// ---- ops: same names and argument meaning as the reference's traits -------------------------------------------------
int h2e_op_assign_w(h2e_records* R, const void* d_inputs, h2e_int* out, void* stream) {   // IntegerChipOps::assign_w (integer_chip.rs:236-258)
    if (!out) return fail(H2E_ERR_INVALID, "out is null");
    return records_op(R, key_of("assign_w", {}), 1, d_inputs, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t s) { *out = from_int(r.assign_w(s)); });
}
int h2e_op_assign(h2e_records* R, const void* d_inputs, uint32_t* out_cell, void* stream) {   // BaseChipOps::assign (base_chip.rs:351-355)
    if (!out_cell) return fail(H2E_ERR_INVALID, "out is null");
    return records_op(R, key_of("assign", {}), 1, d_inputs, stream, {OpOut{out_cell, sizeof(*out_cell)}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t s) { *out_cell = r.assign(s).ref; });
}
int h2e_op_int(h2e_records* R, int which, const h2e_int* a, const h2e_int* b, h2e_int* out, uint32_t* out_cond, void* stream) {
    const bool binary = which == H2E_INT_ADD || which == H2E_INT_SUB || which == H2E_INT_MUL || which == H2E_INT_DIV || which == H2E_INT_IS_EQUAL ||
                        which == H2E_INT_ASSERT_EQUAL;
    const bool no_out = which == H2E_INT_IS_ZERO || which == H2E_INT_IS_EQUAL || which == H2E_INT_ASSERT_EQUAL;
    if (!a || (binary && !b) || (!no_out && !out)) return fail(H2E_ERR_INVALID, "null operand");
    if (no_out) out = nullptr;
    return records_op(R, key_of("int", {{&which, sizeof(which)}, {a, sizeof(*a)}, {b, b ? sizeof(*b) : 0}}), 0, nullptr, stream, {OpOut{out, out ? sizeof(*out) : 0}, OpOut{out_cond, out_cond ? sizeof(*out_cond) : 0}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t) {
        h2e::AssignedInteger x = to_int(*a), y = b ? to_int(*b) : h2e::AssignedInteger();
        switch (which) {
            case H2E_INT_ADD: *out = from_int(r.int_add(x, y)); break;
            case H2E_INT_SUB: *out = from_int(r.int_sub(x, y)); break;
            case H2E_INT_MUL: *out = from_int(r.int_mul(x, y)); break;
            case H2E_INT_REDUCE: *out = from_int(r.reduce(x)); break;
            case H2E_INT_DIV: {
                auto d = r.int_div(x, y);
                *out = from_int(d.second);
                if (out_cond) *out_cond = d.first.v.ref;
            } break;
            case H2E_INT_NEG: *out = from_int(r.int_neg(x)); break;
            case H2E_INT_SQUARE: *out = from_int(r.int_square(x)); break;
            case H2E_INT_UNSAFE_INVERT: *out = from_int(r.int_unsafe_invert(x)); break;
            case H2E_INT_IS_ZERO: {
                h2e::AssignedCondition c = r.is_int_zero(x);
                if (out_cond) *out_cond = c.v.ref;
            } break;
            case H2E_INT_IS_EQUAL: {
                h2e::AssignedCondition c = r.is_int_equal(x, y);
                if (out_cond) *out_cond = c.v.ref;
            } break;
            case H2E_INT_ASSERT_EQUAL: r.assert_int_equal(x, y); break;
            default: throw std::runtime_error("h2e_op_int: unknown op");
        }
    });
}
int h2e_op_int_mul_small_constant(h2e_records* R, const h2e_int* a, uint64_t k, h2e_int* out, void* stream) {   // integer_chip.rs:618-658
    if (!a || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("int_mul_small", {{a, sizeof(*a)}, {&k, sizeof(k)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t) { *out = from_int(r.int_mul_small_constant(to_int(*a), k)); });
}
int h2e_op_assign_int_constant(h2e_records* R, const uint64_t* w_words, h2e_int* out, void* stream) {   // integer_chip.rs:580-598
    if (!w_words || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("int_const", {{w_words, (size_t)field_pair(R ? R->field_pair : 0).w_words * 8}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t) {
        *out = from_int(r.assign_int_constant(h2e::HBig::from_words(w_words, r.fp.w_words)));
    });
}
int h2e_op_bisec_int(h2e_records* R, uint32_t cond_cell, const h2e_int* a, const h2e_int* b, h2e_int* out, void* stream) {   // integer_chip.rs:660-681
    if (!a || !b || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("bisec_int", {{&cond_cell, sizeof(cond_cell)}, {a, sizeof(*a)}, {b, sizeof(*b)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t) {
        *out = from_int(r.bisec_int(h2e::AssignedCondition{h2e::AssignedValue{cond_cell}}, to_int(*a), to_int(*b)));
    });
}
namespace {
static h2e::AssignedFq2 to_fq2(const h2e_int* a) { return h2e::AssignedFq2{to_int(a[0]), to_int(a[1])}; }
static h2e::AssignedFq6 to_fq6(const h2e_int* a) { return h2e::AssignedFq6{to_fq2(a), to_fq2(a + 2), to_fq2(a + 4)}; }
static h2e::AssignedFq12 to_fq12(const h2e_int* a) { return h2e::AssignedFq12{to_fq6(a), to_fq6(a + 6)}; }
void from_fq2(const h2e::AssignedFq2& x, h2e_int* o) {
    o[0] = from_int(x.c0);
    o[1] = from_int(x.c1);
}
void from_fq6(const h2e::AssignedFq6& x, h2e_int* o) {
    from_fq2(x.c0, o);
    from_fq2(x.c1, o + 2);
    from_fq2(x.c2, o + 4);
}
void from_fq12(const h2e::AssignedFq12& x, h2e_int* o) {
    from_fq6(x.c0, o);
    from_fq6(x.c1, o + 6);
}
static std::unique_ptr<h2e::PairingOps> tower_of(h2e::Recorder& r) {
    if (r.fp.id == H2E_FIELD_BN256_FQ) return std::unique_ptr<h2e::PairingOps>(new h2e::Bn256PairingOps(r));
    if (r.fp.id == H2E_FIELD_BLS12_381_FQ) return std::unique_ptr<h2e::PairingOps>(new h2e::Bls12381PairingOps(r));
    throw std::runtime_error("no extension tower over this field");
}
}  // namespace
// Fq2ChipOps / Fq6ChipOps / Fq12ChipOps (src/circuit/fq12.rs:24-459) on assigned elements
int h2e_op_fq(h2e_records* R, int degree, int which, const h2e_int* a, const h2e_int* b, uint64_t imm, h2e_int* out, void* stream) {
    if (!a || (degree != 2 && degree != 6 && degree != 12)) return fail(H2E_ERR_INVALID, "bad argument");
    bool binary = which == H2E_FQ_ADD || which == H2E_FQ_SUB || which == H2E_FQ_MUL || which == H2E_FQ_ASSERT_EQUAL;
    if ((binary && !b) || (which != H2E_FQ_ASSERT_EQUAL && !out)) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("fq", {{&degree, sizeof(degree)}, {&which, sizeof(which)}, {a, sizeof(*a) * (size_t)degree}, {b, b ? sizeof(*b) * (size_t)degree : 0}, {&imm, sizeof(imm)}}), 0, nullptr, stream, {OpOut{out, out ? sizeof(*out) * (size_t)degree : 0}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t) {
        std::unique_ptr<h2e::PairingOps> t = tower_of(r);
        r.auto_cut_every = pairing_cut_every(r.fp.id == H2E_FIELD_BN256_FQ ? 0 : 1);
        auto bad = [] { throw std::runtime_error("h2e_op_fq: no such op at this degree"); };
        if (degree == 2) {
            h2e::AssignedFq2 x = to_fq2(a), y = b ? to_fq2(b) : h2e::AssignedFq2(), o;
            switch (which) {
                case H2E_FQ_ADD: o = t->fq2_add(x, y); break;
                case H2E_FQ_SUB: o = t->fq2_sub(x, y); break;
                case H2E_FQ_MUL: o = t->fq2_mul(x, y); break;
                case H2E_FQ_SQUARE: o = t->fq2_square(x); break;
                case H2E_FQ_NEG: o = t->fq2_neg(x); break;
                case H2E_FQ_DOUBLE: o = t->fq2_double(x); break;
                case H2E_FQ_CONJUGATE: o = t->fq2_conjugate(x); break;
                case H2E_FQ_UNSAFE_INVERT: o = t->fq2_unsafe_invert(x); break;
                case H2E_FQ_MUL_BY_NONRESIDUE: o = t->fq2_mul_by_nonresidue(x); break;
                case H2E_FQ_FROBENIUS_MAP: o = t->fq2_frobenius_map(x, (size_t)imm); break;
                case H2E_FQ_REDUCE: o = t->fq2_reduce(x); break;
                case H2E_FQ_ASSERT_EQUAL: t->fq2_assert_equal(x, y); return;
                default: bad();
            }
            from_fq2(o, out);
        } else if (degree == 6) {
            h2e::AssignedFq6 x = to_fq6(a), y = b ? to_fq6(b) : h2e::AssignedFq6(), o;
            switch (which) {
                case H2E_FQ_ADD: o = t->fq6_add(x, y); break;
                case H2E_FQ_SUB: o = t->fq6_sub(x, y); break;
                case H2E_FQ_MUL: o = t->fq6_mul(x, y); break;
                case H2E_FQ_SQUARE: o = t->fq6_square(x); break;
                case H2E_FQ_NEG: o = t->fq6_neg(x); break;
                case H2E_FQ_DOUBLE: o = t->fq6_double(x); break;
                case H2E_FQ_UNSAFE_INVERT: o = t->fq6_unsafe_invert(x); break;
                case H2E_FQ_MUL_BY_NONRESIDUE: o = t->fq6_mul_by_nonresidue(x); break;
                case H2E_FQ_FROBENIUS_MAP: o = t->fq6_frobenius_map(x, (size_t)imm); break;
                case H2E_FQ_REDUCE: o = t->fq6_reduce(x); break;
                case H2E_FQ_ASSERT_EQUAL: t->fq6_assert_equal(x, y); return;
                default: bad();
            }
            from_fq6(o, out);
        } else {
            h2e::AssignedFq12 x = to_fq12(a), y = b ? to_fq12(b) : h2e::AssignedFq12(), o;
            switch (which) {
                case H2E_FQ_ADD: o = t->fq12_add(x, y); break;
                case H2E_FQ_SUB: o = t->fq12_sub(x, y); break;
                case H2E_FQ_MUL: o = t->fq12_mul(x, y); break;
                case H2E_FQ_SQUARE: o = t->fq12_square(x); break;
                case H2E_FQ_NEG: o = t->fq12_neg(x); break;
                case H2E_FQ_DOUBLE: o = t->fq12_double(x); break;
                case H2E_FQ_CONJUGATE: o = t->fq12_conjugate(x); break;
                case H2E_FQ_UNSAFE_INVERT: o = t->fq12_unsafe_invert(x); break;
                case H2E_FQ_FROBENIUS_MAP: o = t->fq12_frobenius_map(x, (size_t)imm); break;
                case H2E_FQ_CYCLOTOMIC_SQUARE: o = t->fq12_cyclotomic_square(x); break;
                case H2E_FQ_REDUCE: o = t->fq12_reduce(x); break;
                case H2E_FQ_ASSERT_EQUAL: t->fq12_assert_eq(x, y); return;
                default: bad();
            }
            from_fq12(o, out);
        }
    });
}
int h2e_op_assign_points(h2e_records* R, uint32_t n, const void* d_inputs, h2e_point* out, void* stream) {   // EccChipBaseOps::assign_point x n
    if (!out || n == 0) return fail(H2E_ERR_INVALID, "bad argument");
    return records_op(R, key_of("assign_points", {{&n, sizeof(n)}}), 3 * n, d_inputs, stream, {OpOut{out, sizeof(*out) * (size_t)n}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& ecc, uint32_t s) {
        std::vector<h2e::AssignedPoint> pts = ecc.assign_points_from_inputs(n, s);
        for (uint32_t k = 0; k < n; k++) out[k] = from_point(pts[k]);
    });
}
int h2e_op_assign_scalars(h2e_records* R, uint32_t n, const void* d_inputs, h2e_int* out, void* stream) {   // ctx.assign / scalar_integer_ctx.assign_w x n
    if (!out || n == 0) return fail(H2E_ERR_INVALID, "bad argument");
    return records_op(R, key_of("assign_scalars", {{&n, sizeof(n)}}), n, d_inputs, stream, {OpOut{out, sizeof(*out) * (size_t)n}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& ecc, uint32_t s) {
        std::vector<h2e::AssignedInteger> sc = ecc.assign_scalars_from_inputs(n, s);
        for (uint32_t k = 0; k < n; k++) out[k] = from_int(sc[k]);
    });
}
int h2e_op_msm_unsafe(h2e_records* R, uint32_t n, const h2e_point* points, const h2e_int* scalars, const void* d_inputs, h2e_point* out,
                      void* stream) {   // EccChipScalarOps::msm_unsafe (ecc_chip.rs:373-408); inputs: generator (x, y), r1 (x, y), r2 (x, y)
    if (!points || !scalars || !out || n == 0) return fail(H2E_ERR_INVALID, "bad argument");
    return records_op(R, key_of("msm_unsafe", {{&n, sizeof(n)}, {points, sizeof(*points) * (size_t)n}, {scalars, sizeof(*scalars) * (size_t)n}}), 6, d_inputs, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& ecc, uint32_t s) {
        std::vector<h2e::AssignedPoint> pts;
        std::vector<h2e::AssignedInteger> sc;
        for (uint32_t k = 0; k < n; k++) {
            pts.push_back(to_point(points[k]));
            sc.push_back(to_int(scalars[k]));
        }
        h2e::NativeScalarEccContext::MsmInputs mi{s + 2, s + 3, s + 4, s + 5};
        *out = from_point(ecc.msm_unsafe(pts, sc, mi, s, s + 1));
    });
}
int h2e_op_ecc_assert_equal(h2e_records* R, const h2e_point* a, const h2e_point* b, void* stream) {   // ecc_chip.rs:644-658
    if (!a || !b) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("ecc_assert_equal", {{a, sizeof(*a)}, {b, sizeof(*b)}}), 0, nullptr, stream, {}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& ecc, uint32_t) { ecc.ecc_assert_equal(to_point(*a), to_point(*b)); });
}
// ---- the complete-addition / curvature surface of EccChipBaseOps ----
namespace {
static h2e::AssignedPointWithCurvature to_pc(const h2e_point_c& a) {
    return h2e::AssignedPointWithCurvature{to_int(a.p.x), to_int(a.p.y), h2e::AssignedCondition{h2e::AssignedValue{a.p.z}},
                                           h2e::AssignedCurvature{to_int(a.cv), h2e::AssignedCondition{h2e::AssignedValue{a.cz}}}};
}
h2e_point_c from_pc(const h2e::AssignedPointWithCurvature& a) {
    h2e_point_c r;
    r.p = from_point(a.to_point());
    r.cv = from_int(a.curvature.v);
    r.cz = a.curvature.z.v.ref;
    return r;
}
}  // namespace
int h2e_op_to_point_with_curvature(h2e_records* R, const h2e_point* a, h2e_point_c* out, void* stream) {   // ecc_chip.rs:695-708
    if (!a || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("to_point_with_curvature", {{a, sizeof(*a)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) { *out = from_pc(e.to_point_with_curvature(to_point(*a))); });
}
int h2e_op_ecc_reduce_with_curvature(h2e_records* R, const h2e_point* a, h2e_point_c* out, void* stream) {   // :677-693 (ecc_reduce :668-675, assign_identity :514-529)
    if (!a || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("ecc_reduce_with_curvature", {{a, sizeof(*a)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) { *out = from_pc(e.ecc_reduce_with_curvature(to_point(*a))); });
}
int h2e_op_ecc_double(h2e_records* R, const h2e_point_c* a, h2e_point* out, void* stream) {   // :630-642
    if (!a || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("ecc_double", {{a, sizeof(*a)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) { *out = from_point(e.ecc_double(to_pc(*a))); });
}
int h2e_op_ecc_add(h2e_records* R, const h2e_point_c* a, const h2e_point* b, h2e_point* out, void* stream) {   // :606-628
    if (!a || !b || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("ecc_add", {{a, sizeof(*a)}, {b, sizeof(*b)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) { *out = from_point(e.ecc_add(to_pc(*a), to_point(*b))); });
}
int h2e_op_ecc_neg(h2e_records* R, const h2e_point* a, h2e_point* out, void* stream) {   // :660-666
    if (!a || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("ecc_neg", {{a, sizeof(*a)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) { *out = from_point(e.ecc_neg(to_point(*a))); });
}
int h2e_op_ecc_encode(h2e_records* R, const h2e_point* a, uint32_t* out_cells3, void* stream) {   // :710-732
    if (!a || !out_cells3) return fail(H2E_ERR_INVALID, "null operand");
    if (R && field_pair(R->field_pair).limbs != 3) return fail(H2E_ERR_INVALID, "ecc_encode packs two 3-limb coordinates");
    return records_op(R, key_of("ecc_encode", {{a, sizeof(*a)}}), 0, nullptr, stream, {OpOut{out_cells3, 3 * sizeof(uint32_t)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) {
        std::vector<h2e::AssignedValue> v = e.ecc_encode(to_point(*a));
        for (int i = 0; i < 3; i++) out_cells3[i] = v[i].ref;
    });
}
int h2e_op_ecc_mul(h2e_records* R, const h2e_point* a, const h2e_int* scalar, const void* d_inputs, h2e_point* out, void* stream) {   // :418-420
    return h2e_op_msm_unsafe(R, 1, a, scalar, d_inputs, out, stream);
}
int h2e_op_assign_constant_point(h2e_records* R, const uint64_t* x_words, const uint64_t* y_words, int is_identity, h2e_point* out, void* stream) {   // :441-456
    if (!out || (!is_identity && (!x_words || !y_words))) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("assign_constant_point", {{x_words, x_words ? (size_t)field_pair(R ? R->field_pair : 0).w_words * 8 : 0}, {y_words, y_words ? (size_t)field_pair(R ? R->field_pair : 0).w_words * 8 : 0}, {&is_identity, sizeof(is_identity)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext& e, uint32_t) {
        h2e::HBig x, y;
        if (!is_identity) {
            x = h2e::HBig::from_words(x_words, r.fp.w_words);
            y = h2e::HBig::from_words(y_words, r.fp.w_words);
        }
        *out = from_point(e.assign_constant_point(x, y, is_identity != 0));
    });
}
int h2e_op_bisec_point_with_curvature(h2e_records* R, uint32_t cond_cell, const h2e_point_c* a, const h2e_point_c* b, h2e_point_c* out, void* stream) {   // :562-578
    if (!a || !b || !out) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("bisec_point_with_curvature", {{&cond_cell, sizeof(cond_cell)}, {a, sizeof(*a)}, {b, sizeof(*b)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) {
        *out = from_pc(e.bisec_point_with_curvature(h2e::AssignedCondition{h2e::AssignedValue{cond_cell}}, to_pc(*a), to_pc(*b)));
    });
}
int h2e_op_assign_cache_point(h2e_records* R, const h2e_point_c* p, uint64_t group, uint64_t selector, void* stream) {   // :779-788
    if (!p) return fail(H2E_ERR_INVALID, "null operand");
    return records_op(R, key_of("assign_cache_point", {{p, sizeof(*p)}, {&group, sizeof(group)}, {&selector, sizeof(selector)}}), 0, nullptr, stream, {}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) { e.assign_cache_point(to_pc(*p), (size_t)group, (size_t)selector); });
}
int h2e_op_assign_selected_point(h2e_records* R, uint32_t n, const h2e_point_c* candidates, uint32_t index_cell, uint64_t group, h2e_point_c* out,
                                 void* stream) {   // :790-812, the candidate picked on the device by the value of the index cell
    if (!candidates || !out || n == 0) return fail(H2E_ERR_INVALID, "bad argument");
    return records_op(R, key_of("assign_selected_point", {{&n, sizeof(n)}, {candidates, sizeof(*candidates) * (size_t)n}, {&index_cell, sizeof(index_cell)}, {&group, sizeof(group)}}), 0, nullptr, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder&, h2e::NativeScalarEccContext& e, uint32_t) {
        std::vector<h2e::AssignedPointWithCurvature> c;
        for (uint32_t k = 0; k < n; k++) c.push_back(to_pc(candidates[k]));
        *out = from_pc(e.assign_selected_point(c, h2e::AssignedValue{index_cell}, (size_t)group));
    });
}
int h2e_op_assign_g2_constant(h2e_records* R, const void* d_inputs, h2e_g2* out, void* stream) {   // fq2_assign_constant x 2 + assign_constant(0)
    if (!out) return fail(H2E_ERR_INVALID, "out is null");
    return records_op(R, key_of("assign_g2_constant", {}), 4, d_inputs, stream, {OpOut{out, sizeof(*out)}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t s) {
        out->x0 = from_int(r.assign_int_constant_input(s + 0));
        out->x1 = from_int(r.assign_int_constant_input(s + 1));
        out->y0 = from_int(r.assign_int_constant_input(s + 2));
        out->y1 = from_int(r.assign_int_constant_input(s + 3));
        out->z = r.assign_constant_u64(0).ref;
    });
}
int h2e_op_check_pairing(h2e_records* R, uint32_t n_pairs, const h2e_point* g1, const h2e_g2* g2, void* stream) {   // pairing_chip.rs:173-176
    if (!g1 || !g2 || n_pairs == 0) return fail(H2E_ERR_INVALID, "bad argument");
    if (R && R->field_pair == H2E_FIELD_BLS12_381_FR) return fail(H2E_ERR_INVALID, "no pairing over this field");
    return records_op(R, key_of("check_pairing", {{&n_pairs, sizeof(n_pairs)}, {g1, sizeof(*g1) * (size_t)n_pairs}, {g2, sizeof(*g2) * (size_t)n_pairs}}), 0, nullptr, stream, {}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t) {
        r.auto_cut_every = pairing_cut_every(r.fp.id == H2E_FIELD_BN256_FQ ? 0 : 1);
        std::unique_ptr<h2e::PairingOps> po;
        if (r.fp.id == H2E_FIELD_BN256_FQ) po.reset(new h2e::Bn256PairingOps(r));
        else po.reset(new h2e::Bls12381PairingOps(r));
        std::vector<h2e::AssignedPoint> a;
        std::vector<h2e::AssignedG2Affine> b;
        for (uint32_t k = 0; k < n_pairs; k++) {
            a.push_back(to_point(g1[k]));
            b.push_back(h2e::AssignedG2Affine{h2e::AssignedFq2{to_int(g2[k].x0), to_int(g2[k].x1)}, h2e::AssignedFq2{to_int(g2[k].y0), to_int(g2[k].y1)},
                                              h2e::AssignedCondition{h2e::AssignedValue{g2[k].z}}});
        }
        std::vector<h2e::PairingOps::Term> terms;
        for (uint32_t k = 0; k < n_pairs; k++) terms.push_back(h2e::PairingOps::Term(&a[k], &b[k]));
        po->check_pairing(terms);
    });
}
int h2e_op_pairing(h2e_records* R, uint32_t n_pairs, const h2e_point* g1, const h2e_g2* g2, h2e_int* out12, void* stream) {   // pairing_chip.rs:157-171
    if (!g1 || !g2 || !out12 || n_pairs == 0) return fail(H2E_ERR_INVALID, "bad argument");
    if (R && R->field_pair == H2E_FIELD_BLS12_381_FR) return fail(H2E_ERR_INVALID, "no pairing over this field");
    return records_op(R, key_of("pairing", {{&n_pairs, sizeof(n_pairs)}, {g1, sizeof(*g1) * (size_t)n_pairs}, {g2, sizeof(*g2) * (size_t)n_pairs}}), 0, nullptr, stream, {OpOut{out12, sizeof(*out12) * 12}}, [&](h2e::Recorder& r, h2e::NativeScalarEccContext&, uint32_t) {
        r.auto_cut_every = pairing_cut_every(r.fp.id == H2E_FIELD_BN256_FQ ? 0 : 1);
        std::unique_ptr<h2e::PairingOps> po = tower_of(r);
        std::vector<h2e::AssignedPoint> a;
        std::vector<h2e::AssignedG2Affine> b;
        for (uint32_t k = 0; k < n_pairs; k++) {
            a.push_back(to_point(g1[k]));
            b.push_back(h2e::AssignedG2Affine{h2e::AssignedFq2{to_int(g2[k].x0), to_int(g2[k].x1)}, h2e::AssignedFq2{to_int(g2[k].y0), to_int(g2[k].y1)},
                                              h2e::AssignedCondition{h2e::AssignedValue{g2[k].z}}});
        }
        std::vector<h2e::PairingOps::Term> terms;
        for (uint32_t k = 0; k < n_pairs; k++) terms.push_back(h2e::PairingOps::Term(&a[k], &b[k]));
        from_fq12(po->pairing(terms), out12);
    });
}

}  // extern "C"
