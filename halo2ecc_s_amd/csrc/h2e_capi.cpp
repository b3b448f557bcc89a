// C ABI of the witness engine (include/h2e.h): program recording (host) + execution (HIP).  ONE translation unit, in parts:
//   capi_common.hpp       includes, engine entry points, error reporting, debug hooks
//   program.hpp           struct h2e_program (+ program_value_chain.hpp, program_replay.hpp, program_schedule.hpp: the compiler passes)
//   run_state.hpp         job slots, struct h2e_ctx
//   run.hpp               order tables, device copy of a program, run_impl, h2e_run / h2e_submit / h2e_wait
//   (this file)           contexts, the whole-program constructors, options / statistics, hand-off, check, named entry points
//   records_api.hpp       the operator API on a device-resident Context
#include "capi_common.hpp"
#include "program.hpp"
#include "run_state.hpp"

extern "C" {

const char* h2e_last_error(void) { return g_last_error.c_str(); }
extern "C" int h2e_engine_digit_rows_selftest_fp0(int, uint32_t, uint32_t, const void*, void*, hipStream_t);
extern "C" int h2e_engine_digit_rows_selftest_fp1(int, uint32_t, uint32_t, const void*, void*, hipStream_t);
int h2e_selftest_digit_rows(int field_pair, uint32_t op, uint32_t n_cases, const void* d_in, void* d_out, void* stream) {
    if (!d_in || !d_out) return fail(H2E_ERR_INVALID, "null argument");
    int rc = field_pair == 0   ? h2e_engine_digit_rows_selftest_fp0(field_pair, op, n_cases, d_in, d_out, (hipStream_t)stream)
             : field_pair == 1 ? h2e_engine_digit_rows_selftest_fp1(field_pair, op, n_cases, d_in, d_out, (hipStream_t)stream)
                               : -1;
    return rc == 0 ? 0 : fail(rc < 0 ? H2E_ERR_INVALID : H2E_ERR_HIP, "digit-row self-test: no such field pair / launch failed");
}
const char* h2e_last_warning(void) { return g_last_warning.c_str(); }
// hardware queues this process's HIP runtime maps its streams onto: GPU_MAX_HW_QUEUES as the runtime read it when it initialised
// (the environment is the only place it can be set; default 4)
static int64_t process_hw_queues() {
    const char* e = getenv("GPU_MAX_HW_QUEUES");
    int64_t v = e ? atoll(e) : 0;
    return v > 0 ? v : 4;
}
const char* h2e_version(void) {
#ifdef H2E_DEBUG_HOOKS
    return "h2e 0.3 (gfx950, batch-interleaved advice) [debug hooks]";   // exp/build_dbg.sh: launch log, stamp kernels - never the shipped library
#else
    return "h2e 0.3 (gfx950, batch-interleaved advice)";
#endif
}

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
    // (round 5, with the shorter chains: bn256 16 / 24 / 32 ops: 3.17 / 3.51 / 3.52 ms per pipelined 64-check step, bls12_381 8 / 12 / 16 / 24:
    // 1.53 / 1.61 / 1.63 / 1.92 - the defaults stay; 48 fails at launch, so the knob stops at 32)
    if (const char* e = getenv("H2E_PAIRING_CUT")) return (uint32_t)std::max(2, std::min(32, atoi(e)));
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

#include "run.hpp"
#include "ring.hpp"

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
            g_last_warning.clear();
            if (value > 1 && process_hw_queues() < value + 12) {
                // a process-global knob the library cannot set for itself (the runtime reads it once, before the first HIP call): a
                // pipelined host that gets it wrong loses silently (8 x bn256 checks: 1.8 instead of 0.7 ms per step) - so say it
                g_last_warning = "pipeline depth " + std::to_string(value) + " wants GPU_MAX_HW_QUEUES >= " + std::to_string(value + 12) +
                                 " in the environment before HIP initialises; this process runs with " + std::to_string(process_hw_queues()) +
                                 ": streams that share a hardware queue serialise (INTEGRATION.md, threading and streams)";
                static std::atomic<bool> said{false};
                if (!said.exchange(true)) fprintf(stderr, "libh2e: warning: %s\n", g_last_warning.c_str());
            }
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
        case H2E_STAT_HW_QUEUES: return process_hw_queues();
        case H2E_STAT_HW_QUEUES_WANTED: return ctx->depth > 1 ? (int64_t)ctx->depth + 12 : 1;
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
        for (uint32_t i = 0; i <= 2 * limbs; i++) {
            // the kernel reads refs as ABSOLUTE rows of the base array: anything else would be read out of place (or out of bounds)
            if (H2E_REF_REGION(refs[i]) != 0) return fail(H2E_ERR_INVALID, "h2e_unit_records: a result cell outside the base array");
            if (H2E_REF_REL(refs[i])) return fail(H2E_ERR_INVALID, "h2e_unit_records: a strand-relative result cell (the outputs of a program are absolute references)");
            if (H2E_REF_ROW(refs[i]) >= p->base_rows) return fail(H2E_ERR_INVALID, "h2e_unit_records: a result cell beyond the program's base rows");
        }
    }
    if (p->device >= 0 && p->device != ctx->device) return fail(H2E_ERR_INVALID, "program bound to another device");
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

#include "records_api.hpp"
}  // extern "C"

