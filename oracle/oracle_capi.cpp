// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Not part of the product path.
//
// C entry points used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg (ctypes).
// Each run_* builds the same workload the engine's h2e_program_* describes, from the same input
// vector, with the CPU restatement, and keeps the resulting Records for export / checking.
#include <chrono>
#include <cstring>
#include <memory>
#include "checker.hpp"
#include "pairing.hpp"
#include "testutil.hpp"

using namespace h2o;

namespace {
struct Run {
    std::shared_ptr<Context> ctx;
    int status = 0;  // 0 ok, 1 panic, 2 UnsafeError
    std::string error;
    double seconds = 0;
};
const BigUint& wmod(int fp) { return fp == 0 ? BnFq::modulus() : fp == 1 ? BlsFq::modulus() : BlsFr::modulus(); }
int slot_words(int fp) { return fp == 1 ? 6 : 4; }
struct Inputs {
    const uint64_t* p;
    int sw;
    BigUint w(uint32_t slot) const { return BigUint::from_limbs(p + (size_t)slot * sw, sw); }
    Fr fr(uint32_t slot) const { return Fr::from_bn(BigUint::from_limbs(p + (size_t)slot * sw, 4)); }
    NativePoint point(uint32_t xs, uint32_t ys, uint32_t zs) const {
        NativePoint n;
        n.x = w(xs);
        n.y = w(ys);
        n.is_identity = !w(zs).is_zero();
        return n;
    }
};
template <class F>
Run* guarded(F f) {
    init_fields();
    Run* r = new Run();
    r->ctx = std::make_shared<Context>();
    auto t0 = std::chrono::steady_clock::now();
    try {
        f(*r);
    } catch (UnsafeError& e) {
        r->status = 2;
        r->error = "UnsafeError";
    } catch (std::exception& e) {
        r->status = 1;
        r->error = e.what();
    }
    r->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return r;
}
uint32_t enc(const Cell& c) { return ((uint32_t)c.region << 30) | ((uint32_t)c.col << 27) | c.row; }
}  // namespace

extern "C" {

void* oracle_run_int_mul_batch(int fp, uint32_t n, const uint64_t* inputs) {
    return guarded([&](Run& r) {
        IntegerContext ic(r.ctx, wmod(fp));
        Inputs in{inputs, slot_words(fp)};
        for (uint32_t k = 0; k < n; k++) {
            AssignedInteger a = ic.assign_w(in.w(2 * k));
            AssignedInteger b = ic.assign_w(in.w(2 * k + 1));
            ic.int_mul(a, b);
        }
    });
}

// src/tests/integer_chip.rs:11-55
void* oracle_run_integer_chip_st(int fp, const uint64_t* inputs) {
    return guarded([&](Run& r) {
        IntegerContext ic(r.ctx, wmod(fp));
        Inputs in{inputs, slot_words(fp)};
        AssignedInteger a = ic.assign_w(in.w(0)), b = ic.assign_w(in.w(1));
        AssignedInteger c1 = ic.assign_w(in.w(2));
        AssignedInteger c2 = ic.int_add(a, b);
        ic.assert_int_equal(c1, c2);
        AssignedInteger d1 = ic.assign_w(in.w(3));
        AssignedInteger d2 = ic.int_sub(a, b);
        ic.assert_int_equal(d1, d2);
        AssignedInteger e1 = ic.assign_w(in.w(4));
        AssignedInteger e2 = ic.int_mul(a, b);
        ic.assert_int_equal(e1, e2);
        AssignedInteger f1 = ic.assign_w(in.w(5));
        AssignedInteger f2 = ic.int_div(a, b).second;
        ic.assert_int_equal(f1, f2);
        AssignedInteger zero = ic.int_sub(a, a);
        auto g = ic.int_div(a, zero);
        r.ctx->assert_true(g.first);
    });
}

// src/tests/native_scalar_ecc_chip.rs:34-47 for one tile
static void* run_msm_bn256_tile(uint32_t n, const uint64_t* inputs, int threads, bool with_select);
void* oracle_run_msm_bn256_tile(uint32_t n, const uint64_t* inputs, int threads) { return run_msm_bn256_tile(n, inputs, threads, true); }
// the same body on NativeScalarEccContext::new_without_select_chip (src/context.rs:190-207; ecc_chip.rs:91-221)
void* oracle_run_msm_bn256_tile_no_select(uint32_t n, const uint64_t* inputs, int threads) {
    return run_msm_bn256_tile(n, inputs, threads, false);
}
static void* run_msm_bn256_tile(uint32_t n, const uint64_t* inputs, int threads, bool with_select) {
    return guarded([&](Run& r) {
        IntegerContext ic(r.ctx, BnFq::modulus());
        Inputs in{inputs, 4};
        NativeScalarEccContext ecc = with_select ? NativeScalarEccContext::new_with_select_chip(ic, bn256_g1_params())
                                                 : NativeScalarEccContext::new_without_select_chip(ic, bn256_g1_params());
        ecc.n_threads = threads;
        // the generator comes in as an input too (slots 4n, 4n+1); it must be C::generator()
        ecc.curve.generator = in.point(4 * n, 4 * n + 1, 4 * n + 8);
        ecc.curve.generator.is_identity = false;
        std::vector<AssignedPoint> ap;
        std::vector<AssignedValue> as;
        for (uint32_t k = 0; k < n; k++) ap.push_back(ecc.assign_point(in.point(3 * k, 3 * k + 1, 3 * k + 2)));
        for (uint32_t k = 0; k < n; k++) as.push_back(r.ctx->assign(in.fr(3 * n + k)));
        NativePoint r1 = in.point(4 * n + 2, 4 * n + 3, 4 * n + 8), r2 = in.point(4 * n + 4, 4 * n + 5, 4 * n + 8);
        r1.is_identity = r2.is_identity = false;
        AssignedPoint res = ecc.msm_unsafe(ap, as, r1, r2);
        AssignedPoint res_expect = ecc.assign_point(in.point(4 * n + 6, 4 * n + 7, 4 * n + 8));
        ecc.ecc_assert_equal(res, res_expect);
    });
}

// second block of src/tests/native_scalar_pairing_chip.rs:67-97
void* oracle_run_pairing_check_bn256(const uint64_t* inputs) {
    return guarded([&](Run& r) {
        IntegerContext ic(r.ctx, BnFq::modulus());
        Inputs in{inputs, 4};
        NativeScalarEccContext ecc(ic, bn256_g1_params(), 0);
        Bn256PairingOps po(ecc.base);
        AssignedFq2 bx = po.fq2_assign_constant(Fq2Const{in.w(0), in.w(1)});
        AssignedFq2 by = po.fq2_assign_constant(Fq2Const{in.w(2), in.w(3)});
        AssignedG2Affine B{bx, by, AssignedCondition(r.ctx->assign_constant(Fr::zero()))};
        AssignedPoint neg_a = ecc.assign_point(in.point(4, 5, 6));
        AssignedPoint a = ecc.assign_point(in.point(7, 8, 9));
        po.check_pairing({PairingOps::Term(&a, &B), PairingOps::Term(&neg_a, &B)});
    });
}

// Operator-API scenario (tests/test_ops_gpu.py): chip ops called one after the other on ONE context that already holds rows -
// assign_point x n, assign x n, five integer ops, msm_unsafe twice (second call: msm prefix 2^20, quirk Q9; swapped blinding
// points), ecc_assert_equal(res1, res2).  Inputs as for the MSM tile (the expected-result slots are unused).
void* oracle_run_ops_msm_twice(uint32_t n, const uint64_t* inputs) {
    return guarded([&](Run& r) {
        IntegerContext ic(r.ctx, BnFq::modulus());
        Inputs in{inputs, 4};
        NativeScalarEccContext ecc = NativeScalarEccContext::new_with_select_chip(ic, bn256_g1_params());
        ecc.curve.generator = in.point(4 * n, 4 * n + 1, 4 * n + 8);
        ecc.curve.generator.is_identity = false;
        std::vector<AssignedPoint> ap;
        std::vector<AssignedValue> as;
        for (uint32_t k = 0; k < n; k++) ap.push_back(ecc.assign_point(in.point(3 * k, 3 * k + 1, 3 * k + 2)));
        for (uint32_t k = 0; k < n; k++) as.push_back(r.ctx->assign(in.fr(3 * n + k)));
        AssignedInteger m = ic.int_mul(ap[0].x, ap[0].y);
        AssignedInteger s = ic.int_add(m, m);
        AssignedInteger s2 = ic.int_sub(s, ap[0].x);
        AssignedInteger rd = ic.reduce(s2);
        ic.int_div(rd, ap[0].y);
        NativePoint r1 = in.point(4 * n + 2, 4 * n + 3, 4 * n + 8), r2 = in.point(4 * n + 4, 4 * n + 5, 4 * n + 8);
        r1.is_identity = r2.is_identity = false;
        AssignedPoint res1 = ecc.msm_unsafe(ap, as, r1, r2);
        AssignedPoint res2 = ecc.msm_unsafe(ap, as, r2, r1);
        ecc.ecc_assert_equal(res1, res2);
    });
}

// Operator-API scenario over the complete-addition / curvature surface of EccChipBaseOps (tests/test_ops_gpu.py).
// inputs (4-word slots): P (x, y, z), Q (x, y, z), scalar, index (= 1), generator (x, y), r1 (x, y), r2 (x, y)
void* oracle_run_ops_ecc_surface(const uint64_t* inputs) {
    return guarded([&](Run& r) {
        IntegerContext ic(r.ctx, BnFq::modulus());
        Inputs in{inputs, 4};
        NativeScalarEccContext ecc = NativeScalarEccContext::new_with_select_chip(ic, bn256_g1_params());
        ecc.curve.generator = in.point(8, 9, 2);
        ecc.curve.generator.is_identity = false;
        AssignedPoint P = ecc.assign_point(in.point(0, 1, 2));
        AssignedPoint Q = ecc.assign_point(in.point(3, 4, 5));
        AssignedValue s = r.ctx->assign(in.fr(6));
        AssignedValue idx = r.ctx->assign(in.fr(7));
        AssignedPointWithCurvature Pc = ecc.ecc_reduce_with_curvature(P);
        AssignedPoint D = ecc.ecc_double(Pc);
        AssignedPointWithCurvature Qc = ecc.to_point_with_curvature(Q);
        AssignedPoint S = ecc.ecc_add(Qc, D);
        AssignedPoint N = ecc.ecc_neg(S);
        ecc.ecc_encode(N);
        NativePoint r1 = in.point(10, 11, 2), r2 = in.point(12, 13, 2);
        r1.is_identity = r2.is_identity = false;
        ecc.msm_unsafe({P}, {s}, r1, r2);                                  // ecc_mul (ecc_chip.rs:418-420)
        AssignedPoint C = ecc.assign_constant_point(ecc.curve.generator);
        AssignedPointWithCurvature Cc = ecc.to_point_with_curvature(C);
        ecc.bisec_point_with_curvature(P.z, Pc, Cc);
        ecc.assign_cache_point(Pc, 7, 0);
        ecc.assign_cache_point(Cc, 7, 1);
        std::vector<AssignedPointWithCurvature> cands{Pc, Cc};
        uint64_t w[4];
        idx.val.to_canonical(w);
        AssignedPointWithCurvature Sel = ecc.assign_selected_point(cands[w[0] & 0xff], idx, 7);
        ecc.ecc_assert_equal(Sel.to_point(), C);
    });
}

// src/tests/general_scalar_ecc_chip.rs:14-49 for one tile of n points (GeneralScalarEccContext<bls12_381::G1Affine, bn256::Fr>);
// same input layout as h2e_program_msm_bls12_381_tile (6-word slots)
void* oracle_run_msm_bls12_381_tile(uint32_t n, const uint64_t* inputs) {
    return guarded([&](Run& r) {
        IntegerContext ic(r.ctx, BlsFq::modulus());
        IntegerContext sc(r.ctx, BlsFr::modulus());
        Inputs in{inputs, 6};
        GeneralScalarEccContext ecc(ic, sc, bls12_381_g1_params(), 0);
        ecc.curve.generator = in.point(4 * n, 4 * n + 1, 4 * n + 8);
        ecc.curve.generator.is_identity = false;
        std::vector<AssignedPoint> ap;
        std::vector<AssignedInteger> as;
        for (uint32_t k = 0; k < n; k++) ap.push_back(ecc.assign_point(in.point(3 * k, 3 * k + 1, 3 * k + 2)));
        for (uint32_t k = 0; k < n; k++) as.push_back(ecc.scalar.assign_w(in.w(3 * n + k)));
        NativePoint r1 = in.point(4 * n + 2, 4 * n + 3, 4 * n + 8), r2 = in.point(4 * n + 4, 4 * n + 5, 4 * n + 8);
        r1.is_identity = r2.is_identity = false;
        AssignedPoint res = ecc.msm_unsafe(ap, as, r1, r2);
        AssignedPoint res_expect = ecc.assign_point(in.point(4 * n + 6, 4 * n + 7, 4 * n + 8));
        ecc.ecc_assert_equal(res, res_expect);
    });
}

// first block of the pairing tests: pairing(terms) [== expected]
// (src/tests/native_scalar_pairing_chip.rs:20-65, general_scalar_pairing_chip.rs:20-72); same inputs as h2e_program_pairing
void* oracle_run_pairing(int curve, uint32_t n_pairs, int with_expected, const uint64_t* inputs) {
    return guarded([&](Run& r) {
        IntegerContext ic(r.ctx, curve == 0 ? BnFq::modulus() : BlsFq::modulus());
        Inputs in{inputs, curve == 0 ? 4 : 6};
        NativeScalarEccContext ecc(ic, curve == 0 ? bn256_g1_params() : bls12_381_g1_params(), 0);
        std::unique_ptr<PairingOps> po;
        if (curve == 0) po.reset(new Bn256PairingOps(ecc.base));
        else po.reset(new Bls12381PairingOps(ecc.base));
        std::vector<AssignedG2Affine> g2;
        for (uint32_t k = 0; k < n_pairs; k++) {
            AssignedFq2 x = po->fq2_assign_constant(Fq2Const{in.w(4 * k), in.w(4 * k + 1)});
            AssignedFq2 y = po->fq2_assign_constant(Fq2Const{in.w(4 * k + 2), in.w(4 * k + 3)});
            g2.push_back(AssignedG2Affine{x, y, AssignedCondition(r.ctx->assign_constant(Fr::zero()))});
        }
        uint32_t e0 = 4 * n_pairs;
        AssignedFq12 expected;
        if (with_expected) {
            AssignedFq2 v[6];
            for (int i = 0; i < 6; i++) v[i] = po->fq2_assign_constant(Fq2Const{in.w(e0 + 2 * i), in.w(e0 + 2 * i + 1)});
            expected = AssignedFq12{AssignedFq6{v[0], v[1], v[2]}, AssignedFq6{v[3], v[4], v[5]}};
        }
        uint32_t p0 = e0 + (with_expected ? 12 : 0);
        std::vector<AssignedPoint> g1;
        for (uint32_t k = 0; k < n_pairs; k++) g1.push_back(ecc.assign_point(in.point(p0 + 3 * k, p0 + 3 * k + 1, p0 + 3 * k + 2)));
        std::vector<PairingOps::Term> terms;
        for (uint32_t k = 0; k < n_pairs; k++) terms.push_back(PairingOps::Term(&g1[k], &g2[k]));
        AssignedFq12 res = po->pairing(terms);
        if (with_expected) po->fq12_assert_eq(expected, res);
    });
}

// Operator-API scenario over the rest of IntegerChipOps (src/circuit/integer_chip.rs:15-70: int_neg, int_square,
// int_unsafe_invert, is_int_zero, is_int_equal, assign_int_constant, int_mul_small_constant, bisec_int, assert_int_equal) and
// the Fq2 / Fq6 / Fq12 surface (src/circuit/fq12.rs:24-459) on assigned elements (tests/test_ops_gpu.py).
// inputs: a, b, x_0 .. x_11, y_0 .. y_11 (W values; x, y = two Fq12 elements in fq12_assign_constant order)
void* oracle_run_ops_int_tower(int curve, const uint64_t* inputs) {
    return guarded([&](Run& r) {
        IntegerContext ic(r.ctx, curve == 0 ? BnFq::modulus() : BlsFq::modulus());
        Inputs in{inputs, curve == 0 ? 4 : 6};
        NativeScalarEccContext ecc(ic, curve == 0 ? bn256_g1_params() : bls12_381_g1_params(), 0);
        std::unique_ptr<PairingOps> po;
        if (curve == 0) po.reset(new Bn256PairingOps(ecc.base));
        else po.reset(new Bls12381PairingOps(ecc.base));
        IntegerContext& c = ecc.base;
        AssignedInteger A = c.assign_w(in.w(0)), B = c.assign_w(in.w(1));
        std::vector<AssignedInteger> X, Y;
        for (int i = 0; i < 12; i++) X.push_back(c.assign_w(in.w(2 + i)));
        for (int i = 0; i < 12; i++) Y.push_back(c.assign_w(in.w(14 + i)));
        c.int_neg(A);
        c.int_square(A);
        AssignedInteger inv = c.int_unsafe_invert(B);
        AssignedCondition z = c.is_int_zero(A);
        c.is_int_equal(A, B);
        c.assign_int_constant(c.info->w_modulus - BigUint(5));
        c.int_mul_small_constant(A, 5);
        c.bisec_int(z, A, B);
        AssignedInteger t = c.int_mul(inv, B);
        AssignedInteger one = c.assign_int_constant(BigUint(1));
        c.assert_int_equal(t, one);
        auto f2 = [&](const std::vector<AssignedInteger>& v, int k) { return AssignedFq2{v[k], v[k + 1]}; };
        auto f6 = [&](const std::vector<AssignedInteger>& v, int k) { return AssignedFq6{f2(v, k), f2(v, k + 2), f2(v, k + 4)}; };
        AssignedFq2 x2 = f2(X, 0), y2 = f2(Y, 0);
        AssignedFq2 a2 = po->fq2_add(x2, y2);
        po->fq2_sub(x2, y2);
        AssignedFq2 m2 = po->fq2_mul(x2, y2);
        po->fq2_square(x2);
        po->fq2_neg(x2);
        po->fq2_double(x2);
        po->fq2_conjugate(x2);
        po->fq2_unsafe_invert(y2);
        po->fq2_mul_by_nonresidue(m2);
        po->fq2_frobenius_map(x2, 1);
        AssignedFq2 r2 = po->fq2_reduce(a2);
        po->fq2_assert_equal(r2, a2);
        AssignedFq6 x6 = f6(X, 0), y6 = f6(Y, 0);
        AssignedFq6 a6 = po->fq6_add(x6, y6);
        po->fq6_sub(x6, y6);
        AssignedFq6 m6 = po->fq6_mul(x6, y6);
        po->fq6_square(x6);
        po->fq6_neg(x6);
        po->fq6_double(x6);
        po->fq6_unsafe_invert(y6);
        po->fq6_mul_by_nonresidue(m6);
        po->fq6_frobenius_map(x6, 1);
        AssignedFq6 r6 = po->fq6_reduce(a6);
        po->fq6_assert_equal(r6, a6);
        AssignedFq12 x12{f6(X, 0), f6(X, 6)}, y12{f6(Y, 0), f6(Y, 6)};
        AssignedFq12 a12 = po->fq12_add(x12, y12);
        po->fq12_sub(x12, y12);
        po->fq12_mul(x12, y12);
        po->fq12_square(x12);
        po->fq12_neg(x12);
        po->fq12_double(x12);
        po->fq12_conjugate(x12);
        po->fq12_frobenius_map(x12, 1);
        po->fq12_cyclotomic_square(x12);
        po->fq12_unsafe_invert(y12);
        AssignedFq12 r12 = po->fq12_reduce(a12);
        po->fq12_assert_eq(r12, a12);
    });
}

// second block of src/tests/general_scalar_pairing_chip.rs:74-105
void* oracle_run_pairing_check_bls12_381(const uint64_t* inputs) {
    return guarded([&](Run& r) {
        IntegerContext ic(r.ctx, BlsFq::modulus());
        IntegerContext sc(r.ctx, BlsFr::modulus());  // GeneralScalarEccContext::new builds both (context.rs:230-239)
        (void)sc;
        Inputs in{inputs, 6};
        NativeScalarEccContext ecc(ic, bls12_381_g1_params(), 0);
        Bls12381PairingOps po(ecc.base);
        AssignedFq2 bx = po.fq2_assign_constant(Fq2Const{in.w(0), in.w(1)});
        AssignedFq2 by = po.fq2_assign_constant(Fq2Const{in.w(2), in.w(3)});
        AssignedG2Affine B{bx, by, AssignedCondition(r.ctx->assign_constant(Fr::zero()))};
        AssignedFq2 bcx = po.fq2_assign_constant(Fq2Const{in.w(4), in.w(5)});
        AssignedFq2 bcy = po.fq2_assign_constant(Fq2Const{in.w(6), in.w(7)});
        AssignedG2Affine BC{bcx, bcy, AssignedCondition(r.ctx->assign_constant(Fr::zero()))};
        AssignedPoint neg_a = ecc.assign_point(in.point(8, 9, 10));
        AssignedPoint ac = ecc.assign_point(in.point(11, 12, 13));
        po.check_pairing({PairingOps::Term(&ac, &B), PairingOps::Term(&neg_a, &BC)});
    });
}

struct OracleInfo {
    uint64_t base_offset, range_offset, select_offset;
    uint64_t base_height, range_height, select_height;
    uint64_t n_permutations, n_advice_cells;
    int32_t status;
    double seconds;
};
void oracle_info(void* h, OracleInfo* out) {
    Run* r = (Run*)h;
    Context& c = *r->ctx;
    out->base_offset = c.base_offset;
    out->range_offset = c.range_offset;
    out->select_offset = c.select_offset;
    out->base_height = c.records.base_height;
    out->range_height = c.records.range_height;
    out->select_height = c.records.select_height;
    out->n_permutations = c.records.permutations.size();
    size_t cells = 0;
    RecordsInner& in = *c.records.inner;
    for (auto& x : in.base_adv) cells += x.present;
    for (auto& x : in.range_adv) cells += x.present;
    for (auto& x : in.select_adv) cells += x.present;
    out->n_advice_cells = cells;
    out->status = r->status;
    out->seconds = r->seconds;
}
const char* oracle_error(void* h) { return ((Run*)h)->error.c_str(); }

// region 0/1/2; out[rows][cols][4] canonical, flags[rows][cols] bit0 assigned bit1 permute; rows beyond storage = 0
void oracle_export_adv(void* h, int region, uint64_t* out, uint8_t* flags, uint64_t rows) {
    Run* r = (Run*)h;
    RecordsInner& in = *r->ctx->records.inner;
    const std::vector<AdvCell>& v = region == 0 ? in.base_adv : region == 1 ? in.range_adv : in.select_adv;
    int cols = region == 0 ? 5 : region == 1 ? 3 : 2;
    size_t have = v.size() / cols;
    for (uint64_t row = 0; row < rows; row++)
        for (int c = 0; c < cols; c++) {
            uint64_t* o = out + (row * cols + c) * 4;
            if (row < have && v[row * cols + c].present) {
                v[row * cols + c].val.to_canonical(o);
            } else {
                o[0] = o[1] = o[2] = o[3] = 0;
            }
            flags[row * cols + c] = row < have ? (uint8_t)(v[row * cols + c].present | (v[row * cols + c].permute << 1)) : 0;
        }
}
// The streaming-job digest of one advice array (definition: include/h2e.h, h2e_digest): sum over assigned cells of
// sm(w_j ^ sm(row * COLS + col) ^ j * 0xA24BAED4963EE407) mod 2^64, sm = SplitMix64 finaliser, w = canonical words.
static uint64_t digest_sm64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void oracle_digest(void* h, int region, uint64_t* out4) {
    Run* r = (Run*)h;
    RecordsInner& in = *r->ctx->records.inner;
    const std::vector<AdvCell>& v = region == 0 ? in.base_adv : region == 1 ? in.range_adv : in.select_adv;
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    for (size_t cell = 0; cell < v.size(); cell++) {
        if (!v[cell].present) continue;
        uint64_t w[4];
        v[cell].val.to_canonical(w);
        uint64_t t = digest_sm64((uint64_t)cell);
        for (int j = 0; j < 4; j++) out4[j] += digest_sm64(w[j] ^ t ^ ((uint64_t)j * 0xA24BAED4963EE407ull));
    }
}
// the stream digest of include/h2e.h (h2e_run_digest): the position-keyed linear checksum the engine's expansion accumulates
// while it stores: per assigned cell, pos = row * COLS + col (32 bit), h = pos * 0x9E3779B1, k0 = (h ^ h >> 15) | 1,
// k1 = (h * 0x85EBCA77 + 0xC2B2AE3D) | 1; digest[j] += lo32(w_j) * k0 + hi32(w_j) * k1  (mod 2^64)
void oracle_stream_digest(void* h, int region, uint64_t* out4) {
    Run* r = (Run*)h;
    RecordsInner& in = *r->ctx->records.inner;
    const std::vector<AdvCell>& v = region == 0 ? in.base_adv : region == 1 ? in.range_adv : in.select_adv;
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    for (size_t cell = 0; cell < v.size(); cell++) {
        if (!v[cell].present) continue;
        uint64_t w[4];
        v[cell].val.to_canonical(w);
        uint32_t hh = (uint32_t)cell * 0x9E3779B1u;
        uint32_t k0 = (hh ^ (hh >> 15)) | 1u, k1 = (hh * 0x85EBCA77u + 0xC2B2AE3Du) | 1u;
        for (int j = 0; j < 4; j++) out4[j] += (uint64_t)(uint32_t)w[j] * k0 + (uint64_t)(uint32_t)(w[j] >> 32) * k1;
    }
}
void oracle_export_fix(void* h, int region, uint64_t* out, uint8_t* present, uint64_t rows) {
    Run* r = (Run*)h;
    RecordsInner& in = *r->ctx->records.inner;
    const std::vector<FixCell>& v = region == 0 ? in.base_fix : region == 1 ? in.range_fix : in.select_fix;
    int cols = region == 0 ? 9 : 2;
    size_t have = v.size() / cols;
    for (uint64_t row = 0; row < rows; row++)
        for (int c = 0; c < cols; c++) {
            uint64_t* o = out + (row * cols + c) * 4;
            bool p = row < have && v[row * cols + c].present;
            if (p)
                v[row * cols + c].val.to_canonical(o);
            else
                o[0] = o[1] = o[2] = o[3] = 0;
            present[row * cols + c] = p;
        }
}
void oracle_export_permutations(void* h, uint32_t* out) {
    Run* r = (Run*)h;
    size_t i = 0;
    for (auto& p : r->ctx->records.permutations) {
        out[i++] = enc(p.first);
        out[i++] = enc(p.second);
    }
}
// constraint checker: 0 = all satisfied
int oracle_check(void* h, char* msg, int cap) {
    Run* r = (Run*)h;
    CheckReport rep = check_records(r->ctx->records);
    if (msg && cap > 0) {
        std::strncpy(msg, rep.first_error.c_str(), cap - 1);
        msg[cap - 1] = 0;
    }
    return rep.ok() ? 0 : 1;
}
// the same, per class: failing rows of the base gate, the range gates, the range lookups, the select lookup, failing copy constraints
// (the order of include/h2e.h H2E_CHECK_*: the device-side check of the engine's arrays is held against these counts)
int oracle_check_counts(void* h, uint64_t* out5) {
    Run* r = (Run*)h;
    CheckReport rep = check_records(r->ctx->records);
    out5[0] = rep.base_gate_failures;
    out5[1] = rep.range_gate_failures;
    out5[2] = rep.range_lookup_failures;
    out5[3] = rep.select_lookup_failures;
    out5[4] = rep.permutation_failures;
    return rep.ok() ? 0 : 1;
}
// flip one advice cell (for checker self-tests)
void oracle_corrupt_adv(void* h, int region, uint64_t row, int col) {
    Run* r = (Run*)h;
    RecordsInner& in = *r->ctx->records.inner;
    std::vector<AdvCell>& v = region == 0 ? in.base_adv : region == 1 ? in.range_adv : in.select_adv;
    int cols = region == 0 ? 5 : region == 1 ? 3 : 2;
    v[row * cols + col].val = v[row * cols + col].val + Fr::one();
}
void oracle_free(void* h) { delete (Run*)h; }

// ---- native helpers for input generation (not chip code) -----------------------------------------
// k * P on bn256 G1 / bls12_381 G1 / G2 with canonical little-endian words; returns 1 if the result is identity
int oracle_bn256_g1_mul(const uint64_t* px, const uint64_t* py, const uint64_t* k, uint64_t* ox, uint64_t* oy) {
    init_fields();
    BnG1 p{BnFq::from_bn(BigUint::from_limbs(px, 4)), BnFq::from_bn(BigUint::from_limbs(py, 4)), false};
    BnG1 q = JacT<BnFq>::from_affine(p).mul(BigUint::from_limbs(k, 4)).to_affine();
    if (q.inf) return 1;
    q.x.to_canonical(ox);
    q.y.to_canonical(oy);
    return 0;
}
// sum_i k_i * P_i on bn256 G1
int oracle_bn256_g1_msm(uint32_t n, const uint64_t* pts /*[n][2][4]*/, const uint64_t* ks /*[n][4]*/, uint64_t* ox, uint64_t* oy) {
    init_fields();
    JacT<BnFq> acc = JacT<BnFq>::identity();
    for (uint32_t i = 0; i < n; i++) {
        BnG1 p{BnFq::from_bn(BigUint::from_limbs(pts + i * 8, 4)), BnFq::from_bn(BigUint::from_limbs(pts + i * 8 + 4, 4)), false};
        acc = acc.add(JacT<BnFq>::from_affine(p).mul(BigUint::from_limbs(ks + i * 4, 4)));
    }
    BnG1 q = acc.to_affine();
    if (q.inf) return 1;
    q.x.to_canonical(ox);
    q.y.to_canonical(oy);
    return 0;
}
int oracle_bn256_g2_mul_gen(const uint64_t* k, uint64_t* out /*x.c0,x.c1,y.c0,y.c1 each 4 words*/) {
    init_fields();
    BnG2 q = JacT<BnFq2>::from_affine(bn_g2_generator()).mul(BigUint::from_limbs(k, 4)).to_affine();
    if (q.inf) return 1;
    q.x.c0.to_canonical(out);
    q.x.c1.to_canonical(out + 4);
    q.y.c0.to_canonical(out + 8);
    q.y.c1.to_canonical(out + 12);
    return 0;
}
int oracle_bls12_381_g1_mul_gen(const uint64_t* k, uint64_t* out /*x,y each 6 words*/) {
    init_fields();
    BlsG1 q = JacT<BlsFq>::from_affine(bls_g1_generator()).mul(BigUint::from_limbs(k, 4)).to_affine();
    if (q.inf) return 1;
    q.x.to_canonical(out);
    q.y.to_canonical(out + 6);
    return 0;
}
int oracle_bls12_381_g2_mul_gen(const uint64_t* k, uint64_t* out /*x.c0,x.c1,y.c0,y.c1 each 6 words*/) {
    init_fields();
    BlsG2 q = JacT<BlsFq2>::from_affine(bls_g2_generator()).mul(BigUint::from_limbs(k, 4)).to_affine();
    if (q.inf) return 1;
    q.x.c0.to_canonical(out);
    q.x.c1.to_canonical(out + 6);
    q.y.c0.to_canonical(out + 12);
    q.y.c1.to_canonical(out + 18);
    return 0;
}

}  // extern "C"
