// ORACLE — TEST INFRASTRUCTURE ONLY. Mirrors the second blocks of
// src/tests/native_scalar_pairing_chip.rs:67-97 and src/tests/general_scalar_pairing_chip.rs:74-105.
#include <cstdio>
#include <chrono>
#include <array>
#include "testutil.hpp"
#include "pairing.hpp"
#include "checker.hpp"
using namespace h2o;

template <class F2>
static Fq2Const fq2c(const F2& v) {
    return Fq2Const{v.c0.to_bn(), v.c1.to_bn()};
}

static size_t count_cells(Context& ctx) {
    size_t cells = 0;
    RecordsInner& in = *ctx.records.inner;
    for (auto& c : in.base_adv) cells += c.present;
    for (auto& c : in.range_adv) cells += c.present;
    for (auto& c : in.select_adv) cells += c.present;
    return cells;
}

static int run_bn() {
    SplitMix64 rng(0x68326563632d73ull + 4);
    BigUint sa = rng.below(Fr::modulus()), sb = rng.below(Fr::modulus());
    BnG1 a = JacT<BnFq>::from_affine(bn_g1_generator()).mul(sa).to_affine();
    BnG2 g2 = bn_g2_generator();
    if (!on_curve(g2, bn_g2_b())) {
        printf("bn256 G2 generator not on curve\n");
        return 1;
    }
    if (!JacT<BnFq2>::from_affine(g2).mul(Fr::modulus()).is_identity()) {
        printf("bn256 G2 generator not of order r\n");
        return 1;
    }
    BnG2 b = JacT<BnFq2>::from_affine(g2).mul(sb).to_affine();

    auto t0 = std::chrono::steady_clock::now();
    auto ctx = std::make_shared<Context>();
    IntegerContext ic(ctx, BnFq::modulus());
    NativeScalarEccContext ecc(ic, bn256_g1_params(), 0);
    Bn256PairingOps po(ecc.base);
    AssignedFq2 bx = po.fq2_assign_constant(fq2c(b.x));
    AssignedFq2 by = po.fq2_assign_constant(fq2c(b.y));
    AssignedG2Affine B{bx, by, AssignedCondition(ctx->assign_constant(Fr::zero()))};
    AssignedPoint neg_a = ecc.assign_point(to_native(a.neg()));
    AssignedPoint A = ecc.assign_point(to_native(a));
    size_t b0 = ctx->base_offset, r0 = ctx->range_offset;
    po.check_pairing({PairingOps::Term(&A, &B), PairingOps::Term(&neg_a, &B)});
    auto t1 = std::chrono::steady_clock::now();
    double secs = std::chrono::duration<double>(t1 - t0).count();
    size_t cells = count_cells(*ctx);
    printf("bn256 check_pairing: rows base %zu range %zu (total offsets %zu %zu) cells %zu  %.2f s (%.0f cells/s)\n",
           ctx->base_offset - b0, ctx->range_offset - r0, ctx->base_offset, ctx->range_offset, cells, secs, cells / secs);
    CheckReport rep = check_records(ctx->records);
    printf("  check: %s %s\n", rep.ok() ? "OK" : "FAIL", rep.first_error.c_str());
    return rep.ok() ? 0 : 1;
}

static int run_bls() {
    SplitMix64 rng(0x68326563632d73ull + 5);
    BlsG1 g1 = bls_g1_generator();
    BlsG2 g2 = bls_g2_generator();
    if (!on_curve(g1, bls_g1_b()) || !on_curve(g2, bls_g2_b())) {
        printf("bls12_381 generator not on curve\n");
        return 1;
    }
    if (!JacT<BlsFq>::from_affine(g1).mul(BlsFr::modulus()).is_identity() ||
        !JacT<BlsFq2>::from_affine(g2).mul(BlsFr::modulus()).is_identity()) {
        printf("bls12_381 generator not of order r\n");
        return 1;
    }
    BigUint sa = rng.below(BlsFr::modulus()), sb = rng.below(BlsFr::modulus()), c = rng.below(BlsFr::modulus());
    JacT<BlsFq> aj = JacT<BlsFq>::from_affine(g1).mul(sa);
    BlsG1 a = aj.to_affine();
    BlsG1 ac = aj.mul(c).to_affine();
    JacT<BlsFq2> bj = JacT<BlsFq2>::from_affine(g2).mul(sb);
    BlsG2 b = bj.to_affine();
    BlsG2 bc = bj.mul(c).to_affine();

    auto t0 = std::chrono::steady_clock::now();
    auto ctx = std::make_shared<Context>();
    // GeneralScalarEccContext::new builds both integer contexts (context.rs:230-239); only the base one is used
    IntegerContext ic(ctx, BlsFq::modulus());
    IntegerContext sc(ctx, BlsFr::modulus());
    (void)sc;
    NativeScalarEccContext ecc(ic, bls12_381_g1_params(), 0);  // EccChipBaseOps only (assign_point)
    Bls12381PairingOps po(ecc.base);
    AssignedFq2 bx = po.fq2_assign_constant(fq2c(b.x));
    AssignedFq2 by = po.fq2_assign_constant(fq2c(b.y));
    AssignedG2Affine B{bx, by, AssignedCondition(ctx->assign_constant(Fr::zero()))};
    AssignedFq2 bcx = po.fq2_assign_constant(fq2c(bc.x));
    AssignedFq2 bcy = po.fq2_assign_constant(fq2c(bc.y));
    AssignedG2Affine BC{bcx, bcy, AssignedCondition(ctx->assign_constant(Fr::zero()))};
    AssignedPoint neg_a = ecc.assign_point(to_native(a.neg()));
    AssignedPoint AC = ecc.assign_point(to_native(ac));
    size_t b0 = ctx->base_offset, r0 = ctx->range_offset;
    po.check_pairing({PairingOps::Term(&AC, &B), PairingOps::Term(&neg_a, &BC)});
    auto t1 = std::chrono::steady_clock::now();
    double secs = std::chrono::duration<double>(t1 - t0).count();
    size_t cells = count_cells(*ctx);
    printf("bls12_381 check_pairing: rows base %zu range %zu (total offsets %zu %zu) cells %zu  %.2f s (%.0f cells/s)\n",
           ctx->base_offset - b0, ctx->range_offset - r0, ctx->base_offset, ctx->range_offset, cells, secs, cells / secs);
    CheckReport rep = check_records(ctx->records);
    printf("  check: %s %s\n", rep.ok() ? "OK" : "FAIL", rep.first_error.c_str());
    return rep.ok() ? 0 : 1;
}

int main() {
    init_fields();
    int rc = 0;
    try {
        rc |= run_bn();
        rc |= run_bls();
    } catch (std::exception& e) {
        printf("EXCEPTION: %s\n", e.what());
        return 2;
    }
    return rc;
}
