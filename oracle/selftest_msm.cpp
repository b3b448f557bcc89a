// ORACLE — TEST INFRASTRUCTURE ONLY. Mirrors src/tests/native_scalar_ecc_chip.rs:13-110 at small n.
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include "testutil.hpp"
#include "checker.hpp"
using namespace h2o;

static int run(size_t n, bool with_select, int threads) {
    SplitMix64 rng(0x68326563632d73ull + 2);
    std::vector<NativePoint> pts;
    std::vector<BigUint> scalars;
    JacT<BnFq> G = JacT<BnFq>::from_affine(bn_g1_generator());
    JacT<BnFq> acc = JacT<BnFq>::identity();
    for (size_t i = 0; i < n; i++) {
        BigUint a = rng.below(Fr::modulus()), b = rng.below(Fr::modulus());
        JacT<BnFq> p = G.mul(a);
        acc = acc.add(p.mul(b));
        pts.push_back(to_native(p.to_affine()));
        scalars.push_back(b);
    }
    NativePoint r1 = to_native(G.mul(rng.below(Fr::modulus())).to_affine());
    NativePoint r2 = to_native(G.mul(rng.below(Fr::modulus())).to_affine());
    NativePoint expect = to_native(acc.to_affine());

    auto t0 = std::chrono::steady_clock::now();
    auto ctx = std::make_shared<Context>();
    IntegerContext ic(ctx, BnFq::modulus());
    NativeScalarEccContext ecc = with_select ? NativeScalarEccContext::new_with_select_chip(ic, bn256_g1_params())
                                             : NativeScalarEccContext::new_without_select_chip(ic, bn256_g1_params());
    ecc.n_threads = threads;
    std::vector<AssignedPoint> ap;
    std::vector<AssignedValue> as;
    for (auto& p : pts) ap.push_back(ecc.assign_point(p));
    for (auto& s : scalars) as.push_back(ctx->assign(Fr::from_bn(s)));
    size_t b0 = ctx->base_offset, r0 = ctx->range_offset, s0 = ctx->select_offset;
    AssignedPoint res = ecc.msm_unsafe(ap, as, r1, r2);
    printf("n=%zu select=%d: msm core rows base %zu range %zu select %zu\n", n, (int)with_select, ctx->base_offset - b0,
           ctx->range_offset - r0, ctx->select_offset - s0);
    AssignedPoint res_expect = ecc.assign_point(expect);
    ecc.ecc_assert_equal(res, res_expect);
    auto t1 = std::chrono::steady_clock::now();
    double secs = std::chrono::duration<double>(t1 - t0).count();
    size_t cells = 0;
    {
        RecordsInner& in = *ctx->records.inner;
        for (auto& c : in.base_adv) cells += c.present;
        for (auto& c : in.range_adv) cells += c.present;
        for (auto& c : in.select_adv) cells += c.present;
    }
    printf("  offsets base %zu range %zu select %zu | heights %zu %zu %zu | perms %zu | adv cells %zu | %.3f s (%.0f cells/s)\n",
           ctx->base_offset, ctx->range_offset, ctx->select_offset, ctx->records.base_height, ctx->records.range_height,
           ctx->records.select_height, ctx->records.permutations.size(), cells, secs, cells / secs);
    CheckReport rep = check_records(ctx->records);
    printf("  check: %s %s\n", rep.ok() ? "OK" : "FAIL", rep.first_error.c_str());
    return rep.ok() ? 0 : 1;
}

int main(int argc, char** argv) {
    init_fields();
    size_t n = argc > 1 ? atoi(argv[1]) : 12;
    int threads = argc > 2 ? atoi(argv[2]) : 1;
    int rc = 0;
    rc |= run(n, true, threads);
    if (argc <= 3) rc |= run(std::min<size_t>(n, 8), false, threads);
    return rc;
}
