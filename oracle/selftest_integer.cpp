// ORACLE — TEST INFRASTRUCTURE ONLY. Quick self-check mirroring src/tests/integer_chip.rs:11-99.
#include <cstdio>
#include "integer_chip.hpp"
#include "checker.hpp"
using namespace h2o;

static uint64_t sm_state = 0x68326563632d73ull;
static uint64_t splitmix() {
    uint64_t z = (sm_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static BigUint rand_below(const BigUint& m) {
    for (;;) {
        uint64_t w[8];
        for (int i = 0; i < 8; i++) w[i] = splitmix();
        int words = (int)((m.bits() + 63) / 64);
        BigUint x = BigUint::from_limbs(w, words);
        x = x & ((BigUint(1) << m.bits()) - BigUint(1));
        if (x < m) return x;
    }
}

static int run(const char* name, const BigUint& wmod, int muls) {
    auto ctx = std::make_shared<Context>();
    IntegerContext ic(ctx, wmod);
    printf("%s: limbs=%lu w_ceil_bits=%lu d_bits=%lu mul_check=%lu reduce_check=%lu pure_w=%lu\n", name,
           ic.info->limbs, ic.info->w_ceil_bits, ic.info->d_bits, ic.info->mul_check_limbs,
           ic.info->reduce_check_limbs, ic.info->pure_w_check_limbs);
    BigUint a = rand_below(wmod), b = rand_below(wmod);
    BigUint binv;
    BigUint::invmod(b, wmod, binv);
    BigUint c = (a + b) % wmod, d = (a + wmod - b) % wmod, e = (a * b) % wmod, f = (a * binv) % wmod;
    AssignedInteger A = ic.assign_w(a), B = ic.assign_w(b);
    size_t b0 = ctx->base_offset, r0 = ctx->range_offset;
    AssignedInteger e2 = ic.int_mul(A, B);
    printf("  int_mul rows: base %zu range %zu\n", ctx->base_offset - b0, ctx->range_offset - r0);
    AssignedInteger e1 = ic.assign_w(e);
    ic.assert_int_equal(e1, e2);
    AssignedInteger c1 = ic.assign_w(c);
    AssignedInteger c2 = ic.int_add(A, B);
    ic.assert_int_equal(c1, c2);
    AssignedInteger d1 = ic.assign_w(d);
    AssignedInteger d2 = ic.int_sub(A, B);
    ic.assert_int_equal(d1, d2);
    AssignedInteger f1 = ic.assign_w(f);
    b0 = ctx->base_offset, r0 = ctx->range_offset;
    AssignedInteger f2 = ic.int_div(A, B).second;
    printf("  int_div rows: base %zu range %zu\n", ctx->base_offset - b0, ctx->range_offset - r0);
    ic.assert_int_equal(f1, f2);
    AssignedInteger zero = ic.int_sub(A, A);
    auto g = ic.int_div(A, zero);
    ctx->assert_true(g.first);
    for (int i = 0; i < muls; i++) {
        BigUint x = rand_below(wmod), y = rand_below(wmod);
        AssignedInteger X = ic.assign_w(x), Y = ic.assign_w(y);
        AssignedInteger Z = ic.int_mul(X, Y);
        AssignedInteger Z1 = ic.assign_w((x * y) % wmod);
        ic.assert_int_equal(Z, Z1);
    }
    CheckReport rep = check_records(ctx->records);
    printf("  heights base %zu range %zu select %zu perms %zu -> %s %s\n", ctx->records.base_height,
           ctx->records.range_height, ctx->records.select_height, ctx->records.permutations.size(),
           rep.ok() ? "OK" : "FAIL", rep.first_error.c_str());
    return rep.ok() ? 0 : 1;
}

int main() {
    init_fields();
    int rc = 0;
    rc |= run("bn256 Fq over Fr", BnFq::modulus(), 100);
    rc |= run("bls12_381 Fq over Fr", BlsFq::modulus(), 100);
    rc |= run("bls12_381 Fr over Fr", BlsFr::modulus(), 100);
    return rc;
}
