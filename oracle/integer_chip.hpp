// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Not part of the product path.
//
// L1 range/select ops + L2 integer chip, restating:
//   src/circuit/range_chip.rs:262-348   (decompose_bn, RangeChipOps on IntegerContext)
//   src/circuit/select_chip.rs:99-162   (SelectChipOps, encode_offset)
//   src/circuit/integer_chip.rs:15-686  (IntegerChipOps on IntegerContext)
//   src/assign.rs:31-37                 (AssignedInteger)
#pragma once
#include "range_info.hpp"

namespace h2o {

struct AssignedInteger {  // assign.rs:31-37
    std::vector<AssignedValue> limbs_le;
    AssignedValue native;
    uint64_t times = 1;
    AssignedInteger() {}
    AssignedInteger(const std::vector<AssignedValue>& l, const AssignedValue& n, uint64_t t)
        : limbs_le(l), native(n), times(t) {}
};

// range_chip.rs:270-280
inline void decompose_bn(const BigUint& bn, uint64_t decompose, const BigUint& mask, Fr& v, std::vector<Fr>& out) {
    v = Fr::from_bn(bn);
    out.clear();
    for (uint64_t i = 0; i < decompose; i++) out.push_back(Fr::from_bn((bn >> (i * COMMON_RANGE_BITS)) & mask));
}

// select_chip.rs:118-122
inline Fr encode_offset(size_t g, size_t offset, size_t limb_offset) {
    return Fr::from_bn((BigUint((uint64_t)offset) << 128) + (BigUint((uint64_t)g) << 64) +
                       BigUint((uint64_t)limb_offset));
}

// context.rs:161-188 — Rc<RefCell<Context<N>>> + Arc<RangeInfo<W,N>>
struct IntegerContext {
    std::shared_ptr<Context> ctx;
    std::shared_ptr<RangeInfo> info;

    IntegerContext(std::shared_ptr<Context> c, const BigUint& w_modulus)
        : ctx(c), info(std::make_shared<RangeInfo>(w_modulus)) {}
    IntegerContext(std::shared_ptr<Context> c, std::shared_ptr<RangeInfo> i) : ctx(c), info(i) {}

    typedef Context::Elem Elem;

    // ---------------- RangeChipOps (range_chip.rs:282-348) ----------------
    AssignedValue assign_common(const BigUint& bn) {  // :287-298
        Fr v = Fr::from_bn(bn);
        size_t offset = ctx->range_offset;
        AssignedValue res = ctx->records.assign_one_line_range_value(offset, &v, v, COMMON_RANGE_BITS);
        ctx->range_offset += 1;
        return res;
    }
    AssignedValue assign_nonleading_limb(const BigUint& bn) {  // :300-315
        Fr v;
        std::vector<Fr> dv;
        decompose_bn(bn, MAX_CHUNKS * 2, info->common_range_mask, v, dv);
        size_t offset = ctx->range_offset;
        auto r = ctx->records.assign_range_value(offset, dv, v, info->limb_bits);
        ctx->range_offset += r.second;
        return r.first;
    }
    AssignedValue assign_w_ceil_leading_limb(const BigUint& bn) {  // :317-333
        Fr v;
        std::vector<Fr> dv;
        decompose_bn(bn, info->w_ceil_leading_decompose, info->common_range_mask, v, dv);
        size_t offset = ctx->range_offset;
        auto r = ctx->records.assign_range_value(offset, dv, v, info->w_ceil_bits % info->limb_bits);
        ctx->range_offset += r.second;
        return r.first;
    }
    AssignedValue assign_d_leading_limb(const BigUint& bn) {  // :335-347
        Fr v;
        std::vector<Fr> dv;
        decompose_bn(bn, info->d_leading_decompose, info->common_range_mask, v, dv);
        size_t offset = ctx->range_offset;
        auto r = ctx->records.assign_range_value(offset, dv, v, info->d_bits % info->limb_bits);
        ctx->range_offset += r.second;
        return r.first;
    }

    // ---------------- SelectChipOps (select_chip.rs:124-161) ----------------
    void assign_cache_value(const AssignedValue& v, size_t offset, size_t group_index, size_t selector) {
        size_t select_offset = ctx->select_offset;
        Fr enc = encode_offset(group_index, selector, offset);
        ctx->records.assign_cache_value(select_offset, v, enc);
        ctx->select_offset += 1;
    }
    AssignedValue assign_selected_value(const AssignedValue& v, size_t offset, size_t group_index,
                                        const AssignedValue& selector) {
        size_t select_offset = ctx->select_offset;
        Fr enc = encode_offset(group_index, 0, offset);
        AssignedValue r = ctx->records.assign_select_value(select_offset, v, enc, selector);
        ctx->select_offset += 1;
        return r;
    }

    // ---------------- IntegerChipOps ----------------
    // integer_chip.rs:217-224
    BigUint get_w_bn(const AssignedInteger& a) const {
        BigUint res(0);
        for (int i = (int)info->limbs - 1; i >= 0; i--) {
            res = res << info->limb_bits;
            res = res + a.limbs_le[i].val.to_bn();
        }
        return res;
    }

    // integer_chip.rs:73-193
    void add_constraints_for_mul_equation_on_limbs(const AssignedInteger& a, const AssignedInteger& b,
                                                   const std::vector<AssignedValue>& d, const AssignedInteger& rem) {
        if (!(a.times < info->overflow_limit)) throw PanicError("mul: a.times >= overflow_limit");
        if (!(b.times < info->overflow_limit)) throw PanicError("mul: b.times >= overflow_limit");
        if (!(rem.times == 1)) throw PanicError("mul: rem.times != 1");
        Fr one = Fr::one();
        size_t L = (size_t)info->limbs;

        std::vector<AssignedValue> limbs;
        for (size_t pos = 0; pos < (size_t)info->mul_check_limbs; pos++) {
            size_t r_bound = std::min(pos + 1, L);
            size_t l_bound = pos >= L - 1 ? pos - (L - 1) : 0;
            std::vector<Context::MulAddTerm> ls;
            for (size_t i = l_bound; i < r_bound; i++) {
                Context::MulAddTerm t;
                t.a = &a.limbs_le[i];
                t.b = &b.limbs_le[pos - i];
                t.c = &d[i];
                t.c_coeff = -info->w_modulus_limbs_le[pos - i];
                ls.push_back(t);
            }
            limbs.push_back(ctx->mul_add_with_next_line(ls));
        }

        Fr borrow = Fr::from_u64(info->limbs) * info->limb_modulus_n + Fr::from_u64(2);

        // check sum limb[0]
        Fr k0 = info->limb_modulus_n * borrow;
        AssignedValue u = ctx->sum_with_constant({Elem(&limbs[0], one), Elem(&rem.limbs_le[0], -one)}, &k0);
        BigUint v, r;
        BigUint::div_rem(u.val.to_bn(), info->limb_modulus, v, r);
        if (!r.is_zero()) throw PanicError("mul: u0 not divisible by limb modulus");
        BigUint v_h_bn, v_l_bn;
        BigUint::div_rem(v, info->limb_modulus, v_h_bn, v_l_bn);
        AssignedValue v_h = assign_common(v_h_bn);
        AssignedValue v_l = assign_nonleading_limb(v_l_bn);
        ctx->one_line_with_last({pr(v_h, info->limb_coeffs[2]), pr(v_l, info->limb_coeffs[1])}, pr(u, -one), nullptr,
                                {}, nullptr);

        Fr k1 = info->limb_modulus_n * borrow - borrow;
        // check sum limb[1..] with carry
        for (size_t i = 1; i < L; i++) {
            AssignedValue u2 = ctx->sum_with_constant({Elem(&limbs[i], one), Elem(&rem.limbs_le[i], -one),
                                                       Elem(&v_h, info->limb_coeffs[1]),
                                                       Elem(&v_l, info->limb_coeffs[0])},
                                                      &k1);
            BigUint::div_rem(u2.val.to_bn(), info->limb_modulus, v, r);
            if (!r.is_zero()) throw PanicError("mul: u not divisible by limb modulus");
            BigUint::div_rem(v, info->limb_modulus, v_h_bn, v_l_bn);
            v_h = assign_common(v_h_bn);
            v_l = assign_nonleading_limb(v_l_bn);
            ctx->one_line_with_last({pr(v_h, info->limb_coeffs[2]), pr(v_l, info->limb_coeffs[1])}, pr(u2, -one),
                                    nullptr, {}, nullptr);
        }
        assert(info->limbs <= info->mul_check_limbs);
        // Only required by bls12_381 base field
        for (size_t i = L; i < (size_t)info->mul_check_limbs; i++) {
            AssignedValue u2 = ctx->sum_with_constant(
                {Elem(&limbs[i], one), Elem(&v_h, info->limb_coeffs[1]), Elem(&v_l, info->limb_coeffs[0])}, &k1);
            BigUint::div_rem(u2.val.to_bn(), info->limb_modulus, v, r);
            if (!r.is_zero()) throw PanicError("mul: u not divisible by limb modulus");
            BigUint::div_rem(v, info->limb_modulus, v_h_bn, v_l_bn);
            v_h = assign_common(v_h_bn);
            v_l = assign_nonleading_limb(v_l_bn);
            ctx->one_line_with_last({pr(v_h, info->limb_coeffs[2]), pr(v_l, info->limb_coeffs[1])}, pr(u2, -one),
                                    nullptr, {}, nullptr);
        }
    }

    // integer_chip.rs:195-215
    void add_constraints_for_mul_equation_on_native(const AssignedInteger& a, const AssignedInteger& b,
                                                    const AssignedValue& d_native, const AssignedInteger& rem) {
        Fr zero = Fr::zero(), one = Fr::one();
        ctx->one_line({pr(a.native, zero), pr(b.native, zero), pr(d_native, info->w_native), pr(rem.native, one)},
                      nullptr, {-one}, nullptr);
    }

    // integer_chip.rs:236-258
    AssignedInteger assign_w(const BigUint& w) {
        std::vector<AssignedValue> limbs;
        for (uint64_t i = 0; i + 1 < info->limbs; i++)
            limbs.push_back(assign_nonleading_limb((w >> (i * info->limb_bits)) & info->limb_mask));
        limbs.push_back(assign_w_ceil_leading_limb((w >> ((info->limbs - 1) * info->limb_bits)) & info->limb_mask));
        std::vector<Elem> schemas;
        for (size_t i = 0; i < limbs.size(); i++) schemas.push_back(Elem(&limbs[i], info->limb_coeffs[i]));
        AssignedValue native = ctx->sum_with_constant(schemas, nullptr);
        return AssignedInteger(limbs, native, 1);
    }

    // integer_chip.rs:260-281
    std::pair<std::vector<AssignedValue>, AssignedValue> assign_d(const BigUint& d) {
        std::vector<AssignedValue> limbs;
        for (uint64_t i = 0; i + 1 < info->limbs; i++)
            limbs.push_back(assign_nonleading_limb((d >> (i * info->limb_bits)) & info->limb_mask));
        limbs.push_back(assign_d_leading_limb((d >> ((info->limbs - 1) * info->limb_bits)) & info->limb_mask));
        std::vector<Elem> schemas;
        for (size_t i = 0; i < limbs.size(); i++) schemas.push_back(Elem(&limbs[i], info->limb_coeffs[i]));
        AssignedValue native = ctx->sum_with_constant(schemas, nullptr);
        return std::make_pair(limbs, native);
    }

    // integer_chip.rs:283-373
    AssignedInteger reduce(const AssignedInteger& a) {
        if (a.times == 1) return a;
        Fr zero = Fr::zero(), one = Fr::one();
        uint64_t overflow_limit = info->overflow_limit;
        if (!(a.times < overflow_limit)) throw PanicError("reduce: times >= overflow_limit");

        BigUint a_bn = get_w_bn(a);
        BigUint d, rem;
        BigUint::div_rem(a_bn, info->w_modulus, d, rem);

        AssignedInteger assigned_rem = assign_w(rem);
        AssignedValue assigned_d = assign_common(d);

        ctx->one_line_with_last({pr(assigned_d, info->w_native), pr(assigned_rem.native, one)}, pr(a.native, -one),
                                nullptr, {}, nullptr);

        bool have_last = false;
        AssignedValue last_v;
        for (size_t i = 0; i < (size_t)info->reduce_check_limbs; i++) {
            uint64_t last_borrow = i != 0 ? overflow_limit : 0;
            BigUint carry = have_last ? last_v.val.to_bn() : BigUint(0);
            BigUint u = d * info->w_modulus_limbs_le_bn[i] + info->bn_to_limb_le(rem)[i] +
                        info->limb_modulus * BigUint(overflow_limit) - a.limbs_le[i].val.to_bn() + carry -
                        BigUint(last_borrow);
            BigUint v, v_rem;
            BigUint::div_rem(u, info->limb_modulus, v, v_rem);
            if (!v_rem.is_zero()) throw PanicError("reduce: u not divisible");
            AssignedValue va = assign_nonleading_limb(v);
            Fr k = Fr::from_bn(info->limb_modulus * BigUint(overflow_limit) - BigUint(i == 0 ? 0 : overflow_limit));
            ctx->one_line_with_last({pr(assigned_d, info->w_modulus_limbs_le[i]), pr(assigned_rem.limbs_le[i], one),
                                     pr(a.limbs_le[i], -one), have_last ? pr(last_v, one) : pr(zero, zero)},
                                    pr(va, -Fr::from_bn(info->limb_modulus)), &k, {}, nullptr);
            last_v = va;
            have_last = true;
        }
        return assigned_rem;
    }

    // integer_chip.rs:375-382
    AssignedInteger conditionally_reduce(const AssignedInteger& a) {
        uint64_t threshold = 1ull << (info->overflow_bits - 2);
        if (a.times > threshold) return reduce(a);
        return a;
    }

    AssignedValue native_of(const std::vector<AssignedValue>& limbs) {
        std::vector<Elem> schemas;
        for (size_t i = 0; i < limbs.size(); i++) schemas.push_back(Elem(&limbs[i], info->limb_coeffs[i]));
        return ctx->sum_with_constant(schemas, nullptr);
    }

    // integer_chip.rs:384-406
    AssignedInteger int_add(const AssignedInteger& a, const AssignedInteger& b) {
        std::vector<AssignedValue> limbs;
        for (size_t i = 0; i < (size_t)info->limbs; i++) limbs.push_back(ctx->add(a.limbs_le[i], b.limbs_le[i]));
        AssignedValue native = native_of(limbs);
        return conditionally_reduce(AssignedInteger(limbs, native, a.times + b.times));
    }
    // integer_chip.rs:408-437
    AssignedInteger int_sub(const AssignedInteger& a, const AssignedInteger& b) {
        if (b.times >= info->overflow_limit) throw PanicError("int_sub: b.times out of table");
        const std::vector<Fr>& upper_limbs = info->w_modulus_of_ceil_times[b.times];
        Fr one = Fr::one();
        std::vector<AssignedValue> limbs;
        for (size_t i = 0; i < (size_t)info->limbs; i++)
            limbs.push_back(
                ctx->sum_with_constant({Elem(&a.limbs_le[i], one), Elem(&b.limbs_le[i], -one)}, &upper_limbs[i]));
        AssignedValue native = native_of(limbs);
        return conditionally_reduce(AssignedInteger(limbs, native, a.times + b.times + 1));
    }
    // integer_chip.rs:439-464
    AssignedInteger int_neg(const AssignedInteger& a) {
        if (a.times >= info->overflow_limit) throw PanicError("int_neg: a.times out of table");
        const std::vector<Fr>& upper_limbs = info->w_modulus_of_ceil_times[a.times];
        Fr one = Fr::one();
        std::vector<AssignedValue> limbs;
        for (size_t i = 0; i < (size_t)info->limbs; i++)
            limbs.push_back(ctx->sum_with_constant({Elem(&a.limbs_le[i], -one)}, &upper_limbs[i]));
        AssignedValue native = native_of(limbs);
        return conditionally_reduce(AssignedInteger(limbs, native, a.times + 1));
    }
    // integer_chip.rs:466-483
    AssignedInteger int_mul(const AssignedInteger& a, const AssignedInteger& b) {
        BigUint a_bn = get_w_bn(a), b_bn = get_w_bn(b);
        BigUint d, rem;
        BigUint::div_rem(a_bn * b_bn, info->w_modulus, d, rem);
        AssignedInteger rem_a = assign_w(rem);
        auto d_a = assign_d(d);
        add_constraints_for_mul_equation_on_limbs(a, b, d_a.first, rem_a);
        add_constraints_for_mul_equation_on_native(a, b, d_a.second, rem_a);
        return rem_a;
    }
    // integer_chip.rs:485-491
    AssignedInteger int_unsafe_invert(const AssignedInteger& x) {
        AssignedInteger one = assign_int_constant(BigUint(1));
        auto r = int_div(one, x);
        ctx->assert_false(r.first);
        return r.second;
    }
    // integer_chip.rs:493-538
    std::pair<AssignedCondition, AssignedInteger> int_div(const AssignedInteger& a_in, const AssignedInteger& b_in) {
        AssignedInteger b = reduce(b_in);
        AssignedCondition is_b_zero = is_int_zero(b);
        AssignedCondition a_coeff = ctx->not_(is_b_zero);
        AssignedInteger a;
        {
            AssignedInteger ar = reduce(a_in);
            std::vector<AssignedValue> limbs_le;
            for (size_t i = 0; i < (size_t)info->limbs; i++) limbs_le.push_back(ctx->mul(ar.limbs_le[i], a_coeff.v));
            AssignedValue native = ctx->mul(ar.native, a_coeff.v);
            a = AssignedInteger(limbs_le, native, ar.times);
        }
        BigUint a_bn = get_w_bn(a), b_bn = get_w_bn(b);
        // W::invert() of the un-vendored field crate: None for zero, else the canonical inverse
        BigUint binv, c_bn(0);
        if (BigUint::invmod(b_bn, info->w_modulus, binv)) c_bn = ((a_bn % info->w_modulus) * binv) % info->w_modulus;
        BigUint d_bn = (b_bn * c_bn - a_bn) / info->w_modulus;

        AssignedInteger c = assign_w(c_bn);
        auto d = assign_d(d_bn);
        add_constraints_for_mul_equation_on_limbs(b, c, d.first, a);
        add_constraints_for_mul_equation_on_native(b, c, d.second, a);
        return std::make_pair(is_b_zero, c);
    }
    // integer_chip.rs:540-548
    AssignedCondition is_pure_zero(const AssignedInteger& a) {
        Fr one = Fr::one();
        std::vector<Elem> e;
        for (auto& v : a.limbs_le) e.push_back(Elem(&v, one));
        AssignedValue sum = ctx->sum_with_constant(e, nullptr);
        return ctx->is_zero(sum);
    }
    // integer_chip.rs:550-570
    AssignedCondition is_pure_w_modulus(const AssignedInteger& a) {
        if (!(a.times == 1)) throw PanicError("is_pure_w_modulus: times != 1");
        AssignedValue native_diff = ctx->add_constant(a.native, -info->w_native);
        AssignedCondition is_eq = ctx->is_zero(native_diff);
        for (size_t i = 0; i < (size_t)info->pure_w_check_limbs; i++) {
            AssignedValue limb_diff = ctx->add_constant(a.limbs_le[i], -info->w_modulus_limbs_le[i]);
            AssignedCondition is_limb_eq = ctx->is_zero(limb_diff);
            is_eq = ctx->and_(is_eq, is_limb_eq);
        }
        return is_eq;
    }
    // integer_chip.rs:572-578
    AssignedCondition is_int_zero(const AssignedInteger& a_in) {
        AssignedInteger a = reduce(a_in);
        AssignedCondition is_zero = is_pure_zero(a);
        AssignedCondition is_w_modulus = is_pure_w_modulus(a);
        return ctx->or_(is_zero, is_w_modulus);
    }
    // integer_chip.rs:47-54
    AssignedCondition is_int_equal(const AssignedInteger& a, const AssignedInteger& b) {
        AssignedInteger diff = int_sub(a, b);
        return is_int_zero(diff);
    }
    // integer_chip.rs:580-598 (w is the canonical value of the W element)
    AssignedInteger assign_int_constant(const BigUint& w) {
        std::vector<Fr> limbs_value = info->bn_to_limb_le_n(w);
        std::vector<AssignedValue> limbs;
        for (auto& l : limbs_value) limbs.push_back(ctx->assign_constant(l));
        AssignedValue native = ctx->assign_constant(Fr::from_bn(w % info->n_modulus));
        return AssignedInteger(limbs, native, 1);
    }
    // integer_chip.rs:600-612
    void assert_int_equal(const AssignedInteger& a, const AssignedInteger& b) {
        Fr zero = Fr::zero(), one = Fr::one();
        AssignedInteger diff = int_sub(a, b);
        diff = reduce(diff);
        std::vector<Elem> e;
        for (auto& v : diff.limbs_le) e.push_back(Elem(&v, one));
        AssignedValue sum = ctx->sum_with_constant(e, nullptr);
        ctx->assert_constant(sum, zero);
    }
    AssignedInteger int_square(const AssignedInteger& a) { return int_mul(a, a); }  // :614-616
    // integer_chip.rs:618-658
    AssignedInteger int_mul_small_constant(const AssignedInteger& a_in, uint64_t b) {
        uint64_t threshold = 1ull << (info->overflow_bits - 2);
        if (!(b < threshold)) throw PanicError("int_mul_small_constant: b >= threshold");
        AssignedInteger a = a_in;
        if (a_in.times * b >= info->overflow_limit) a = reduce(a_in);
        std::vector<AssignedValue> limbs;
        for (size_t i = 0; i < (size_t)info->limbs; i++)
            limbs.push_back(ctx->sum_with_constant({Elem(&a.limbs_le[i], Fr::from_u64(b))}, nullptr));
        AssignedValue native = native_of(limbs);
        return conditionally_reduce(AssignedInteger(limbs, native, a.times * b));
    }
    // integer_chip.rs:660-681
    AssignedInteger bisec_int(const AssignedCondition& cond, const AssignedInteger& a, const AssignedInteger& b) {
        std::vector<AssignedValue> limbs;
        for (size_t i = 0; i < (size_t)info->limbs; i++) limbs.push_back(ctx->bisec(cond, a.limbs_le[i], b.limbs_le[i]));
        AssignedValue native = ctx->bisec(cond, a.native, b.native);
        return AssignedInteger(limbs, native, std::max(a.times, b.times));
    }
    BigUint get_w(const AssignedInteger& a) const { return get_w_bn(a) % info->w_modulus; }  // :683-685
};

}  // namespace h2o
