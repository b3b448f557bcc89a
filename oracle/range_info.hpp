// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Not part of the product path.
//
// Restates src/range_info.rs:13-360 (RangeInfo<W,N>::new, pre_check, d_bits, bn_to_limb_le[_n],
// find_w_modulus_of_ceil_times).  W is given at run time by its modulus; N is bn256 Fr.
#pragma once
#include <vector>
#include "records.hpp"

namespace h2o {

struct RangeInfo {
    uint64_t limbs, limb_bits;
    uint64_t w_ceil_leading_decompose, n_floor_leading_decompose, d_leading_decompose;
    uint64_t w_ceil_bits, d_bits, n_floor_bits;
    uint64_t d_leading_bits, w_ceil_leading_bits, n_floor_leading_bits;
    BigUint w_ceil, n_modulus, w_modulus, common_range_mask, limb_mask, limb_modulus, max_d;
    std::vector<BigUint> w_modulus_limbs_le_bn;
    std::vector<Fr> w_modulus_limbs_le;
    std::vector<Fr> limb_coeffs;
    Fr limb_modulus_n;
    uint64_t overflow_bits, overflow_limit;
    Fr w_native;
    uint64_t pure_w_check_limbs, reduce_check_limbs, mul_check_limbs;
    std::vector<std::vector<Fr>> w_modulus_of_ceil_times;  // index 0 unused (None)

    // range_info.rs:57-75
    static void bits_to_leading_bits_and_decompose(uint64_t bits, uint64_t common_bits, uint64_t& leading_bits,
                                                   uint64_t& decompose) {
        uint64_t common_limb_bits = RANGE_VALUE_DECOMPOSE * common_bits;
        leading_bits = (bits % common_limb_bits == 0) ? common_limb_bits : bits % common_limb_bits;
        if (!(leading_bits >= 2 * common_bits)) throw PanicError("leading bits < 2*common");
        if (!(leading_bits <= RANGE_VALUE_DECOMPOSE * common_bits)) throw PanicError("leading bits too large");
        uint64_t leading_chunk_bits = leading_bits % common_bits;
        if (leading_chunk_bits == 0) {
            uint64_t lb = leading_bits;
            leading_bits = common_bits;
            decompose = lb / common_bits;
        } else {
            uint64_t lb = leading_bits;
            leading_bits = leading_chunk_bits;
            decompose = lb / common_bits + 1;
        }
    }

    // range_info.rs:299-314
    static uint64_t compute_d_bits(const BigUint& w_modulus, uint64_t overflow_bits) {
        BigUint w_max = w_modulus - BigUint(1);
        uint64_t w_ceil_bits = w_max.bits();
        uint64_t d_bits = w_ceil_bits + overflow_bits * 2 + 1;
        BigUint max_a = BigUint(1) << (w_ceil_bits + overflow_bits);
        if (!((BigUint(1) << d_bits) * w_modulus >= max_a * max_a)) throw PanicError("d_bits completeness");
        return d_bits;
    }

    // range_info.rs:77-184
    RangeInfo(const BigUint& w_mod, uint64_t common_bits = COMMON_RANGE_BITS, uint64_t overflow_bits_ = OVERFLOW_BITS) {
        assert(common_bits == COMMON_RANGE_BITS);
        assert(overflow_bits_ == OVERFLOW_BITS);
        BigUint w_max = w_mod - BigUint(1);
        w_ceil_bits = w_max.bits();
        bits_to_leading_bits_and_decompose(w_ceil_bits, common_bits, w_ceil_leading_bits, w_ceil_leading_decompose);

        BigUint n_max = Fr::modulus() - BigUint(1);
        n_floor_bits = n_max.bits() - 1;
        bits_to_leading_bits_and_decompose(n_floor_bits, common_bits, n_floor_leading_bits, n_floor_leading_decompose);

        d_bits = compute_d_bits(w_mod, overflow_bits_);
        bits_to_leading_bits_and_decompose(d_bits, common_bits, d_leading_bits, d_leading_decompose);

        limb_bits = common_bits * RANGE_VALUE_DECOMPOSE;
        limbs = (w_ceil_bits + limb_bits - 1) / limb_bits;

        max_d = BigUint(1) << d_bits;
        limb_mask = (BigUint(1) << limb_bits) - BigUint(1);
        n_modulus = n_max + BigUint(1);
        w_modulus = w_max + BigUint(1);
        BigUint w_native_bn = w_modulus % n_modulus;

        for (uint64_t i = 0; i < limbs; i++) {
            w_modulus_limbs_le_bn.push_back((w_modulus >> (i * limb_bits)) & limb_mask);
            w_modulus_limbs_le.push_back(Fr::from_bn(w_modulus_limbs_le_bn.back()));
        }
        limb_modulus = BigUint(1) << limb_bits;
        limb_modulus_n = Fr::from_bn(limb_modulus);
        overflow_bits = overflow_bits_;
        overflow_limit = 1ull << overflow_bits;
        w_ceil = BigUint(1) << w_ceil_bits;
        common_range_mask = BigUint((1ull << common_bits) - 1);
        for (uint64_t i = 0; i < limbs; i++) limb_coeffs.push_back(Fr::from_bn(BigUint(1) << (i * limb_bits)));
        w_native = Fr::from_bn(w_native_bn);

        pure_w_check_limbs = (w_ceil_bits - n_floor_bits + limb_bits - 1) / limb_bits;
        mul_check_limbs =
            (std::max(w_ceil_bits * 2 + overflow_bits * 2, d_bits + w_ceil_bits) - n_floor_bits + limb_bits - 1) /
            limb_bits;
        reduce_check_limbs =
            (std::max(w_ceil_bits + overflow_bits, common_bits + w_ceil_bits) - n_floor_bits + limb_bits - 1) /
            limb_bits;

        w_modulus_of_ceil_times.resize(overflow_limit);
        for (uint64_t i = 1; i < overflow_limit; i++) w_modulus_of_ceil_times[i] = find_w_modulus_of_ceil_times(i);

        pre_check();
    }

    // range_info.rs:186-297
    void pre_check() const {
        uint64_t common_modulus = 1ull << COMMON_RANGE_BITS;
        auto chk = [](bool c, const char* what) {
            if (!c) throw PanicError(std::string("RangeInfo::pre_check: ") + what);
        };
        {
            BigUint limb_check_modulus = BigUint(1) << (limb_bits * pure_w_check_limbs);
            chk(BigUint::lcm(n_modulus, limb_check_modulus) >= w_ceil, "pure_w lcm");
        }
        BigUint max_wi = w_modulus_limbs_le_bn[0];
        for (auto& x : w_modulus_limbs_le_bn)
            if (x > max_wi) max_wi = x;
        {
            BigUint max_a = w_ceil * BigUint(overflow_limit - 1) - BigUint(1);
            BigUint max_dd = (BigUint(1) << COMMON_RANGE_BITS) - BigUint(1);
            chk(max_a <= max_dd * w_modulus, "reduce completeness of d");
            BigUint lm = BigUint(1) << (limb_bits * reduce_check_limbs);
            chk(BigUint::lcm(n_modulus, lm) >= max_dd * w_modulus + w_ceil, "reduce soundness of d");
            BigUint max_v = limb_modulus - BigUint(1);
            BigUint max_rem = limb_modulus - BigUint(1);
            chk(max_v * limb_modulus >= max_dd * max_wi + max_rem + max_v + BigUint(overflow_limit) * limb_modulus,
                "reduce completeness of v");
            chk(max_v * limb_modulus < n_modulus, "reduce v overflow");
            chk(max_dd * max_wi + max_rem + max_v + BigUint(overflow_limit) * limb_modulus < n_modulus,
                "reduce sum overflow");
            BigUint max_ai = limb_modulus * BigUint(overflow_limit - 1) - BigUint(1);
            chk(BigUint(overflow_limit) * limb_modulus - BigUint(overflow_limit) >= max_ai, "reduce borrow");
        }
        {
            BigUint max_a = w_ceil * BigUint(overflow_limit - 1) - BigUint(1);
            BigUint max_dd = (BigUint(1) << d_bits) - BigUint(1);
            chk(max_a * max_a <= max_dd * w_modulus, "mul completeness of d");
            BigUint l = BigUint::lcm(n_modulus, BigUint(1) << (limb_bits * mul_check_limbs));
            BigUint max_rem = w_ceil - BigUint(1);
            chk(l > max_a * max_a, "mul lcm > a*b");
            chk(l > max_dd * w_modulus + max_rem, "mul lcm > d*w+rem");
            BigUint borrow = BigUint(limbs) * limb_modulus + BigUint(2);
            BigUint max_d_j = limb_modulus - BigUint(1);
            BigUint max_rem_i = limb_modulus - BigUint(1);
            chk(borrow * limb_modulus - borrow >= BigUint(limbs) * max_d_j * max_wi + max_rem_i, "mul borrow");
            BigUint max_v = limb_modulus * BigUint(common_modulus) - BigUint(1);
            BigUint max_a_j = limb_modulus * BigUint(overflow_limit - 1);
            chk(max_v * limb_modulus >= max_a_j * max_a_j * BigUint(limbs) + limb_modulus * borrow,
                "mul completeness of v");
            chk(max_v * limb_modulus < n_modulus, "mul v overflow");
        }
        chk(limbs >= 3, "limbs >= 3");
    }

    // range_info.rs:316-332
    std::vector<Fr> bn_to_limb_le_n(const BigUint& w) const {
        std::vector<Fr> r;
        for (uint64_t i = 0; i < limbs; i++) r.push_back(Fr::from_bn((w >> (i * limb_bits)) & limb_mask));
        return r;
    }
    std::vector<BigUint> bn_to_limb_le(const BigUint& w) const {
        std::vector<BigUint> r;
        for (uint64_t i = 0; i < limbs; i++) r.push_back((w >> (i * limb_bits)) & limb_mask);
        return r;
    }

    // range_info.rs:334-359
    std::vector<Fr> find_w_modulus_of_ceil_times(uint64_t times) const {
        BigUint max = w_ceil * BigUint(times);
        BigUint nn, rem;
        BigUint::div_rem(max, w_modulus, nn, rem);
        if (rem > BigUint(0)) nn = nn + BigUint(1);
        BigUint upper = w_modulus * nn;
        std::vector<Fr> out;
        for (uint64_t i = 0; i + 1 < limbs; i++) {
            BigUint r = (upper & limb_mask) + limb_modulus * BigUint(times);
            upper = (upper - r) >> limb_bits;
            out.push_back(Fr::from_bn(r));
            if (!(r >= limb_modulus * BigUint(times) - BigUint(1))) throw PanicError("ceil_times rem low");
            if (!(r < limb_modulus * BigUint(times + 1))) throw PanicError("ceil_times rem high");
        }
        BigUint lead = BigUint(1) << (w_ceil_bits % limb_bits);
        if (!(upper >= lead * BigUint(times))) throw PanicError("ceil_times upper low");
        if (!(upper < lead * BigUint(times + 1))) throw PanicError("ceil_times upper high");
        out.push_back(Fr::from_bn(upper));
        return out;
    }
};

}  // namespace h2o
