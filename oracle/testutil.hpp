// ORACLE — TEST INFRASTRUCTURE ONLY. Seeded synthetic inputs (SplitMix64 seeded
// 0x68326563632d73 + config index; field elements by rejection sampling; points = s*G).
#pragma once
#include "curves.hpp"
#include "ecc_chip.hpp"

namespace h2o {

struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    // rejection sampling of ceil(bits/64) words masked to m.bits()
    BigUint below(const BigUint& m) {
        int words = (int)((m.bits() + 63) / 64);
        BigUint mask = (BigUint(1) << m.bits()) - BigUint(1);
        for (;;) {
            uint64_t w[16];
            for (int i = 0; i < words; i++) w[i] = next();
            BigUint x = BigUint::from_limbs(w, words) & mask;
            if (x < m) return x;
        }
    }
};

template <class F>
inline NativePoint to_native(const AffineT<F>& p) {
    NativePoint n;
    n.is_identity = p.inf;
    if (!p.inf) {
        n.x = p.x.to_bn();
        n.y = p.y.to_bn();
    }
    return n;
}

inline CurveParams bn256_g1_params() {
    CurveParams cp;
    cp.base_modulus = BnFq::modulus();
    cp.scalar_modulus = Fr::modulus();
    cp.b = BigUint(3);
    cp.generator = to_native(bn_g1_generator());
    cp.scalar_num_bits = 254;
    return cp;
}
inline CurveParams bls12_381_g1_params() {
    CurveParams cp;
    cp.base_modulus = BlsFq::modulus();
    cp.scalar_modulus = BlsFr::modulus();
    cp.b = BigUint(4);
    cp.generator = to_native(bls_g1_generator());
    cp.scalar_num_bits = 255;
    return cp;
}

}  // namespace h2o
