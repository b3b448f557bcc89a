// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Not part of the product path.
//
// Prime fields used by the reference through the un-vendored crates `pairing_bn256` 0.1.1
// (Cargo.lock:693-695) and `bls12_381` 0.7.0 (Cargo.lock:94-96): bn256 Fr (the native field N
// of every circuit), bn256 Fq, bls12_381 Fq / Fr.  Textbook Montgomery arithmetic; the canonical
// representative is what crosses into integers (`field_to_bn` / `bn_to_field`, utils.rs:4-17).
#pragma once
#include "bigint.hpp"

namespace h2o {

template <int L, int TAG>
struct Fp {
    uint64_t v[L];  // Montgomery form

    struct Params {
        uint64_t p[L];
        uint64_t r[L];   // R mod p
        uint64_t r2[L];  // R^2 mod p
        uint64_t inv;    // -p^{-1} mod 2^64
        BigUint modulus;
        bool ready = false;
    };
    static Params& P() {
        static Params params;
        return params;
    }
    static void init(const char* modulus_hex) {
        Params& q = P();
        if (q.ready) return;
        q.modulus = BigUint::from_hex(modulus_hex);
        for (int i = 0; i < L; i++) q.p[i] = q.modulus.w[i];
        BigUint R = (BigUint(1) << (64 * L)) % q.modulus;
        BigUint R2 = (R * R) % q.modulus;
        for (int i = 0; i < L; i++) {
            q.r[i] = R.w[i];
            q.r2[i] = R2.w[i];
        }
        uint64_t inv = 1;
        for (int i = 0; i < 63; i++) {
            inv = inv * inv;
            inv = inv * q.p[0];
        }
        q.inv = (uint64_t)0 - inv;
        q.ready = true;
    }
    static const BigUint& modulus() { return P().modulus; }

    static Fp zero() {
        Fp r;
        for (int i = 0; i < L; i++) r.v[i] = 0;
        return r;
    }
    static Fp one() {
        Fp r;
        for (int i = 0; i < L; i++) r.v[i] = P().r[i];
        return r;
    }
    static Fp from_u64(uint64_t x) {
        Fp r = zero();
        r.v[0] = x;
        return mont_mul(r, r2());
    }
    static Fp r2() {
        Fp r;
        for (int i = 0; i < L; i++) r.v[i] = P().r2[i];
        return r;
    }
    // bn_to_field: reduces mod p (utils.rs:10-17)
    static Fp from_bn(const BigUint& bn) {
        BigUint x = bn % P().modulus;
        Fp r;
        for (int i = 0; i < L; i++) r.v[i] = x.w[i];
        return mont_mul(r, r2());
    }
    // raw Montgomery limbs (bls12_381 `Fq::from_raw_unchecked`, bls12_381_pairing_chip.rs:58-107)
    static Fp from_raw_mont(const uint64_t* limbs) {
        Fp r;
        for (int i = 0; i < L; i++) r.v[i] = limbs[i];
        return r;
    }
    // field_to_bn: canonical value (utils.rs:4-8)
    BigUint to_bn() const {
        Fp one_raw = zero();
        one_raw.v[0] = 1;
        Fp c = mont_mul(*this, one_raw);
        return BigUint::from_limbs(c.v, L);
    }
    void to_canonical(uint64_t* out) const {
        Fp one_raw = zero();
        one_raw.v[0] = 1;
        Fp c = mont_mul(*this, one_raw);
        for (int i = 0; i < L; i++) out[i] = c.v[i];
    }

    bool operator==(const Fp& o) const {
        for (int i = 0; i < L; i++)
            if (v[i] != o.v[i]) return false;
        return true;
    }
    bool operator!=(const Fp& o) const { return !(*this == o); }
    bool is_zero() const {
        for (int i = 0; i < L; i++)
            if (v[i]) return false;
        return true;
    }

    static bool geq_p(const uint64_t* a) {
        const uint64_t* p = P().p;
        for (int i = L - 1; i >= 0; i--)
            if (a[i] != p[i]) return a[i] > p[i];
        return true;
    }
    Fp operator+(const Fp& o) const {
        Fp r;
        u128 c = 0;
        for (int i = 0; i < L; i++) {
            c += (u128)v[i] + o.v[i];
            r.v[i] = (uint64_t)c;
            c >>= 64;
        }
        if (c || geq_p(r.v)) {
            uint64_t b = 0;
            for (int i = 0; i < L; i++) {
                u128 t = (u128)r.v[i] - P().p[i] - b;
                r.v[i] = (uint64_t)t;
                b = (uint64_t)(t >> 64) & 1;
            }
        }
        return r;
    }
    Fp operator-(const Fp& o) const {
        Fp r;
        uint64_t b = 0;
        for (int i = 0; i < L; i++) {
            u128 t = (u128)v[i] - o.v[i] - b;
            r.v[i] = (uint64_t)t;
            b = (uint64_t)(t >> 64) & 1;
        }
        if (b) {
            u128 c = 0;
            for (int i = 0; i < L; i++) {
                c += (u128)r.v[i] + P().p[i];
                r.v[i] = (uint64_t)c;
                c >>= 64;
            }
        }
        return r;
    }
    Fp operator-() const { return zero() - *this; }
    static Fp mont_mul(const Fp& a, const Fp& b) {
        const Params& q = P();
        uint64_t t[L + 2];
        for (int i = 0; i < L + 2; i++) t[i] = 0;
        for (int i = 0; i < L; i++) {
            u128 c = 0;
            for (int j = 0; j < L; j++) {
                c += (u128)a.v[j] * b.v[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[L];
            t[L] = (uint64_t)c;
            t[L + 1] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * q.inv;
            c = (u128)m * q.p[0] + t[0];
            c >>= 64;
            for (int j = 1; j < L; j++) {
                c += (u128)m * q.p[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[L];
            t[L - 1] = (uint64_t)c;
            t[L] = t[L + 1] + (uint64_t)(c >> 64);
        }
        Fp r;
        for (int i = 0; i < L; i++) r.v[i] = t[i];
        if (t[L] || geq_p(r.v)) {
            uint64_t bw = 0;
            for (int i = 0; i < L; i++) {
                u128 s = (u128)r.v[i] - q.p[i] - bw;
                r.v[i] = (uint64_t)s;
                bw = (uint64_t)(s >> 64) & 1;
            }
        }
        return r;
    }
    Fp operator*(const Fp& o) const { return mont_mul(*this, o); }
    Fp square() const { return mont_mul(*this, *this); }
    Fp dbl() const { return *this + *this; }
    Fp pow(const BigUint& e) const {
        Fp r = one(), b = *this;
        uint64_t nb = e.bits();
        for (uint64_t i = 0; i < nb; i++) {
            if (e.bit(i)) r = r * b;
            b = b.square();
        }
        return r;
    }
    // Field::invert(): None for zero
    bool invert(Fp& out) const {
        if (is_zero()) return false;
        out = pow(P().modulus - BigUint(2));
        return true;
    }
    Fp inv_or_zero() const {
        Fp r;
        if (!invert(r)) return zero();
        return r;
    }
};

// Field tags
typedef Fp<4, 0> Fr;      // bn256 scalar field = native field N of all circuits
typedef Fp<4, 1> BnFq;    // bn256 base field
typedef Fp<6, 2> BlsFq;   // bls12_381 base field
typedef Fp<4, 3> BlsFr;   // bls12_381 scalar field

static const char* BN256_FR_HEX = "30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001";
static const char* BN256_FQ_HEX = "30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47";
static const char* BLS_FQ_HEX =
    "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab";
static const char* BLS_FR_HEX = "73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001";

inline void init_fields() {
    Fr::init(BN256_FR_HEX);
    BnFq::init(BN256_FQ_HEX);
    BlsFq::init(BLS_FQ_HEX);
    BlsFr::init(BLS_FR_HEX);
}

}  // namespace h2o
