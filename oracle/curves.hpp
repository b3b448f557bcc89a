// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Not part of the product path.
//
// Native (out-of-circuit) curve arithmetic standing in for the un-vendored `pairing_bn256` /
// `bls12_381` crates, which the reference's tests use to make inputs and expected values
// (src/tests/native_scalar_ecc_chip.rs:15-26, native_scalar_pairing_chip.rs:26-27,
// general_scalar_pairing_chip.rs:26-29,79-84).  Short-Weierstrass y^2 = x^3 + b, Jacobian
// coordinates; results are affine canonical values, which are unique.
#pragma once
#include "field.hpp"

namespace h2o {

// Fq2 = Fq[u]/(u^2+1) for both bn256 and bls12_381
template <class F>
struct Fq2T {
    F c0, c1;
    static Fq2T zero() { return {F::zero(), F::zero()}; }
    static Fq2T one() { return {F::one(), F::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fq2T& o) const { return c0 == o.c0 && c1 == o.c1; }
    bool operator!=(const Fq2T& o) const { return !(*this == o); }
    Fq2T operator+(const Fq2T& o) const { return {c0 + o.c0, c1 + o.c1}; }
    Fq2T operator-(const Fq2T& o) const { return {c0 - o.c0, c1 - o.c1}; }
    Fq2T operator-() const { return {-c0, -c1}; }
    Fq2T operator*(const Fq2T& o) const {
        F a = c0 * o.c0, b = c1 * o.c1;
        F c = (c0 + c1) * (o.c0 + o.c1);
        return {a - b, c - a - b};
    }
    Fq2T square() const { return (*this) * (*this); }
    Fq2T dbl() const { return *this + *this; }
    bool invert(Fq2T& out) const {
        F t = c0.square() + c1.square();
        F ti;
        if (!t.invert(ti)) return false;
        out = {c0 * ti, -(c1 * ti)};
        return true;
    }
    static Fq2T from_u64(uint64_t x) { return {F::from_u64(x), F::zero()}; }
};

template <class F>
struct AffineT {
    F x, y;
    bool inf;
    static AffineT identity() { return {F::zero(), F::zero(), true}; }
    AffineT neg() const { return {x, -y, inf}; }
    bool operator==(const AffineT& o) const { return (inf && o.inf) || (!inf && !o.inf && x == o.x && y == o.y); }
};

template <class F>
struct JacT {
    F x, y, z;  // z == 0 <=> identity
    static JacT identity() { return {F::one(), F::one(), F::zero()}; }
    static JacT from_affine(const AffineT<F>& a) {
        if (a.inf) return identity();
        return {a.x, a.y, F::one()};
    }
    bool is_identity() const { return z.is_zero(); }
    JacT dbl() const {
        if (is_identity() || y.is_zero()) return identity();
        F a = x.square(), b = y.square(), c = b.square();
        F d = ((x + b).square() - a - c).dbl();
        F e = a.dbl() + a;
        F f = e.square();
        F x3 = f - d.dbl();
        F y3 = e * (d - x3) - c.dbl().dbl().dbl();
        F z3 = (y * z).dbl();
        return {x3, y3, z3};
    }
    JacT add(const JacT& o) const {
        if (is_identity()) return o;
        if (o.is_identity()) return *this;
        F z1z1 = z.square(), z2z2 = o.z.square();
        F u1 = x * z2z2, u2 = o.x * z1z1;
        F s1 = y * o.z * z2z2, s2 = o.y * z * z1z1;
        if (u1 == u2) {
            if (s1 == s2) return dbl();
            return identity();
        }
        F h = u2 - u1;
        F i = h.dbl().square();
        F j = h * i;
        F r = (s2 - s1).dbl();
        F v = u1 * i;
        F x3 = r.square() - j - v.dbl();
        F y3 = r * (v - x3) - (s1 * j).dbl();
        F z3 = ((z + o.z).square() - z1z1 - z2z2) * h;
        return {x3, y3, z3};
    }
    AffineT<F> to_affine() const {
        if (is_identity()) return AffineT<F>::identity();
        F zi;
        z.invert(zi);
        F zi2 = zi.square();
        return {x * zi2, y * zi2 * zi, false};
    }
    JacT mul(const BigUint& s) const {
        JacT r = identity();
        for (int i = (int)s.bits() - 1; i >= 0; i--) {
            r = r.dbl();
            if (s.bit(i)) r = r.add(*this);
        }
        return r;
    }
};

template <class F>
inline bool on_curve(const AffineT<F>& p, const F& b) {
    if (p.inf) return true;
    return p.y.square() == p.x.square() * p.x + b;
}

typedef Fq2T<BnFq> BnFq2;
typedef Fq2T<BlsFq> BlsFq2;
typedef AffineT<BnFq> BnG1;
typedef AffineT<BnFq2> BnG2;
typedef AffineT<BlsFq> BlsG1;
typedef AffineT<BlsFq2> BlsG2;

inline BnG1 bn_g1_generator() { return {BnFq::from_u64(1), BnFq::from_u64(2), false}; }
inline BnFq bn_g1_b() { return BnFq::from_u64(3); }
inline BnFq2 bn_g2_b() {  // 3 / (9 + u)
    BnFq2 xi = {BnFq::from_u64(9), BnFq::from_u64(1)}, xi_inv;
    xi.invert(xi_inv);
    return BnFq2::from_u64(3) * xi_inv;
}
inline BnG2 bn_g2_generator() {
    auto f = [](const char* d) { return BnFq::from_bn(BigUint::from_dec(d)); };
    return {{f("10857046999023057135944570762232829481370756359578518086990519993285655852781"),
             f("11559732032986387107991004021392285783925812861821192530917403151452391805634")},
            {f("8495653923123431417604973247489272438418190587263600148770280649306958101930"),
             f("4082367875863433681332203403145435568316851327593401208105741076214120093531")},
            false};
}
inline BlsG1 bls_g1_generator() {
    auto f = [](const char* h) { return BlsFq::from_bn(BigUint::from_hex(h)); };
    return {f("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"),
            f("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1"),
            false};
}
inline BlsFq bls_g1_b() { return BlsFq::from_u64(4); }
inline BlsFq2 bls_g2_b() { return {BlsFq::from_u64(4), BlsFq::from_u64(4)}; }
inline BlsG2 bls_g2_generator() {
    auto f = [](const char* h) { return BlsFq::from_bn(BigUint::from_hex(h)); };
    return {{f("024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"),
             f("13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e")},
            {f("0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801"),
             f("0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be")},
            false};
}

}  // namespace h2o
