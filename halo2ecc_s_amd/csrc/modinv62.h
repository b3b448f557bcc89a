// Modular inversion by Bernstein-Yang division steps ("safegcd", https://gcd.cr.yp.to/papers.html#safegcd) in batches of
// 62 steps on the low limbs, the way constant-time big-number libraries do it: every lane of a wave executes the same
// instruction sequence (no data-dependent loops inside a batch), and a batch costs 62 cheap 64-bit steps plus ten
// 62 x 64-bit multiply-accumulate rows - against ~750 full-width shift / subtract rounds with divergent inner loops of
// the binary extended Euclid this replaces (the dominant cost of every batched inversion in the value chain: inverse
// fix-ups, hint finalisation).
//
// Values are signed, in NL = ceil(64 N / 62) limbs of 62 bits: x = sum v[i] 2^(62 i).  modinv62<N>(a, p) returns a^-1 mod p
// for ANY odd modulus p < 2^(64 N) and 0 <= a < p, and 0 for a = 0 (what Field::invert() -> None maps to in the reference:
// base_chip.rs:301, integer_chip.rs:524-527).  What the representation needs is room for the intermediate values, which stay
// inside (-2 p, 2 p): 2 p < 2^(62 NL - 1), i.e. 64 N + 2 <= 62 NL - true for every N up to 15 and asserted in inv<N>() - so
// the 255-bit bls12-381 Fr (above 2^254) is as good a modulus as the 254-bit bn256 fields.  Host-compilable
// (tests/test_modinv_cpu.py builds it with g++).
#pragma once
#include <stdint.h>

#ifndef MI_INLINE
#ifdef __HIPCC__
#define MI_INLINE __device__ __host__ __forceinline__
#else
#define MI_INLINE inline
#endif
#endif

namespace modinv62 {

typedef __int128 i128;
static constexpr int64_t M62 = (int64_t)(UINT64_MAX >> 2);

template <int NL>
struct S62 {
    int64_t v[NL];
};
struct Trans {
    int64_t u, v, q, r;
};

// 62 division steps on the low limbs of f (odd) and g; eta = -delta.  Returns the new eta; t = the transition matrix
// scaled by 2^62: [f', g'] = t [f, g] / 2^62.
MI_INLINE int64_t divsteps_62(int64_t eta, uint64_t f0, uint64_t g0, Trans& t) {
    uint64_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 2
    for (int i = 0; i < 62; i++) {
        uint64_t c1 = (uint64_t)(eta >> 63);   // eta < 0
        uint64_t c2 = -(g & 1);                // g odd
        uint64_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;   // conditionally negated f, u, v
        g += x & c2;
        q += y & c2;
        r += z & c2;
        c1 &= c2;                              // eta < 0 and g odd: swap roles
        eta = (int64_t)(((uint64_t)eta ^ c1) - (c1 + 1));
        f += g & c1;
        u += q & c1;
        v += r & c1;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t.u = (int64_t)u;
    t.v = (int64_t)v;
    t.q = (int64_t)q;
    t.r = (int64_t)r;
    return eta;
}

// [f, g] <- t [f, g] / 2^62 (exact)
template <int NL>
MI_INLINE void update_fg(S62<NL>& f, S62<NL>& g, const Trans& t) {
    i128 cf = (i128)t.u * f.v[0] + (i128)t.v * g.v[0];
    i128 cg = (i128)t.q * f.v[0] + (i128)t.r * g.v[0];
    cf >>= 62;
    cg >>= 62;
#pragma unroll
    for (int i = 1; i < NL; i++) {
        cf += (i128)t.u * f.v[i] + (i128)t.v * g.v[i];
        cg += (i128)t.q * f.v[i] + (i128)t.r * g.v[i];
        f.v[i - 1] = (int64_t)cf & M62;
        g.v[i - 1] = (int64_t)cg & M62;
        cf >>= 62;
        cg >>= 62;
    }
    f.v[NL - 1] = (int64_t)cf;
    g.v[NL - 1] = (int64_t)cg;
}

// [d, e] <- t [d, e] / 2^62 mod p, d and e staying in (-2p, p); pinv = p^-1 mod 2^62
template <int NL>
MI_INLINE void update_de(S62<NL>& d, S62<NL>& e, const Trans& t, const S62<NL>& p, uint64_t pinv) {
    int64_t sd = d.v[NL - 1] >> 63, se = e.v[NL - 1] >> 63;
    int64_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
    i128 cd = (i128)t.u * d.v[0] + (i128)t.v * e.v[0];
    i128 ce = (i128)t.q * d.v[0] + (i128)t.r * e.v[0];
    // multiples of p that clear the low 62 bits
    md -= (int64_t)((pinv * (uint64_t)cd + (uint64_t)md) & (uint64_t)M62);
    me -= (int64_t)((pinv * (uint64_t)ce + (uint64_t)me) & (uint64_t)M62);
    cd += (i128)p.v[0] * md;
    ce += (i128)p.v[0] * me;
    cd >>= 62;
    ce >>= 62;
#pragma unroll
    for (int i = 1; i < NL; i++) {
        cd += (i128)t.u * d.v[i] + (i128)t.v * e.v[i];
        ce += (i128)t.q * d.v[i] + (i128)t.r * e.v[i];
        cd += (i128)p.v[i] * md;
        ce += (i128)p.v[i] * me;
        d.v[i - 1] = (int64_t)cd & M62;
        e.v[i - 1] = (int64_t)ce & M62;
        cd >>= 62;
        ce >>= 62;
    }
    d.v[NL - 1] = (int64_t)cd;
    e.v[NL - 1] = (int64_t)ce;
}

// r in (-2p, p), times sign (+-1) -> [0, p)
template <int NL>
MI_INLINE void normalize(S62<NL>& r, int64_t sign, const S62<NL>& p) {
    int64_t add = r.v[NL - 1] >> 63;   // negative: add p
    int64_t neg = sign >> 63;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int64_t x = r.v[i] + (p.v[i] & add);
        r.v[i] = (x ^ neg) - neg;
    }
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        r.v[i + 1] += r.v[i] >> 62;
        r.v[i] &= M62;
    }
    add = r.v[NL - 1] >> 63;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] += p.v[i] & add;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        r.v[i + 1] += r.v[i] >> 62;
        r.v[i] &= M62;
    }
}

// N 64-bit words (little endian) <-> NL limbs of 62 bits
template <int N, int NL>
MI_INLINE S62<NL> to_s62(const uint64_t* w) {
    S62<NL> r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int bit = 62 * i, k = bit / 64, sh = bit % 64;
        uint64_t lo = k < N ? w[k] >> sh : 0;
        uint64_t hi = (sh > 2 && k + 1 < N) ? w[k + 1] << (64 - sh) : 0;
        r.v[i] = (int64_t)((lo | hi) & (uint64_t)M62);
    }
    return r;
}
template <int N, int NL>
MI_INLINE void from_s62(const S62<NL>& a, uint64_t* w) {   // a normalised: limbs in [0, 2^62)
#pragma unroll
    for (int k = 0; k < N; k++) w[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int bit = 62 * i, k = bit / 64, sh = bit % 64;
        uint64_t x = (uint64_t)a.v[i];
        if (k < N) w[k] |= x << sh;
        if (sh > 2 && k + 1 < N) w[k + 1] |= x >> (64 - sh);
    }
}

// a^-1 mod p (0 for a = 0); a, p, out: N little-endian 64-bit words
template <int N>
MI_INLINE void inv(const uint64_t* a, const uint64_t* pw, uint64_t* out) {
    constexpr int NL = (64 * N + 61) / 62;
    static_assert(64 * N + 2 <= 62 * NL, "the signed 62-bit limbs must hold every value in (-2 p, 2 p) for p < 2^(64 N)");
    // bound on the division steps for inputs below 2^(64 N): (49 d + 80) / 17 (Bernstein-Yang theorem 11.2), in batches of 62
    constexpr int MAX_BATCHES = ((49 * 64 * N + 80) / 17 + 61) / 62;
    S62<NL> p = to_s62<N, NL>(pw), f = p, g = to_s62<N, NL>(a), d, e;
#pragma unroll
    for (int i = 0; i < NL; i++) d.v[i] = e.v[i] = 0;
    e.v[0] = 1;
    // p^-1 mod 2^62 by Newton iteration (p odd): x <- x (2 - p x) doubles the correct low bits, 3 -> 96
    uint64_t p0 = (uint64_t)p.v[0], pinv = p0;
#pragma unroll
    for (int i = 0; i < 5; i++) pinv *= 2 - p0 * pinv;
    pinv &= (uint64_t)M62;
    int64_t eta = -1;
    for (int it = 0; it < MAX_BATCHES; it++) {
        int64_t nz = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) nz |= g.v[i];
        if (nz == 0) break;
        Trans t;
        eta = divsteps_62(eta, (uint64_t)f.v[0], (uint64_t)g.v[0], t);
        update_de<NL>(d, e, t, p, pinv);
        update_fg<NL>(f, g, t);
    }
    // g = 0 and f = +-gcd = +-1 (or f = p when a = 0, where d = 0 stays 0)
    normalize<NL>(d, f.v[NL - 1], p);
    from_s62<N, NL>(d, out);
}

// ---- the same inversion with its state (f, g, d, e: 4 NL limbs) in caller-provided memory ---------------------------------------
// inv<N> keeps ~4 NL 64-bit limbs + the modulus in registers: ~120 VGPRs for N = 4.  A kernel that calls it ONCE in a cold path (the
// pairings' digit chain: the loader wave inverts the divisors of a division round) still gets that register allocation for every wave -
// and its CU no longer has room for another kernel's waves beside it.  Here the limbs live in memory (LDS on the device: P is a pointer
// type into it, a plain int64_t* on the host) and are streamed through a few registers limb by limb; same steps, same result.
template <int NL, class P>
MI_INLINE void update_fg_mem(P f, P g, const Trans& t) {
    i128 cf = (i128)t.u * f[0] + (i128)t.v * g[0];
    i128 cg = (i128)t.q * f[0] + (i128)t.r * g[0];
    cf >>= 62;
    cg >>= 62;
#pragma unroll 1
    for (int i = 1; i < NL; i++) {
        int64_t fi = f[i], gi = g[i];
        cf += (i128)t.u * fi + (i128)t.v * gi;
        cg += (i128)t.q * fi + (i128)t.r * gi;
        f[i - 1] = (int64_t)cf & M62;
        g[i - 1] = (int64_t)cg & M62;
        cf >>= 62;
        cg >>= 62;
    }
    f[NL - 1] = (int64_t)cf;
    g[NL - 1] = (int64_t)cg;
}
template <int NL, class P, class PW>
MI_INLINE void update_de_mem(P d, P e, const Trans& t, PW pw, uint64_t pinv) {
    auto p_limb = [&](int i) -> int64_t {   // limb i of the modulus from its 64-bit words (to_s62 for one limb)
        int bit = 62 * i, k = bit / 64, sh = bit % 64;
        uint64_t lo = pw[k] >> sh;
        uint64_t hi = sh > 2 ? pw[k + 1] << (64 - sh) : 0;   // (pw has one word of padding above the modulus)
        return (int64_t)((lo | hi) & (uint64_t)M62);
    };
    int64_t sd = d[NL - 1] >> 63, se = e[NL - 1] >> 63;
    int64_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
    int64_t d0 = d[0], e0 = e[0];
    i128 cd = (i128)t.u * d0 + (i128)t.v * e0;
    i128 ce = (i128)t.q * d0 + (i128)t.r * e0;
    md -= (int64_t)((pinv * (uint64_t)cd + (uint64_t)md) & (uint64_t)M62);
    me -= (int64_t)((pinv * (uint64_t)ce + (uint64_t)me) & (uint64_t)M62);
    int64_t p0 = p_limb(0);
    cd += (i128)p0 * md;
    ce += (i128)p0 * me;
    cd >>= 62;
    ce >>= 62;
#pragma unroll 1
    for (int i = 1; i < NL; i++) {
        int64_t di = d[i], ei = e[i], pi = p_limb(i);
        cd += (i128)t.u * di + (i128)t.v * ei;
        ce += (i128)t.q * di + (i128)t.r * ei;
        cd += (i128)pi * md;
        ce += (i128)pi * me;
        d[i - 1] = (int64_t)cd & M62;
        e[i - 1] = (int64_t)ce & M62;
        cd >>= 62;
        ce >>= 62;
    }
    d[NL - 1] = (int64_t)cd;
    e[NL - 1] = (int64_t)ce;
}
// a^-1 mod p (0 for a = 0); a, out: N words; pw: N + 1 words in memory (the modulus and a zero word above it); f, g, d, e: NL limbs each
template <int N, class P, class PW>
MI_INLINE void inv_mem(const uint64_t* a, PW pw, uint64_t* out, P f, P g, P d, P e) {
    constexpr int NL = (64 * N + 61) / 62;
    static_assert(64 * N + 2 <= 62 * NL, "the signed 62-bit limbs must hold every value in (-2 p, 2 p) for p < 2^(64 N)");
    constexpr int MAX_BATCHES = ((49 * 64 * N + 80) / 17 + 61) / 62;
    uint64_t pl[N];
#pragma unroll
    for (int i = 0; i < N; i++) pl[i] = pw[i];
    {
        S62<NL> ps = to_s62<N, NL>(pl), as = to_s62<N, NL>(a);
#pragma unroll
        for (int i = 0; i < NL; i++) {
            f[i] = ps.v[i];
            g[i] = as.v[i];
            d[i] = 0;
            e[i] = i == 0 ? 1 : 0;
        }
    }
    uint64_t p0 = (uint64_t)f[0], pinv = p0;
#pragma unroll
    for (int i = 0; i < 5; i++) pinv *= 2 - p0 * pinv;
    pinv &= (uint64_t)M62;
    int64_t eta = -1;
#pragma unroll 1
    for (int it = 0; it < MAX_BATCHES; it++) {
        int64_t nz = 0;
#pragma unroll 1
        for (int i = 0; i < NL; i++) nz |= g[i];
        if (nz == 0) break;
        Trans t;
        eta = divsteps_62(eta, (uint64_t)f[0], (uint64_t)g[0], t);
        update_de_mem<NL, P, PW>(d, e, t, pw, pinv);
        update_fg_mem<NL, P>(f, g, t);
    }
    S62<NL> ds, ps = to_s62<N, NL>(pl);
#pragma unroll
    for (int i = 0; i < NL; i++) ds.v[i] = d[i];
    normalize<NL>(ds, f[NL - 1], ps);
    from_s62<N, NL>(ds, out);
}

}  // namespace modinv62
