// Fixed-width multi-word integers for the gfx950 witness engine (64-bit words, fully unrolled so
// every word lives in a VGPR).  These replace num-bigint on the reference's witness path
// (src/circuit/integer_chip.rs:472-474, :296-297, :524-529).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "modinv62.h"

typedef uint64_t u64;
typedef uint32_t u32;

#define WI_INLINE __device__ __forceinline__

template <int N>
struct Wd {
    u64 v[N];
};

// 64 x 64 -> 128 from four 32 x 32 -> 64 products (each one v_mad_u64_u32 / v_mul_{lo,hi}_u32 pair); the
// compiler's own lowering of `a * b` next to `__umul64hi(a, b)` repeats the partial products.
WI_INLINE void mul_wide64(u64 a, u64 b, u64& lo, u64& hi) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0, p01 = (u64)a0 * b1, p10 = (u64)a1 * b0, p11 = (u64)a1 * b1;
    u64 mid = (p00 >> 32) + (u32)p01 + (u32)p10;
    lo = (p00 & 0xffffffffull) | (mid << 32);
    hi = p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
}

template <int N>
WI_INLINE Wd<N> wd_zero() {
    Wd<N> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = 0;
    return r;
}
template <int N>
WI_INLINE Wd<N> wd_from_u64(u64 x) {
    Wd<N> r = wd_zero<N>();
    r.v[0] = x;
    return r;
}
template <int N>
WI_INLINE Wd<N> wd_load(const u64* p) {
    Wd<N> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = p[i];
    return r;
}
// resize (zero-extend or truncate)
template <int M, int N>
WI_INLINE Wd<M> wd_resize(const Wd<N>& a) {
    Wd<M> r;
#pragma unroll
    for (int i = 0; i < M; i++) r.v[i] = (i < N) ? a.v[i] : 0;
    return r;
}
template <int N>
WI_INLINE bool wd_is_zero(const Wd<N>& a) {
    u64 o = 0;
#pragma unroll
    for (int i = 0; i < N; i++) o |= a.v[i];
    return o == 0;
}
template <int N>
WI_INLINE bool wd_eq(const Wd<N>& a, const Wd<N>& b) {
    u64 o = 0;
#pragma unroll
    for (int i = 0; i < N; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}
// Carry chains.  hipcc turns neither `s < a` carries nor __builtin_addcll into v_addc_co_u32 chains (a 256-bit addition
// came out as 24 instructions: a 64-bit add, two 64-bit compares and a select per word); with the carry as an explicit
// SGPR-pair operand of the VOP3 forms it is one instruction per 32-bit limb, and the data dependency through that operand
// keeps the chain intact whatever the scheduler puts in between.
WI_INLINE u32 add_co32(u32 a, u32 b, u64& c) {
    u32 r;
    asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(r), "=s"(c) : "v"(a), "v"(b));
    return r;
}
WI_INLINE u32 addc_co32(u32 a, u32 b, u64& c) {
    u32 r;
    u64 co;
    asm("v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(r), "=s"(co) : "v"(a), "v"(b), "s"(c));
    c = co;
    return r;
}
WI_INLINE u32 sub_co32(u32 a, u32 b, u64& c) {
    u32 r;
    asm("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(r), "=s"(c) : "v"(a), "v"(b));
    return r;
}
WI_INLINE u32 subb_co32(u32 a, u32 b, u64& c) {
    u32 r;
    u64 co;
    asm("v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(r), "=s"(co) : "v"(a), "v"(b), "s"(c));
    c = co;
    return r;
}
WI_INLINE u32 carry_bit(u64 c) {   // this lane's bit of a carry mask as 0 / 1
    u32 r;
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(r) : "s"(c));
    return r;
}
WI_INLINE u64 pack64(u32 lo, u32 hi) { return (u64)lo | ((u64)hi << 32); }
// a >= b
template <int N>
WI_INLINE bool wd_geq(const Wd<N>& a, const Wd<N>& b) {
    u64 c;
    (void)sub_co32((u32)a.v[0], (u32)b.v[0], c);
    (void)subb_co32((u32)(a.v[0] >> 32), (u32)(b.v[0] >> 32), c);
#pragma unroll
    for (int i = 1; i < N; i++) {
        (void)subb_co32((u32)a.v[i], (u32)b.v[i], c);
        (void)subb_co32((u32)(a.v[i] >> 32), (u32)(b.v[i] >> 32), c);
    }
    return carry_bit(c) == 0;
}
template <int N>
WI_INLINE Wd<N> wd_add_c(const Wd<N>& a, const Wd<N>& b, u64& carry_out) {
    Wd<N> r;
    u64 c;
    u32 lo = add_co32((u32)a.v[0], (u32)b.v[0], c);
    u32 hi = addc_co32((u32)(a.v[0] >> 32), (u32)(b.v[0] >> 32), c);
    r.v[0] = pack64(lo, hi);
#pragma unroll
    for (int i = 1; i < N; i++) {
        lo = addc_co32((u32)a.v[i], (u32)b.v[i], c);
        hi = addc_co32((u32)(a.v[i] >> 32), (u32)(b.v[i] >> 32), c);
        r.v[i] = pack64(lo, hi);
    }
    carry_out = carry_bit(c);
    return r;
}
template <int N>
WI_INLINE Wd<N> wd_add(const Wd<N>& a, const Wd<N>& b) {
    Wd<N> r;
    u64 c;
    u32 lo = add_co32((u32)a.v[0], (u32)b.v[0], c);
    u32 hi = addc_co32((u32)(a.v[0] >> 32), (u32)(b.v[0] >> 32), c);
    r.v[0] = pack64(lo, hi);
#pragma unroll
    for (int i = 1; i < N; i++) {
        lo = addc_co32((u32)a.v[i], (u32)b.v[i], c);
        hi = addc_co32((u32)(a.v[i] >> 32), (u32)(b.v[i] >> 32), c);
        r.v[i] = pack64(lo, hi);
    }
    return r;
}
template <int N>
WI_INLINE Wd<N> wd_sub(const Wd<N>& a, const Wd<N>& b) {
    Wd<N> r;
    u64 c;
    u32 lo = sub_co32((u32)a.v[0], (u32)b.v[0], c);
    u32 hi = subb_co32((u32)(a.v[0] >> 32), (u32)(b.v[0] >> 32), c);
    r.v[0] = pack64(lo, hi);
#pragma unroll
    for (int i = 1; i < N; i++) {
        lo = subb_co32((u32)a.v[i], (u32)b.v[i], c);
        hi = subb_co32((u32)(a.v[i] >> 32), (u32)(b.v[i] >> 32), c);
        r.v[i] = pack64(lo, hi);
    }
    return r;
}
// acc += a * m for a 32-bit multiplier: one v_mad_u64_u32 and one 64-bit add per 32-bit limb of a
template <int N>
WI_INLINE void wd_mac_small(Wd<N + 1>& acc, const Wd<N>& a, u32 m) {
    u32 carry = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        u64 lo = (u64)(u32)a.v[i] * m + (u32)acc.v[i] + carry;
        u64 hi = (u64)(u32)(a.v[i] >> 32) * m + (u32)(acc.v[i] >> 32) + (u32)(lo >> 32);
        acc.v[i] = pack64((u32)lo, (u32)hi);
        carry = (u32)(hi >> 32);
    }
    acc.v[N] += carry;
}
template <int N>
WI_INLINE Wd<N + 1> wd_mul_small(const Wd<N>& a, u32 m) {
    Wd<N + 1> r;
#pragma unroll
    for (int i = 0; i <= N; i++) r.v[i] = 0;
    wd_mac_small<N>(r, a, m);
    return r;
}
template <int N>
WI_INLINE Wd<N> wd_neg(const Wd<N>& a) {
    return wd_sub<N>(wd_zero<N>(), a);
}
template <int N>
WI_INLINE bool wd_is_neg(const Wd<N>& a) {  // two's complement sign
    return (a.v[N - 1] >> 63) != 0;
}
// 32-bit limb view of a word array
template <int N>
WI_INLINE u32 limb32(const Wd<N>& a, int k) {
    return (k & 1) ? (u32)(a.v[k >> 1] >> 32) : (u32)a.v[k >> 1];
}
// Column accumulator of the product-scanning multiplications: 96 bits (lo64 + a carry word); one partial product is
// one v_mad_u64_u32 into the low 64 bits + one add-with-carry into the third word.  (The operand-scanning form
// `s = a * b + t[j] + c` costs the compiler as many instructions again for zero-extending and re-pairing its 32-bit
// carries.)
struct WdAcc {
    u64 lo;
    u32 hi;
};
WI_INLINE void wd_mac(WdAcc& A, u32 a, u32 b) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(A.lo), "+v"(A.hi) : "v"(a), "v"(b) : "vcc");
}
WI_INLINE u32 wd_acc_shift(WdAcc& A) {   // take the column's 32 bits, move on to the next column
    u32 r = (u32)A.lo;
    A.lo = (A.lo >> 32) | ((u64)A.hi << 32);
    A.hi = 0;
    return r;
}
// full product: product scanning over 32-bit limbs.  LAE / LBE: the operands' significant 32-bit limbs when their bit
// bounds are tighter than the word counts (a 260-bit composed operand lives in 5 words but has 9 limbs: 81 partial
// products instead of 100) - limbs at and above them must be zero.
template <int NA, int NB, int LAE = 2 * NA, int LBE = 2 * NB>
WI_INLINE Wd<NA + NB> wd_mul(const Wd<NA>& a, const Wd<NB>& b) {
    constexpr int LA = 2 * NA, LB = 2 * NB;
    static_assert(LAE <= LA && LBE <= LB, "effective limbs exceed the operand");
    u32 t[LA + LB];
    WdAcc A{0, 0};
#pragma unroll
    for (int k = 0; k < LA + LB - 1; k++) {
#pragma unroll
        for (int i = 0; i < LAE; i++)
            if (i <= k && k - i < LBE) wd_mac(A, limb32<NA>(a, i), limb32<NB>(b, k - i));
        t[k] = wd_acc_shift(A);
    }
    t[LA + LB - 1] = (u32)A.lo;
    Wd<NA + NB> r;
#pragma unroll
    for (int i = 0; i < NA + NB; i++) r.v[i] = (u64)t[2 * i] | ((u64)t[2 * i + 1] << 32);
    return r;
}
// low NR words of the product
template <int NR, int NA, int NB, int LAE = 2 * NA, int LBE = 2 * NB>
WI_INLINE Wd<NR> wd_mul_lo(const Wd<NA>& a, const Wd<NB>& b) {
    constexpr int LA = 2 * NA, LB = 2 * NB, LR = 2 * NR;
    static_assert(LAE <= LA && LBE <= LB, "effective limbs exceed the operand");
    u32 t[LR];
    WdAcc A{0, 0};
#pragma unroll
    for (int k = 0; k < LR; k++) {
#pragma unroll
        for (int i = 0; i < LAE; i++)
            if (i <= k && k - i < LBE) wd_mac(A, limb32<NA>(a, i), limb32<NB>(b, k - i));
        t[k] = wd_acc_shift(A);
    }
    Wd<NR> r;
#pragma unroll
    for (int i = 0; i < NR; i++) r.v[i] = (u64)t[2 * i] | ((u64)t[2 * i + 1] << 32);
    return r;
}
// (a >> SH) truncated / zero-extended to M words; SH compile-time
template <int M, int SH, int N>
WI_INLINE Wd<M> wd_shr(const Wd<N>& a) {
    Wd<M> r;
    constexpr int ws = SH / 64, bs = SH % 64;
#pragma unroll
    for (int i = 0; i < M; i++) {
        u64 lo = (i + ws < N) ? a.v[i + ws] : 0;
        u64 hi = (i + ws + 1 < N) ? a.v[i + ws + 1] : 0;
        r.v[i] = bs ? ((lo >> bs) | (hi << (64 - bs))) : lo;
    }
    return r;
}
// (a << SH) into M words; SH compile-time
template <int M, int SH, int N>
WI_INLINE Wd<M> wd_shl(const Wd<N>& a) {
    Wd<M> r;
    constexpr int ws = SH / 64, bs = SH % 64;
#pragma unroll
    for (int i = 0; i < M; i++) {
        u64 lo = (i - ws >= 0 && i - ws < N) ? a.v[i - ws] : 0;
        u64 pl = (i - ws - 1 >= 0 && i - ws - 1 < N) ? a.v[i - ws - 1] : 0;
        r.v[i] = bs ? ((lo << bs) | (pl >> (64 - bs))) : lo;
    }
    return r;
}
// keep the low BITS bits
template <int BITS, int N>
WI_INLINE Wd<N> wd_mask(const Wd<N>& a) {
    Wd<N> r;
#pragma unroll
    for (int i = 0; i < N; i++) {
        if ((i + 1) * 64 <= BITS)
            r.v[i] = a.v[i];
        else if (i * 64 >= BITS)
            r.v[i] = 0;
        else
            r.v[i] = a.v[i] & ((1ull << (BITS - i * 64)) - 1);
    }
    return r;
}
template <int N>
WI_INLINE Wd<N> wd_shr1(const Wd<N>& a) {
    Wd<N> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = (a.v[i] >> 1) | ((i + 1 < N) ? (a.v[i + 1] << 63) : 0);
    return r;
}
template <int N>
WI_INLINE Wd<N> wd_select(bool c, const Wd<N>& a, const Wd<N>& b) {
    Wd<N> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}

// Exact floor division by Barrett reduction.
//   X < 2^S, modulus m with bit length K, mu = floor(2^S / m) (S-K+1 bits).
//   returns q = floor(X / m) (QW words) and r = X mod m (MW words).
// q3 = floor(floor(X / 2^(K-1)) * mu / 2^(S-K+1)) satisfies q-2 <= q3 <= q; two conditional
// subtractions finish.
template <int S, int K, int XW, int MW, int QW>
WI_INLINE void wd_barrett_divrem(const Wd<XW>& X, const Wd<MW>& m, const Wd<QW>& mu, Wd<QW>& q, Wd<MW>& r) {
    Wd<QW> q1 = wd_shr<QW, K - 1>(X);
    // q1, mu and q3 are below 2^(S-K+1), m below 2^K: their significant limbs, not their word counts, size the products
    constexpr int QL = (S - K + 1 + 31) / 32 < 2 * QW ? (S - K + 1 + 31) / 32 : 2 * QW;
    constexpr int ML = (K + 31) / 32 < 2 * MW ? (K + 31) / 32 : 2 * MW;
    Wd<2 * QW> q2 = wd_mul<QW, QW, QL, QL>(q1, mu);
    Wd<QW> q3 = wd_shr<QW, S - K + 1>(q2);
    // r = X - q3*m on MW+1 words (true value < 3m)
    Wd<MW + 1> qm = wd_mul_lo<MW + 1, QW, MW, QL, ML>(q3, m);
    Wd<MW + 1> rr = wd_sub<MW + 1>(wd_resize<MW + 1>(X), qm);
    Wd<MW + 1> me = wd_resize<MW + 1>(m);
    Wd<QW> one = wd_from_u64<QW>(1);
#pragma unroll
    for (int it = 0; it < 2; it++) {
        bool ge = wd_geq<MW + 1>(rr, me);
        rr = wd_select<MW + 1>(ge, wd_sub<MW + 1>(rr, me), rr);
        q3 = wd_select<QW>(ge, wd_add<QW>(q3, one), q3);
    }
    q = q3;
    r = wd_resize<MW>(rr);
}

// Modular inverse by the binary extended Euclidean algorithm (p odd, 0 < a < p).
// Returns 0 for a == 0 (Field::invert() -> None, mapped to zero by the callers exactly as the
// reference does: base_chip.rs:301, integer_chip.rs:524-527).
template <int N>
WI_INLINE Wd<N> wd_inv_mod_euclid(const Wd<N>& a, const Wd<N>& p);
// Division steps in 62-bit batches (modinv62.h): the same instruction sequence in every lane, ~10 x cheaper than the
// binary extended Euclid below (which stays for widths modinv62 is not instantiated for).  A called function: its
// ~25 k instructions exist once per kernel, not once per call site.
template <int N>
__device__ __attribute__((noinline)) Wd<N> wd_inv_mod_divsteps(Wd<N> a, Wd<N> p) {
    Wd<N> r;
    modinv62::inv<N>(a.v, p.v, r.v);
    return r;
}
template <int N>
WI_INLINE Wd<N> wd_inv_mod(const Wd<N>& a, const Wd<N>& p) {
    if constexpr (N == 4 || N == 6) return wd_inv_mod_divsteps<N>(a, p);
    else return wd_inv_mod_euclid<N>(a, p);
}
template <int N>
WI_INLINE Wd<N> wd_inv_mod_euclid(const Wd<N>& a, const Wd<N>& p) {
    if (wd_is_zero<N>(a)) return wd_zero<N>();
    Wd<N> u = a, v = p;
    Wd<N> x1 = wd_from_u64<N>(1), x2 = wd_zero<N>();
    Wd<N> one = wd_from_u64<N>(1);
    // invariants: x1*a == u, x2*a == v (mod p); x1, x2 in [0, p)
    while (!wd_eq<N>(u, one) && !wd_eq<N>(v, one)) {
        while ((u.v[0] & 1) == 0) {
            u = wd_shr1<N>(u);
            if (x1.v[0] & 1) {
                u64 c;
                Wd<N> t = wd_add_c<N>(x1, p, c);
                x1 = wd_shr1<N>(t);
                x1.v[N - 1] |= c << 63;
            } else {
                x1 = wd_shr1<N>(x1);
            }
        }
        while ((v.v[0] & 1) == 0) {
            v = wd_shr1<N>(v);
            if (x2.v[0] & 1) {
                u64 c;
                Wd<N> t = wd_add_c<N>(x2, p, c);
                x2 = wd_shr1<N>(t);
                x2.v[N - 1] |= c << 63;
            } else {
                x2 = wd_shr1<N>(x2);
            }
        }
        if (wd_geq<N>(u, v)) {
            u = wd_sub<N>(u, v);
            x1 = wd_geq<N>(x1, x2) ? wd_sub<N>(x1, x2) : wd_sub<N>(wd_add<N>(x1, p), x2);
        } else {
            v = wd_sub<N>(v, u);
            x2 = wd_geq<N>(x2, x1) ? wd_sub<N>(x2, x1) : wd_sub<N>(wd_add<N>(x2, p), x1);
        }
    }
    return wd_eq<N>(u, one) ? x1 : x2;
}
