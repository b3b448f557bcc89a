// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Not part of the product path.
//
// L4 tower arithmetic over the integer chip, restating src/circuit/fq12.rs:10-459
// (Fq2ChipOps, Fq6ChipOps, Fq12ChipOps + the curve-specific hook traits) and the tuple aliases
// src/assign.rs:164-167.  Rust evaluates tuple fields left to right; every sequence below is
// written as explicit statements in that order.
#pragma once
#include "ecc_chip.hpp"

namespace h2o {

struct AssignedFq2 {
    AssignedInteger c0, c1;
};
struct AssignedFq6 {
    AssignedFq2 c0, c1, c2;
};
struct AssignedFq12 {
    AssignedFq6 c0, c1;
};
struct Fq2Const {
    BigUint c0, c1;
};

// Curve-specific hooks (fq12.rs:10-22) are virtual; implemented in pairing.hpp.
struct TowerOps {
    IntegerContext& ic;
    explicit TowerOps(IntegerContext& i) : ic(i) {}
    virtual ~TowerOps() {}

    virtual AssignedFq2 fq2_mul_by_nonresidue(const AssignedFq2& a) = 0;
    virtual AssignedFq2 fq2_frobenius_map(const AssignedFq2& x, size_t power) = 0;
    virtual AssignedFq6 fq6_frobenius_map(const AssignedFq6& x, size_t power) = 0;
    virtual AssignedFq12 fq12_frobenius_map(const AssignedFq12& x, size_t power) = 0;
    // identical for bn256 and bls12_381 (bn256_pairing_chip.rs:61-63, bls12_381_pairing_chip.rs:47-49)
    AssignedFq6 fq6_mul_by_nonresidue(const AssignedFq6& a) {
        AssignedFq2 t = fq2_mul_by_nonresidue(a.c2);
        return AssignedFq6{t, a.c0, a.c1};
    }

    // ---- Fq2ChipOps (fq12.rs:24-104) ----
    AssignedFq2 fq2_reduce(const AssignedFq2& x) {
        AssignedInteger a = ic.reduce(x.c0);
        AssignedInteger b = ic.reduce(x.c1);
        return AssignedFq2{a, b};
    }
    void fq2_assert_equal(const AssignedFq2& x, const AssignedFq2& y) {
        ic.assert_int_equal(x.c0, y.c0);
        ic.assert_int_equal(x.c1, y.c1);
    }
    AssignedFq2 fq2_assign_zero() {
        AssignedInteger z = ic.assign_int_constant(BigUint(0));
        return AssignedFq2{z, z};
    }
    AssignedFq2 fq2_assign_one() {
        AssignedInteger a = ic.assign_int_constant(BigUint(1));
        AssignedInteger b = ic.assign_int_constant(BigUint(0));
        return AssignedFq2{a, b};
    }
    AssignedFq2 fq2_assign_constant(const Fq2Const& c) {
        AssignedInteger a = ic.assign_int_constant(c.c0);
        AssignedInteger b = ic.assign_int_constant(c.c1);
        return AssignedFq2{a, b};
    }
    AssignedFq2 fq2_add(const AssignedFq2& a, const AssignedFq2& b) {
        AssignedInteger x = ic.int_add(a.c0, b.c0);
        AssignedInteger y = ic.int_add(a.c1, b.c1);
        return AssignedFq2{x, y};
    }
    AssignedFq2 fq2_mul(const AssignedFq2& a, const AssignedFq2& b) {
        AssignedInteger ab00 = ic.int_mul(a.c0, b.c0);
        AssignedInteger ab11 = ic.int_mul(a.c1, b.c1);
        AssignedInteger c0 = ic.int_sub(ab00, ab11);
        AssignedInteger a01 = ic.int_add(a.c0, a.c1);
        AssignedInteger b01 = ic.int_add(b.c0, b.c1);
        AssignedInteger c1 = ic.int_mul(a01, b01);
        c1 = ic.int_sub(c1, ab00);
        c1 = ic.int_sub(c1, ab11);
        return AssignedFq2{c0, c1};
    }
    AssignedFq2 fq2_sub(const AssignedFq2& a, const AssignedFq2& b) {
        AssignedInteger x = ic.int_sub(a.c0, b.c0);
        AssignedInteger y = ic.int_sub(a.c1, b.c1);
        return AssignedFq2{x, y};
    }
    AssignedFq2 fq2_double(const AssignedFq2& a) {
        AssignedInteger x = ic.int_add(a.c0, a.c0);
        AssignedInteger y = ic.int_add(a.c1, a.c1);
        return AssignedFq2{x, y};
    }
    AssignedFq2 fq2_square(const AssignedFq2& a) { return fq2_mul(a, a); }
    AssignedFq2 fq2_neg(const AssignedFq2& a) {
        AssignedInteger x = ic.int_neg(a.c0);
        AssignedInteger y = ic.int_neg(a.c1);
        return AssignedFq2{x, y};
    }
    AssignedFq2 fq2_conjugate(const AssignedFq2& a) {
        AssignedInteger y = ic.int_neg(a.c1);
        return AssignedFq2{a.c0, y};
    }
    AssignedFq2 fq2_unsafe_invert(const AssignedFq2& x) {
        AssignedInteger t0 = ic.int_square(x.c0);
        AssignedInteger t1 = ic.int_square(x.c1);
        t0 = ic.int_add(t0, t1);
        AssignedInteger t = ic.int_unsafe_invert(t0);
        AssignedInteger c0 = ic.int_mul(x.c0, t);
        AssignedInteger c1 = ic.int_mul(x.c1, t);
        c1 = ic.int_neg(c1);
        return AssignedFq2{c0, c1};
    }

    // ---- Fq6ChipOps (fq12.rs:106-287) ----
    AssignedFq6 fq6_reduce(const AssignedFq6& x) {
        AssignedFq2 a = fq2_reduce(x.c0);
        AssignedFq2 b = fq2_reduce(x.c1);
        AssignedFq2 c = fq2_reduce(x.c2);
        return AssignedFq6{a, b, c};
    }
    void fq6_assert_equal(const AssignedFq6& x, const AssignedFq6& y) {
        fq2_assert_equal(x.c0, y.c0);
        fq2_assert_equal(x.c1, y.c1);
        fq2_assert_equal(x.c2, y.c2);
    }
    AssignedFq6 fq6_assign_zero() {
        AssignedFq2 z = fq2_assign_zero();
        return AssignedFq6{z, z, z};
    }
    AssignedFq6 fq6_assign_one() {
        AssignedFq2 o = fq2_assign_one();
        AssignedFq2 z = fq2_assign_zero();
        return AssignedFq6{o, z, z};
    }
    AssignedFq6 fq6_add(const AssignedFq6& a, const AssignedFq6& b) {
        AssignedFq2 x = fq2_add(a.c0, b.c0);
        AssignedFq2 y = fq2_add(a.c1, b.c1);
        AssignedFq2 z = fq2_add(a.c2, b.c2);
        return AssignedFq6{x, y, z};
    }
    AssignedFq6 fq6_mul(const AssignedFq6& a, const AssignedFq6& b) {
        AssignedFq2 ab00 = fq2_mul(a.c0, b.c0);
        AssignedFq2 ab11 = fq2_mul(a.c1, b.c1);
        AssignedFq2 ab22 = fq2_mul(a.c2, b.c2);
        AssignedFq2 c0, c1, c2;
        {
            AssignedFq2 b12 = fq2_add(b.c1, b.c2);
            AssignedFq2 a12 = fq2_add(a.c1, a.c2);
            AssignedFq2 t = fq2_mul(a12, b12);
            t = fq2_sub(t, ab11);
            t = fq2_sub(t, ab22);
            t = fq2_mul_by_nonresidue(t);
            c0 = fq2_add(t, ab00);
        }
        {
            AssignedFq2 b01 = fq2_add(b.c0, b.c1);
            AssignedFq2 a01 = fq2_add(a.c0, a.c1);
            AssignedFq2 t = fq2_mul(a01, b01);
            t = fq2_sub(t, ab00);
            t = fq2_sub(t, ab11);
            AssignedFq2 ab22n = fq2_mul_by_nonresidue(ab22);
            c1 = fq2_add(t, ab22n);
        }
        {
            AssignedFq2 b02 = fq2_add(b.c0, b.c2);
            AssignedFq2 a02 = fq2_add(a.c0, a.c2);
            AssignedFq2 t = fq2_mul(a02, b02);
            t = fq2_sub(t, ab00);
            t = fq2_add(t, ab11);
            c2 = fq2_sub(t, ab22);
        }
        return AssignedFq6{c0, c1, c2};
    }
    AssignedFq6 fq6_sub(const AssignedFq6& a, const AssignedFq6& b) {
        AssignedFq2 x = fq2_sub(a.c0, b.c0);
        AssignedFq2 y = fq2_sub(a.c1, b.c1);
        AssignedFq2 z = fq2_sub(a.c2, b.c2);
        return AssignedFq6{x, y, z};
    }
    AssignedFq6 fq6_double(const AssignedFq6& a) {
        AssignedFq2 x = fq2_double(a.c0);
        AssignedFq2 y = fq2_double(a.c1);
        AssignedFq2 z = fq2_double(a.c2);
        return AssignedFq6{x, y, z};
    }
    AssignedFq6 fq6_square(const AssignedFq6& a) { return fq6_mul(a, a); }
    AssignedFq6 fq6_neg(const AssignedFq6& a) {
        AssignedFq2 x = fq2_neg(a.c0);
        AssignedFq2 y = fq2_neg(a.c1);
        AssignedFq2 z = fq2_neg(a.c2);
        return AssignedFq6{x, y, z};
    }
    AssignedFq6 fq6_mul_by_1(const AssignedFq6& a, const AssignedFq2& b1) {
        AssignedFq2 ab11 = fq2_mul(a.c1, b1);
        AssignedFq2 c0, c1;
        {
            AssignedFq2 a12 = fq2_add(a.c1, a.c2);
            AssignedFq2 t = fq2_mul(a12, b1);
            t = fq2_sub(t, ab11);
            c0 = fq2_mul_by_nonresidue(t);
        }
        {
            AssignedFq2 a01 = fq2_add(a.c0, a.c1);
            AssignedFq2 t = fq2_mul(a01, b1);
            c1 = fq2_sub(t, ab11);
        }
        return AssignedFq6{c0, c1, ab11};
    }
    AssignedFq6 fq6_mul_by_01(const AssignedFq6& a, const AssignedFq2& b0, const AssignedFq2& b1) {
        AssignedFq2 ab00 = fq2_mul(a.c0, b0);
        AssignedFq2 ab11 = fq2_mul(a.c1, b1);
        AssignedFq2 c0, c1, c2;
        {
            AssignedFq2 a12 = fq2_add(a.c1, a.c2);
            AssignedFq2 t = fq2_mul(a12, b1);
            t = fq2_sub(t, ab11);
            t = fq2_mul_by_nonresidue(t);
            c0 = fq2_add(t, ab00);
        }
        {
            AssignedFq2 b01 = fq2_add(b0, b1);
            AssignedFq2 a01 = fq2_add(a.c0, a.c1);
            AssignedFq2 t = fq2_mul(a01, b01);
            t = fq2_sub(t, ab00);
            c1 = fq2_sub(t, ab11);
        }
        {
            AssignedFq2 a02 = fq2_add(a.c0, a.c2);
            AssignedFq2 t = fq2_mul(a02, b0);
            t = fq2_sub(t, ab00);
            c2 = fq2_add(t, ab11);
        }
        return AssignedFq6{c0, c1, c2};
    }
    AssignedFq6 fq6_unsafe_invert(const AssignedFq6& x) {
        AssignedFq2 c0 = fq2_mul_by_nonresidue(x.c2);
        c0 = fq2_mul(c0, x.c1);
        c0 = fq2_neg(c0);
        AssignedFq2 x0s = fq2_square(x.c0);
        c0 = fq2_add(c0, x0s);

        AssignedFq2 c1 = fq2_square(x.c2);
        c1 = fq2_mul_by_nonresidue(c1);
        AssignedFq2 x01 = fq2_mul(x.c0, x.c1);
        c1 = fq2_sub(c1, x01);

        AssignedFq2 c2 = fq2_square(x.c1);
        AssignedFq2 x02 = fq2_mul(x.c0, x.c2);
        c2 = fq2_sub(c2, x02);

        AssignedFq2 c0x0 = fq2_mul(c0, x.c0);
        AssignedFq2 c1x2 = fq2_mul(c1, x.c2);
        AssignedFq2 c2x1 = fq2_mul(c2, x.c1);
        AssignedFq2 t = fq2_add(c1x2, c2x1);
        t = fq2_mul_by_nonresidue(t);
        t = fq2_add(t, c0x0);
        t = fq2_unsafe_invert(t);

        AssignedFq2 r0 = fq2_mul(t, c0);
        AssignedFq2 r1 = fq2_mul(t, c1);
        AssignedFq2 r2 = fq2_mul(t, c2);
        return AssignedFq6{r0, r1, r2};
    }
    AssignedFq6 fq6_assign_constant(const Fq2Const& a, const Fq2Const& b, const Fq2Const& c) {
        AssignedFq2 x = fq2_assign_constant(a);
        AssignedFq2 y = fq2_assign_constant(b);
        AssignedFq2 z = fq2_assign_constant(c);
        return AssignedFq6{x, y, z};
    }

    // ---- Fq12ChipOps (fq12.rs:289-459) ----
    AssignedFq12 fq12_reduce(const AssignedFq12& x) {
        AssignedFq6 a = fq6_reduce(x.c0);
        AssignedFq6 b = fq6_reduce(x.c1);
        return AssignedFq12{a, b};
    }
    void fq12_assert_one(const AssignedFq12& x) {
        AssignedFq12 one = fq12_assign_one();
        fq12_assert_eq(x, one);
    }
    void fq12_assert_eq(const AssignedFq12& x, const AssignedFq12& y) {
        fq6_assert_equal(x.c0, y.c0);
        fq6_assert_equal(x.c1, y.c1);
    }
    AssignedFq12 fq12_assign_zero() {
        AssignedFq6 z = fq6_assign_zero();
        return AssignedFq12{z, z};
    }
    AssignedFq12 fq12_assign_one() {
        AssignedFq6 o = fq6_assign_one();
        AssignedFq6 z = fq6_assign_zero();
        return AssignedFq12{o, z};
    }
    AssignedFq12 fq12_add(const AssignedFq12& a, const AssignedFq12& b) {
        AssignedFq6 x = fq6_add(a.c0, b.c0);
        AssignedFq6 y = fq6_add(a.c1, b.c1);
        return AssignedFq12{x, y};
    }
    AssignedFq12 fq12_mul(const AssignedFq12& a, const AssignedFq12& b) {
        AssignedFq6 ab00 = fq6_mul(a.c0, b.c0);
        AssignedFq6 ab11 = fq6_mul(a.c1, b.c1);
        AssignedFq6 a01 = fq6_add(a.c0, a.c1);
        AssignedFq6 b01 = fq6_add(b.c0, b.c1);
        AssignedFq6 c1 = fq6_mul(a01, b01);
        c1 = fq6_sub(c1, ab00);
        c1 = fq6_sub(c1, ab11);
        AssignedFq6 ab11n = fq6_mul_by_nonresidue(ab11);
        AssignedFq6 c0 = fq6_add(ab00, ab11n);
        return AssignedFq12{c0, c1};
    }
    AssignedFq12 fq12_sub(const AssignedFq12& a, const AssignedFq12& b) {
        AssignedFq6 x = fq6_sub(a.c0, b.c0);
        AssignedFq6 y = fq6_sub(a.c1, b.c1);
        return AssignedFq12{x, y};
    }
    AssignedFq12 fq12_double(const AssignedFq12& a) {
        AssignedFq6 x = fq6_double(a.c0);
        AssignedFq6 y = fq6_double(a.c1);
        return AssignedFq12{x, y};
    }
    AssignedFq12 fq12_square(const AssignedFq12& a) { return fq12_mul(a, a); }
    AssignedFq12 fq12_neg(const AssignedFq12& a) {
        AssignedFq6 x = fq6_neg(a.c0);
        AssignedFq6 y = fq6_neg(a.c1);
        return AssignedFq12{x, y};
    }
    AssignedFq12 fq12_conjugate(const AssignedFq12& x) {
        AssignedFq6 y = fq6_neg(x.c1);
        return AssignedFq12{x.c0, y};
    }
    AssignedFq12 fq12_mul_by_014(const AssignedFq12& x, const AssignedFq2& c0, const AssignedFq2& c1,
                                 const AssignedFq2& c4) {
        AssignedFq6 t0 = fq6_mul_by_01(x.c0, c0, c1);
        AssignedFq6 t1 = fq6_mul_by_1(x.c1, c4);
        AssignedFq2 o = fq2_add(c1, c4);
        AssignedFq6 x0 = fq6_mul_by_nonresidue(t1);
        x0 = fq6_add(x0, t0);
        AssignedFq6 x1 = fq6_add(x.c0, x.c1);
        x1 = fq6_mul_by_01(x1, c0, o);
        x1 = fq6_sub(x1, t0);
        x1 = fq6_sub(x1, t1);
        return AssignedFq12{x0, x1};
    }
    AssignedFq12 fq12_mul_by_034(const AssignedFq12& x, const AssignedFq2& c0, const AssignedFq2& c3,
                                 const AssignedFq2& c4) {
        AssignedFq2 t00 = fq2_mul(x.c0.c0, c0);
        AssignedFq2 t01 = fq2_mul(x.c0.c1, c0);
        AssignedFq2 t02 = fq2_mul(x.c0.c2, c0);
        AssignedFq6 t0{t00, t01, t02};
        AssignedFq6 t1 = fq6_mul_by_01(x.c1, c3, c4);
        AssignedFq6 t2 = fq6_add(x.c0, x.c1);
        AssignedFq2 o = fq2_add(c0, c3);
        t2 = fq6_mul_by_01(t2, o, c4);
        t2 = fq6_sub(t2, t0);
        AssignedFq6 x1 = fq6_sub(t2, t1);
        t1 = fq6_mul_by_nonresidue(t1);
        AssignedFq6 x0 = fq6_add(t0, t1);
        return AssignedFq12{x0, x1};
    }
    void fp4_square(AssignedFq2& c0, AssignedFq2& c1, const AssignedFq2& a0, const AssignedFq2& a1) {
        AssignedFq2 t0 = fq2_square(a0);
        AssignedFq2 t1 = fq2_square(a1);
        AssignedFq2 t2 = fq2_mul_by_nonresidue(t1);
        c0 = fq2_add(t2, t0);
        t2 = fq2_add(a0, a1);
        t2 = fq2_square(t2);
        t2 = fq2_sub(t2, t0);
        c1 = fq2_sub(t2, t1);
    }
    AssignedFq12 fq12_cyclotomic_square(const AssignedFq12& x) {
        AssignedFq2 zero = fq2_assign_zero();
        AssignedFq2 t3 = zero, t4 = zero, t5 = zero, t6 = zero;
        // copies: fp4_square's outputs alias nothing in x because Rust passes &x fields by value-borrow
        AssignedFq2 x00 = x.c0.c0, x01 = x.c0.c1, x02 = x.c0.c2, x10 = x.c1.c0, x11 = x.c1.c1, x12 = x.c1.c2;
        fp4_square(t3, t4, x00, x11);
        AssignedFq2 t2 = fq2_sub(t3, x00);
        t2 = fq2_double(t2);
        AssignedFq2 c00 = fq2_add(t2, t3);

        t2 = fq2_add(t4, x11);
        t2 = fq2_double(t2);
        AssignedFq2 c11 = fq2_add(t2, t4);

        fp4_square(t3, t4, x10, x02);
        fp4_square(t5, t6, x01, x12);

        t2 = fq2_sub(t3, x01);
        t2 = fq2_double(t2);
        AssignedFq2 c01 = fq2_add(t2, t3);
        t2 = fq2_add(t4, x12);
        t2 = fq2_double(t2);
        AssignedFq2 c12 = fq2_add(t2, t4);
        t3 = t6;
        t3 = fq2_mul_by_nonresidue(t3);
        t2 = fq2_add(t3, x10);
        t2 = fq2_double(t2);
        AssignedFq2 c10 = fq2_add(t2, t3);
        t2 = fq2_sub(t5, x02);
        t2 = fq2_double(t2);
        AssignedFq2 c02 = fq2_add(t2, t5);
        return AssignedFq12{AssignedFq6{c00, c01, c02}, AssignedFq6{c10, c11, c12}};
    }
    AssignedFq12 fq12_unsafe_invert(const AssignedFq12& x) {
        AssignedFq6 x0s = fq6_square(x.c0);
        AssignedFq6 x1s = fq6_square(x.c1);
        AssignedFq6 t = fq6_mul_by_nonresidue(x1s);
        t = fq6_sub(x0s, t);
        t = fq6_unsafe_invert(t);
        AssignedFq6 c0 = fq6_mul(t, x.c0);
        AssignedFq6 c1 = fq6_mul(t, x.c1);
        c1 = fq6_neg(c1);
        return AssignedFq12{c0, c1};
    }
};

}  // namespace h2o
