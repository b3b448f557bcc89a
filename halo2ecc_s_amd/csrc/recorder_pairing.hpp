// Recording implementation of the reference's tower + pairing surface (same names, same call order):
//   Fq2/Fq6/Fq12ChipOps             src/circuit/fq12.rs:24-459
//   PairingChipOps                  src/circuit/pairing_chip.rs:10-177
//   bn256 hooks / Miller / final    src/circuit/bn256_pairing_chip.rs:29-350
//   bls12_381 hooks / Miller / final src/circuit/bls12_381_pairing_chip.rs:29-287
// Everything here is pure composition of IntegerChipOps calls on the Recorder; Rust's left-to-right
// tuple evaluation is spelled out as statement order.
#pragma once
#include <array>
#include "recorder_ecc.hpp"
#include "pairing_constants.hpp"

namespace h2e {

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
    HBig c0, c1;
};

// Curve-specific hooks (fq12.rs:10-22) are virtual; implemented in pairing.hpp.
struct TowerOps {
    Recorder& ic;
    explicit TowerOps(Recorder& i) : ic(i) {}
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
        AssignedInteger z = ic.assign_int_constant(HBig(0));
        return AssignedFq2{z, z};
    }
    AssignedFq2 fq2_assign_one() {
        AssignedInteger a = ic.assign_int_constant(HBig(1));
        AssignedInteger b = ic.assign_int_constant(HBig(0));
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



struct AssignedG2Affine {  // assign.rs:171-192
    AssignedFq2 x, y;
    AssignedCondition z;
};
struct AssignedG2 {  // assign.rs:194-214
    AssignedFq2 x, y, z;
};
typedef std::array<AssignedFq2, 3> G2Coeffs;
struct AssignedG2Prepared {  // assign.rs:216-229
    std::vector<G2Coeffs> coeffs;
};

struct PairingOps : TowerOps {
    explicit PairingOps(Recorder& i) : TowerOps(i) {}

    // pairing_chip.rs:13-76
    G2Coeffs doubling_step(AssignedG2& pt) {
        AssignedFq2 x2 = fq2_square(pt.x);
        AssignedFq2 y2 = fq2_square(pt.y);
        AssignedFq2 _2y2 = fq2_double(y2);
        AssignedFq2 _4y2 = fq2_double(_2y2);
        AssignedFq2 _4y4 = fq2_square(_2y2);
        AssignedFq2 _8y4 = fq2_double(_4y4);
        AssignedFq2 z2 = fq2_square(pt.z);
        AssignedFq2 _4xy2;
        {
            AssignedFq2 t = fq2_mul(y2, pt.x);
            t = fq2_double(t);
            _4xy2 = fq2_double(t);
        }
        AssignedFq2 _3x2;
        {
            AssignedFq2 t = fq2_double(x2);
            _3x2 = fq2_add(t, x2);
        }
        AssignedFq2 _6x2 = fq2_double(_3x2);
        AssignedFq2 _9x4 = fq2_square(_3x2);
        AssignedFq2 _3x2_x = fq2_add(_3x2, pt.x);  // computed and discarded (quirk Q7)
        (void)_3x2_x;
        AssignedFq2 rx;
        {
            AssignedFq2 t = fq2_sub(_9x4, _4xy2);
            rx = fq2_sub(t, _4xy2);
        }
        AssignedFq2 ry;
        {
            AssignedFq2 t = fq2_sub(_4xy2, rx);
            t = fq2_mul(t, _3x2);
            ry = fq2_sub(t, _8y4);
        }
        AssignedFq2 rz;
        {
            AssignedFq2 yz = fq2_mul(pt.y, pt.z);
            rz = fq2_double(yz);
        }
        AssignedFq2 c0;
        {
            AssignedFq2 t = fq2_mul(z2, rz);
            c0 = fq2_double(t);
        }
        AssignedFq2 c1;
        {
            AssignedFq2 _6x2z2 = fq2_mul(z2, _6x2);
            c1 = fq2_neg(_6x2z2);
        }
        AssignedFq2 c2;
        {
            AssignedFq2 _6x3 = fq2_mul(_6x2, pt.x);
            c2 = fq2_sub(_6x3, _4y2);
        }
        pt = AssignedG2{rx, ry, rz};
        return G2Coeffs{c0, c1, c2};
    }

    // pairing_chip.rs:78-133
    G2Coeffs addition_step(AssignedG2& pt, const AssignedG2Affine& pq) {
        AssignedFq2 zt2 = fq2_square(pt.z);
        AssignedFq2 yqzt = fq2_mul(pq.y, pt.z);
        AssignedFq2 yqzt3 = fq2_mul(yqzt, zt2);
        AssignedFq2 yqzt3_yt = fq2_sub(yqzt3, pt.y);
        AssignedFq2 _2yqzt3_2yt = fq2_double(yqzt3_yt);

        AssignedFq2 xqzt2 = fq2_mul(pq.x, zt2);
        AssignedFq2 xqzt2_xt = fq2_sub(xqzt2, pt.x);
        AssignedFq2 _2_xqzt2_xt = fq2_double(xqzt2_xt);
        AssignedFq2 _4_xqzt2_xt_2 = fq2_square(_2_xqzt2_xt);

        AssignedFq2 rx;
        {
            AssignedFq2 t0 = fq2_mul(_4_xqzt2_xt_2, xqzt2_xt);
            AssignedFq2 t1 = fq2_double(_4_xqzt2_xt_2);
            AssignedFq2 t2 = fq2_mul(t1, pt.x);
            AssignedFq2 t = fq2_square(_2yqzt3_2yt);
            t = fq2_sub(t, t0);
            rx = fq2_sub(t, t2);
        }
        AssignedFq2 ry;
        {
            AssignedFq2 t0 = fq2_mul(_4_xqzt2_xt_2, pt.x);
            t0 = fq2_sub(t0, rx);
            t0 = fq2_mul(_2yqzt3_2yt, t0);
            AssignedFq2 t1 = fq2_mul(_2_xqzt2_xt, _4_xqzt2_xt_2);
            t1 = fq2_mul(t1, pt.y);
            ry = fq2_sub(t0, t1);
        }
        AssignedFq2 rz = fq2_mul(pt.z, _2_xqzt2_xt);
        AssignedFq2 c0 = fq2_double(rz);
        AssignedFq2 c1;
        {
            AssignedFq2 t = fq2_double(_2yqzt3_2yt);
            c1 = fq2_neg(t);
        }
        AssignedFq2 c2;
        {
            AssignedFq2 t0 = fq2_double(_2yqzt3_2yt);
            t0 = fq2_mul(t0, pq.x);
            AssignedFq2 t1 = fq2_mul(pq.y, rz);
            t1 = fq2_double(t1);
            c2 = fq2_sub(t0, t1);
        }
        pt = AssignedG2{rx, ry, rz};
        return G2Coeffs{c0, c1, c2};
    }

    // pairing_chip.rs:135-141
    AssignedG2 g2affine_to_g2(const AssignedG2Affine& g2) {
        ic.assert_false(g2.z);
        AssignedFq2 z = fq2_assign_one();
        return AssignedG2{g2.x, g2.y, z};
    }
    // pairing_chip.rs:143-146
    AssignedG2Affine g2_neg(const AssignedG2Affine& g2) {
        AssignedFq2 y = fq2_neg(g2.y);
        return AssignedG2Affine{g2.x, y, g2.z};
    }

    virtual AssignedG2Prepared prepare_g2(const AssignedG2Affine& g2) = 0;
    typedef std::pair<const AssignedPoint*, const AssignedG2Prepared*> PreparedTerm;
    virtual AssignedFq12 multi_miller_loop(const std::vector<PreparedTerm>& terms) = 0;
    virtual AssignedFq12 final_exponentiation(const AssignedFq12& f) = 0;

    typedef std::pair<const AssignedPoint*, const AssignedG2Affine*> Term;
    // pairing_chip.rs:157-171
    AssignedFq12 pairing(const std::vector<Term>& terms) {
        std::vector<AssignedG2Prepared> prepared;
        for (auto& t : terms) prepared.push_back(prepare_g2(*t.second));
        std::vector<PreparedTerm> pt;
        for (size_t i = 0; i < terms.size(); i++) pt.push_back(PreparedTerm(terms[i].first, &prepared[i]));
        AssignedFq12 res = multi_miller_loop(pt);
        stage_boundary(1);
        return final_exponentiation(res);
    }
    // Engine detail, no counterpart in the reference: the places where a pairing may be cut into launches of its own (a segment's
    // expansion then runs under the next segment's value chain).  Level 1: Miller loop | final exponentiation - one Fq12 value
    // crosses, and the final exponentiation cannot start before it anyway; level 2: also inside the final exponentiation, after
    // each exponentiation by x.  `stage_splits` = the highest level that is cut (0: one launch).
    int stage_splits = 0;
    void stage_boundary(int level) {
        if (level <= stage_splits) ic.split_segment();
    }
    // pairing_chip.rs:173-176
    void check_pairing(const std::vector<Term>& terms) {
        AssignedFq12 res = pairing(terms);
        fq12_assert_one(res);
    }
};

// ---------------------------------------------------------------- bn256
struct Bn256PairingOps : PairingOps {
    explicit Bn256PairingOps(Recorder& i) : PairingOps(i) {}
    static Fq2Const c2(const char* const v[2]) { return Fq2Const{HBig::from_hex(v[0]), HBig::from_hex(v[1])}; }

    // bn256_pairing_chip.rs:32-44 (xi = 9 + u via doublings)
    AssignedFq2 fq2_mul_by_nonresidue(const AssignedFq2& a) override {
        AssignedFq2 a2 = fq2_double(a);
        AssignedFq2 a4 = fq2_double(a2);
        AssignedFq2 a8 = fq2_double(a4);
        AssignedInteger t = ic.int_add(a8.c0, a.c0);
        AssignedInteger c0 = ic.int_sub(t, a.c1);
        t = ic.int_add(a8.c1, a.c0);
        AssignedInteger c1 = ic.int_add(t, a.c1);
        return AssignedFq2{c0, c1};
    }
    // bn256_pairing_chip.rs:46-53 (multiplies by the constant 1 for even powers, quirk Q7)
    AssignedFq2 fq2_frobenius_map(const AssignedFq2& x, size_t power) override {
        AssignedInteger v = ic.assign_int_constant(HBig::from_hex(h2e_const::BN_FROBENIUS_COEFF_FQ2_C1[power % 2]));
        AssignedInteger c1 = ic.int_mul(x.c1, v);
        return AssignedFq2{x.c0, c1};
    }
    // bn256_pairing_chip.rs:65-80
    AssignedFq6 fq6_frobenius_map(const AssignedFq6& x, size_t power) override {
        AssignedFq2 c0 = fq2_frobenius_map(x.c0, power);
        AssignedFq2 c1 = fq2_frobenius_map(x.c1, power);
        AssignedFq2 c2_ = fq2_frobenius_map(x.c2, power);
        AssignedFq2 coeff_c1 = fq2_assign_constant(c2(h2e_const::BN_FROBENIUS_COEFF_FQ6_C1[power % 6]));
        c1 = fq2_mul(c1, coeff_c1);
        AssignedFq2 coeff_c2 = fq2_assign_constant(c2(h2e_const::BN_FROBENIUS_COEFF_FQ6_C2[power % 6]));
        c2_ = fq2_mul(c2_, coeff_c2);
        return AssignedFq6{c0, c1, c2_};
    }
    // bn256_pairing_chip.rs:86-97
    AssignedFq12 fq12_frobenius_map(const AssignedFq12& x, size_t power) override {
        AssignedFq6 c0 = fq6_frobenius_map(x.c0, power);
        AssignedFq6 c1 = fq6_frobenius_map(x.c1, power);
        AssignedFq2 coeff = fq2_assign_constant(c2(h2e_const::BN_FROBENIUS_COEFF_FQ12_C1[power % 12]));
        AssignedFq2 c1c0 = fq2_mul(c1.c0, coeff);
        AssignedFq2 c1c1 = fq2_mul(c1.c1, coeff);
        AssignedFq2 c1c2 = fq2_mul(c1.c2, coeff);
        return AssignedFq12{c0, AssignedFq6{c1c0, c1c1, c1c2}};
    }
    // bn256_pairing_chip.rs:104-155
    AssignedG2Prepared prepare_g2(const AssignedG2Affine& g2) override {
        AssignedG2Affine neg_g2 = g2_neg(g2);
        AssignedG2Prepared out;
        AssignedG2 r = g2affine_to_g2(g2);
        const int NAF_LEN = 65;
        for (int i = NAF_LEN - 1; i >= 1; i--) {
            out.coeffs.push_back(doubling_step(r));
            int x = h2e_const::SIX_U_PLUS_2_NAF[i - 1];
            if (x == 1)
                out.coeffs.push_back(addition_step(r, g2));
            else if (x == -1)
                out.coeffs.push_back(addition_step(r, neg_g2));
        }
        AssignedG2Affine q1 = g2;
        AssignedFq2 c11 = fq2_assign_constant(c2(h2e_const::BN_FROBENIUS_COEFF_FQ6_C1[1]));
        AssignedFq2 c12 = fq2_assign_constant(c2(h2e_const::BN_FROBENIUS_COEFF_FQ6_C1[2]));
        AssignedFq2 xi = fq2_assign_constant(c2(h2e_const::BN_XI_TO_Q_MINUS_1_OVER_2));
        q1.x.c1 = ic.int_neg(q1.x.c1);
        q1.x = fq2_mul(q1.x, c11);
        q1.y.c1 = ic.int_neg(q1.y.c1);
        q1.y = fq2_mul(q1.y, xi);
        out.coeffs.push_back(addition_step(r, q1));
        AssignedG2Affine minusq2 = g2;
        minusq2.x = fq2_mul(minusq2.x, c12);
        out.coeffs.push_back(addition_step(r, minusq2));
        return out;
    }
    // bn256_pairing_chip.rs:157-174
    AssignedFq12 ell(const AssignedFq12& f, const G2Coeffs& coeffs, const AssignedPoint& p) {
        AssignedInteger c00 = ic.int_mul(coeffs[0].c0, p.y);
        AssignedInteger c01 = ic.int_mul(coeffs[0].c1, p.y);
        AssignedInteger c10 = ic.int_mul(coeffs[1].c0, p.x);
        AssignedInteger c11 = ic.int_mul(coeffs[1].c1, p.x);
        return fq12_mul_by_034(f, AssignedFq2{c00, c01}, AssignedFq2{c10, c11}, coeffs[2]);
    }
    // bn256_pairing_chip.rs:176-228
    AssignedFq12 multi_miller_loop(const std::vector<PreparedTerm>& terms) override {
        std::vector<size_t> it(terms.size(), 0);
        for (auto& t : terms) ic.assert_false(t.first->z);
        AssignedFq12 f = fq12_assign_one();
        const int NAF_LEN = 65;
        auto round = [&]() {
            for (size_t k = 0; k < terms.size(); k++) f = ell(f, terms[k].second->coeffs.at(it[k]++), *terms[k].first);
        };
        for (int i = NAF_LEN - 1; i >= 1; i--) {
            if (i != NAF_LEN - 1) f = fq12_square(f);
            round();
            int x = h2e_const::SIX_U_PLUS_2_NAF[i - 1];
            if (x == 1 || x == -1) round();
        }
        round();
        round();
        for (size_t k = 0; k < terms.size(); k++)
            if (it[k] != terms[k].second->coeffs.size()) throw std::runtime_error("miller loop: coeffs not exhausted");
        return f;
    }
    // bn256_pairing_chip.rs:230-240
    AssignedFq12 exp_by_x(const AssignedFq12& f) {
        uint64_t x = h2e_const::BN_X;
        AssignedFq12 res = fq12_assign_one();
        for (int i = 63; i >= 0; i--) {
            res = fq12_cyclotomic_square(res);
            if (((x >> i) & 1) == 1) res = fq12_mul(res, f);
        }
        return res;
    }
    // bn256_pairing_chip.rs:242-323
    AssignedFq12 final_exponentiation(const AssignedFq12& f) override {
        AssignedFq12 f1 = fq12_conjugate(f);
        AssignedFq12 f2 = fq12_unsafe_invert(f);
        AssignedFq12 r = fq12_mul(f1, f2);
        f2 = r;
        r = fq12_frobenius_map(r, 2);
        r = fq12_mul(r, f2);
        AssignedFq12 fp = fq12_frobenius_map(r, 1);
        AssignedFq12 fp2 = fq12_frobenius_map(r, 2);
        AssignedFq12 fp3 = fq12_frobenius_map(fp2, 1);
        AssignedFq12 fu = exp_by_x(r);
        stage_boundary(2);
        AssignedFq12 fu2 = exp_by_x(fu);
        stage_boundary(2);
        AssignedFq12 fu3 = exp_by_x(fu2);
        stage_boundary(3);
        AssignedFq12 y3 = fq12_frobenius_map(fu, 1);
        AssignedFq12 fu2p = fq12_frobenius_map(fu2, 1);
        AssignedFq12 fu3p = fq12_frobenius_map(fu3, 1);
        AssignedFq12 y2 = fq12_frobenius_map(fu2, 2);
        AssignedFq12 y0 = fq12_mul(fp, fp2);
        y0 = fq12_mul(y0, fp3);
        AssignedFq12 y1 = fq12_conjugate(r);
        AssignedFq12 y5 = fq12_conjugate(fu2);
        y3 = fq12_conjugate(y3);
        AssignedFq12 y4 = fq12_mul(fu, fu2p);
        y4 = fq12_conjugate(y4);
        AssignedFq12 y6 = fq12_mul(fu3, fu3p);
        y6 = fq12_conjugate(y6);
        y6 = fq12_cyclotomic_square(y6);
        y6 = fq12_mul(y6, y4);
        y6 = fq12_mul(y6, y5);
        AssignedFq12 t1 = fq12_mul(y3, y5);
        t1 = fq12_mul(t1, y6);
        y6 = fq12_mul(y6, y2);
        t1 = fq12_cyclotomic_square(t1);
        t1 = fq12_mul(t1, y6);
        t1 = fq12_cyclotomic_square(t1);
        AssignedFq12 t0 = fq12_mul(t1, y1);
        t1 = fq12_mul(t1, y0);
        t0 = fq12_cyclotomic_square(t0);
        t0 = fq12_mul(t0, t1);
        return t0;
    }
};

// ---------------------------------------------------------------- bls12_381
struct Bls12381PairingOps : PairingOps {
    explicit Bls12381PairingOps(Recorder& i) : PairingOps(i) {}
    static Fq2Const c2(const char* const v[2]) { return Fq2Const{HBig::from_hex(v[0]), HBig::from_hex(v[1])}; }

    // bls12_381_pairing_chip.rs:32-37 (xi = 1 + u)
    AssignedFq2 fq2_mul_by_nonresidue(const AssignedFq2& a) override {
        AssignedInteger c0 = ic.int_sub(a.c0, a.c1);
        AssignedInteger c1 = ic.int_add(a.c0, a.c1);
        return AssignedFq2{c0, c1};
    }
    // bls12_381_pairing_chip.rs:39-41
    AssignedFq2 fq2_frobenius_map(const AssignedFq2& x, size_t) override { return fq2_conjugate(x); }
    // bls12_381_pairing_chip.rs:51-82 (ignores `power`, quirk Q7)
    AssignedFq6 fq6_frobenius_map(const AssignedFq6& x, size_t power) override {
        AssignedFq2 c0 = fq2_frobenius_map(x.c0, power);
        AssignedFq2 c1 = fq2_frobenius_map(x.c1, power);
        AssignedFq2 c2_ = fq2_frobenius_map(x.c2, power);
        AssignedFq2 coeff_c1 = fq2_assign_constant(c2(h2e_const::BLS_FROBENIUS_COEFF_FQ6_C1));
        c1 = fq2_mul(c1, coeff_c1);
        AssignedFq2 coeff_c2 = fq2_assign_constant(c2(h2e_const::BLS_FROBENIUS_COEFF_FQ6_C2));
        c2_ = fq2_mul(c2_, coeff_c2);
        return AssignedFq6{c0, c1, c2_};
    }
    // bls12_381_pairing_chip.rs:88-115
    AssignedFq12 fq12_frobenius_map(const AssignedFq12& x, size_t power) override {
        AssignedFq6 c0 = fq6_frobenius_map(x.c0, power);
        AssignedFq6 c1 = fq6_frobenius_map(x.c1, power);
        AssignedFq2 coeff = fq2_assign_constant(c2(h2e_const::BLS_FROBENIUS_COEFF_FQ12_C1));
        AssignedFq2 c1c0 = fq2_mul(c1.c0, coeff);
        AssignedFq2 c1c1 = fq2_mul(c1.c1, coeff);
        AssignedFq2 c1c2 = fq2_mul(c1.c2, coeff);
        return AssignedFq12{c0, AssignedFq6{c1c0, c1c1, c1c2}};
    }
    // bls12_381_pairing_chip.rs:123-140
    AssignedFq12 ell(const AssignedFq12& f, const G2Coeffs& coeffs, const AssignedPoint& p) {
        AssignedInteger c00 = ic.int_mul(coeffs[0].c0, p.y);
        AssignedInteger c01 = ic.int_mul(coeffs[0].c1, p.y);
        AssignedInteger c10 = ic.int_mul(coeffs[1].c0, p.x);
        AssignedInteger c11 = ic.int_mul(coeffs[1].c1, p.x);
        return fq12_mul_by_014(f, coeffs[2], AssignedFq2{c10, c11}, AssignedFq2{c00, c01});
    }
    // bls12_381_pairing_chip.rs:142-159
    AssignedFq12 cycolotomic_exp(const AssignedFq12& f) {
        uint64_t x = h2e_const::BLS_X;
        AssignedFq12 tmp = fq12_assign_one();
        bool found_one = false;
        for (int b = 63; b >= 0; b--) {
            bool i = ((x >> b) & 1) == 1;
            if (found_one)
                tmp = fq12_cyclotomic_square(tmp);
            else
                found_one = i;
            if (i) tmp = fq12_mul(tmp, f);
        }
        return fq12_conjugate(tmp);
    }
    // bls12_381_pairing_chip.rs:165-189
    AssignedG2Prepared prepare_g2(const AssignedG2Affine& g2) override {
        AssignedG2 f = g2affine_to_g2(g2);
        AssignedG2Prepared out;
        bool found_one = false;
        for (int b = 63; b >= 0; b--) {
            bool i = (((h2e_const::BLS_X >> 1) >> b) & 1) == 1;
            if (!found_one) {
                found_one = i;
                continue;
            }
            out.coeffs.push_back(doubling_step(f));
            if (i) out.coeffs.push_back(addition_step(f, g2));
        }
        out.coeffs.push_back(doubling_step(f));
        return out;
    }
    // bls12_381_pairing_chip.rs:191-234
    AssignedFq12 multi_miller_loop(const std::vector<PreparedTerm>& terms) override {
        std::vector<size_t> it(terms.size(), 0);
        for (auto& t : terms) ic.assert_false(t.first->z);
        AssignedFq12 f = fq12_assign_one();
        auto round = [&]() {
            for (size_t k = 0; k < terms.size(); k++) f = ell(f, terms[k].second->coeffs.at(it[k]++), *terms[k].first);
        };
        bool found_one = false;
        for (int b = 63; b >= 0; b--) {
            bool i = (((h2e_const::BLS_X >> 1) >> b) & 1) == 1;
            if (!found_one) {
                found_one = i;
                continue;
            }
            round();
            if (i) round();
            f = fq12_square(f);
        }
        round();
        f = fq12_conjugate(f);
        return f;
    }
    // bls12_381_pairing_chip.rs:236-286
    AssignedFq12 final_exponentiation(const AssignedFq12& f) override {
        const size_t PH = 1;
        AssignedFq12 t0 = fq12_frobenius_map(f, PH);
        for (int k = 0; k < 5; k++) t0 = fq12_frobenius_map(t0, PH);
        AssignedFq12 t1 = fq12_unsafe_invert(f);
        AssignedFq12 t2 = fq12_mul(t0, t1);
        t1 = t2;
        t2 = fq12_frobenius_map(t2, PH);
        t2 = fq12_frobenius_map(t2, PH);
        t2 = fq12_mul(t2, t1);
        t1 = fq12_cyclotomic_square(t2);
        t1 = fq12_conjugate(t1);
        AssignedFq12 t3 = cycolotomic_exp(t2);
        AssignedFq12 t4 = fq12_cyclotomic_square(t3);
        AssignedFq12 t5 = fq12_mul(t1, t3);
        stage_boundary(2);
        t1 = cycolotomic_exp(t5);
        stage_boundary(3);
        t0 = cycolotomic_exp(t1);
        stage_boundary(2);
        AssignedFq12 t6 = cycolotomic_exp(t0);
        t6 = fq12_mul(t6, t4);
        t4 = cycolotomic_exp(t6);
        t5 = fq12_conjugate(t5);
        AssignedFq12 t = fq12_mul(t5, t2);
        t4 = fq12_mul(t4, t);
        t5 = fq12_conjugate(t2);
        t1 = fq12_mul(t1, t2);
        for (int k = 0; k < 3; k++) t1 = fq12_frobenius_map(t1, PH);
        t6 = fq12_mul(t6, t5);
        t6 = fq12_frobenius_map(t6, PH);
        t3 = fq12_mul(t3, t0);
        for (int k = 0; k < 2; k++) t3 = fq12_frobenius_map(t3, PH);
        t3 = fq12_mul(t3, t1);
        t3 = fq12_mul(t3, t6);
        return fq12_mul(t3, t4);
    }
};

}  // namespace h2e
