// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Not part of the product path.
//
// L3 ECC / MSM, restating:
//   src/assign.rs:39-65                             (point handle types)
//   src/circuit/ecc_chip.rs:20-77                   (MSM_PREFIX_OFFSET, UnsafeError, Offset, ParallelClone)
//   src/circuit/ecc_chip.rs:79-430                  (EccChipScalarOps: both MSM variants, msm_unsafe, msm, ecc_mul)
//   src/circuit/ecc_chip.rs:438-1009                (EccChipBaseOps)
//   src/circuit/native_scalar_ecc_chip.rs:27-194    (NativeScalarEccContext glue, decompose_scalar)
//   src/circuit/general_scalar_ecc_chip.rs:26-169   (GeneralScalarEccContext glue, decompose_scalar)
//   src/context.rs:190-239                          (context wrappers)
// Points enter as affine canonical coordinates (what `to_affine().coordinates()` + field_to_bn give).
#pragma once
#include "integer_chip.hpp"

namespace h2o {

static const size_t MSM_PREFIX_OFFSET = 1u << 20;              // ecc_chip.rs:20
static const size_t MSM_LIMIT = (1u << 8) * MSM_PREFIX_OFFSET; // ecc_chip.rs:21

struct UnsafeError : std::runtime_error {  // ecc_chip.rs:23-34
    enum Kind { AddSameOrNegPoint, AddIdentity, AssignIdentity } kind;
    UnsafeError(Kind k) : std::runtime_error("UnsafeError"), kind(k) {}
    bool can_retry() const { return true; }
};

struct Offset {  // ecc_chip.rs:36-62
    size_t range_offset_diff = 0, base_offset_diff = 0, select_offset_diff = 0;
    Offset operator-(const Offset& r) const {
        Offset o;
        o.base_offset_diff = base_offset_diff - r.base_offset_diff;
        o.range_offset_diff = range_offset_diff - r.range_offset_diff;
        o.select_offset_diff = select_offset_diff - r.select_offset_diff;
        return o;
    }
    Offset scale(size_t n) const {
        Offset o;
        o.range_offset_diff = range_offset_diff * n;
        o.base_offset_diff = base_offset_diff * n;
        o.select_offset_diff = select_offset_diff * n;
        return o;
    }
    bool operator==(const Offset& r) const {
        return range_offset_diff == r.range_offset_diff && base_offset_diff == r.base_offset_diff &&
               select_offset_diff == r.select_offset_diff;
    }
};

struct NativePoint {  // affine canonical coordinates of a C::CurveExt value
    BigUint x, y;
    bool is_identity = false;
};

struct AssignedPoint {  // assign.rs:46-51
    AssignedInteger x, y;
    AssignedCondition z;
};
struct AssignedNonZeroPoint {  // assign.rs:53-57
    AssignedInteger x, y;
};
struct AssignedCurvature {  // assign.rs:39-43
    AssignedInteger v;
    AssignedCondition z;
};
struct AssignedPointWithCurvature {  // assign.rs:59-65
    AssignedInteger x, y;
    AssignedCondition z;
    AssignedCurvature curvature;
    AssignedPoint to_point() const { return AssignedPoint{x, y, z}; }
};

struct CurveParams {
    BigUint base_modulus;    // C::Base
    BigUint scalar_modulus;  // C::ScalarExt
    BigUint b;               // C::b()
    NativePoint generator;   // C::generator()
    uint32_t scalar_num_bits;  // PrimeField::NUM_BITS of the scalar field
};

// EccBaseIntegerChipWrapper + EccChipBaseOps (ecc_chip.rs:432-1009)
struct EccContext {
    IntegerContext base;  // base_integer_chip(); also the select chip when available
    CurveParams curve;
    size_t msm_prefix;    // usize::MAX => no select chip (NativeScalarEccContext.1)
    std::shared_ptr<Context> native_ctx() { return base.ctx; }

    EccContext(const IntegerContext& ic, const CurveParams& cp, size_t prefix) : base(ic), curve(cp), msm_prefix(prefix) {}
    virtual ~EccContext() {}

    bool has_select_chip() const { return msm_prefix != (size_t)-1; }
    IntegerContext& select_chip() {
        if (!has_select_chip()) throw PanicError("ERROR: select chip is not available");
        return base;
    }

    // ecc_chip.rs:441-456
    AssignedPoint assign_constant_point(const NativePoint& c) {
        BigUint x = c.is_identity ? BigUint(0) : c.x, y = c.is_identity ? BigUint(0) : c.y;
        Fr z = c.is_identity ? Fr::one() : Fr::zero();
        AssignedInteger ax = base.assign_int_constant(x);
        AssignedInteger ay = base.assign_int_constant(y);
        AssignedValue az = base.ctx->assign_constant(z);
        return AssignedPoint{ax, ay, AssignedCondition(az)};
    }
    // ecc_chip.rs:458-487
    AssignedPoint assign_point(const NativePoint& c) {
        BigUint xv = c.is_identity ? BigUint(0) : c.x, yv = c.is_identity ? BigUint(0) : c.y;
        Fr zv = c.is_identity ? Fr::one() : Fr::zero();
        AssignedInteger x = base.assign_w(xv);
        AssignedInteger y = base.assign_w(yv);
        AssignedCondition z = base.ctx->assign_bit(zv);
        AssignedInteger b = base.assign_int_constant(curve.b);
        AssignedInteger y2 = base.int_square(y);
        AssignedInteger x2 = base.int_square(x);
        AssignedInteger x3 = base.int_mul(x2, x);
        AssignedInteger right = base.int_add(x3, b);
        AssignedCondition eq = base.is_int_equal(y2, right);
        AssignedCondition eq_or_identity = base.ctx->or_(eq, z);
        base.ctx->assert_true(eq_or_identity);
        return AssignedPoint{x, y, z};
    }
    // ecc_chip.rs:489-512
    AssignedNonZeroPoint assign_non_zero_point(const NativePoint& c) {
        if (c.is_identity) throw PanicError("assign_non_zero_point: identity");
        AssignedInteger x = base.assign_w(c.x);
        AssignedInteger y = base.assign_w(c.y);
        AssignedInteger b = base.assign_int_constant(curve.b);
        AssignedInteger y2 = base.int_square(y);
        AssignedInteger x2 = base.int_square(x);
        AssignedInteger x3 = base.int_mul(x2, x);
        AssignedInteger right = base.int_add(x3, b);
        base.assert_int_equal(y2, right);
        return AssignedNonZeroPoint{x, y};
    }
    // ecc_chip.rs:514-529
    AssignedPointWithCurvature assign_identity() {
        AssignedInteger zero = base.assign_int_constant(BigUint(0));
        AssignedValue one = base.ctx->assign_constant(Fr::one());
        return AssignedPointWithCurvature{zero, zero, AssignedCondition(one),
                                          AssignedCurvature{zero, AssignedCondition(one)}};
    }
    // ecc_chip.rs:531-545
    AssignedPoint bisec_point(const AssignedCondition& cond, const AssignedPoint& a, const AssignedPoint& b) {
        AssignedInteger x = base.bisec_int(cond, a.x, b.x);
        AssignedInteger y = base.bisec_int(cond, a.y, b.y);
        AssignedCondition z = base.ctx->bisec_cond(cond, a.z, b.z);
        return AssignedPoint{x, y, z};
    }
    // ecc_chip.rs:547-560
    AssignedCurvature bisec_curvature(const AssignedCondition& cond, const AssignedCurvature& a,
                                      const AssignedCurvature& b) {
        AssignedInteger v = base.bisec_int(cond, a.v, b.v);
        AssignedCondition z = base.ctx->bisec_cond(cond, a.z, b.z);
        return AssignedCurvature{v, z};
    }
    // ecc_chip.rs:562-578
    AssignedPointWithCurvature bisec_point_with_curvature(const AssignedCondition& cond,
                                                          const AssignedPointWithCurvature& a,
                                                          const AssignedPointWithCurvature& b) {
        AssignedInteger x = base.bisec_int(cond, a.x, b.x);
        AssignedInteger y = base.bisec_int(cond, a.y, b.y);
        AssignedCondition z = base.ctx->bisec_cond(cond, a.z, b.z);
        AssignedCurvature c = bisec_curvature(cond, a.curvature, b.curvature);
        return AssignedPointWithCurvature{x, y, z, c};
    }
    // ecc_chip.rs:580-604
    AssignedPoint lambda_to_point(const AssignedCurvature& lambda, const AssignedPoint& a, const AssignedPoint& b) {
        const AssignedInteger& l = lambda.v;
        AssignedInteger l_square = base.int_square(l);
        AssignedInteger t = base.int_sub(l_square, a.x);
        AssignedInteger cx = base.int_sub(t, b.x);
        AssignedInteger t2 = base.int_sub(a.x, cx);
        t2 = base.int_mul(t2, l);
        AssignedInteger cy = base.int_sub(t2, a.y);
        return AssignedPoint{cx, cy, lambda.z};
    }
    // ecc_chip.rs:606-628
    AssignedPoint ecc_add(const AssignedPointWithCurvature& a, const AssignedPoint& b) {
        AssignedInteger diff_x = base.int_sub(a.x, b.x);
        AssignedInteger diff_y = base.int_sub(a.y, b.y);
        auto dv = base.int_div(diff_y, diff_x);
        AssignedCondition x_eq = dv.first;
        AssignedCondition y_eq = base.is_int_zero(diff_y);
        AssignedCondition eq = base.ctx->and_(x_eq, y_eq);
        AssignedCurvature tangent{dv.second, x_eq};
        AssignedCurvature lambda = bisec_curvature(eq, a.curvature, tangent);
        AssignedPoint a_p = a.to_point();
        AssignedPoint p = lambda_to_point(lambda, a_p, b);
        p = bisec_point(a.z, b, p);
        p = bisec_point(b.z, a_p, p);
        return p;
    }
    // ecc_chip.rs:630-642
    AssignedPoint ecc_double(const AssignedPointWithCurvature& a) {
        if (curve.scalar_modulus.bit(0) == false) throw PanicError("ecc_double: even scalar order");
        AssignedPoint a_p = a.to_point();
        AssignedPoint p = lambda_to_point(a.curvature, a_p, a_p);
        p.z = base.ctx->bisec_cond(a.z, a.z, p.z);
        return p;
    }
    // ecc_chip.rs:644-658
    void ecc_assert_equal(const AssignedPoint& a, const AssignedPoint& b) {
        AssignedCondition eq_x = base.is_int_equal(a.x, b.x);
        AssignedCondition eq_y = base.is_int_equal(a.y, b.y);
        AssignedCondition eq_z = base.ctx->xnor(a.z, b.z);
        AssignedCondition eq_xy = base.ctx->and_(eq_x, eq_y);
        AssignedCondition eq_xyz = base.ctx->and_(eq_xy, eq_z);
        AssignedCondition is_both_identity = base.ctx->and_(a.z, b.z);
        AssignedCondition eq = base.ctx->or_(eq_xyz, is_both_identity);
        base.ctx->assert_true(eq);
    }
    // ecc_chip.rs:660-666
    AssignedPoint ecc_neg(const AssignedPoint& a) {
        AssignedInteger y = base.int_neg(a.y);
        return AssignedPoint{a.x, y, a.z};
    }
    // ecc_chip.rs:668-675
    AssignedPoint ecc_reduce(const AssignedPoint& a) {
        AssignedInteger x = base.reduce(a.x);
        AssignedInteger y = base.reduce(a.y);
        AssignedCondition z = a.z;
        AssignedPointWithCurvature identity = assign_identity();
        return bisec_point(z, identity.to_point(), AssignedPoint{x, y, z});
    }
    // ecc_chip.rs:677-693
    AssignedPointWithCurvature ecc_reduce_with_curvature(const AssignedPoint& a_in) {
        AssignedPoint a = ecc_reduce(a_in);
        AssignedInteger x_square = base.int_square(a.x);
        AssignedInteger numerator = base.int_mul_small_constant(x_square, 3);
        AssignedInteger denominator = base.int_mul_small_constant(a.y, 2);
        auto zv = base.int_div(numerator, denominator);
        AssignedInteger v = base.reduce(zv.second);
        return AssignedPointWithCurvature{a.x, a.y, a.z, AssignedCurvature{v, zv.first}};
    }
    // ecc_chip.rs:695-708
    AssignedPointWithCurvature to_point_with_curvature(const AssignedPoint& a) {
        AssignedInteger x_square = base.int_square(a.x);
        AssignedInteger numerator = base.int_mul_small_constant(x_square, 3);
        AssignedInteger denominator = base.int_mul_small_constant(a.y, 2);
        auto zv = base.int_div(numerator, denominator);
        return AssignedPointWithCurvature{a.x, a.y, a.z, AssignedCurvature{zv.second, zv.first}};
    }
    // ecc_chip.rs:710-732
    std::vector<AssignedValue> ecc_encode(const AssignedPoint& p_in) {
        AssignedPoint p = ecc_reduce(p_in);
        Fr shift = Fr::from_bn(BigUint(1) << base.info->limb_bits);
        typedef Context::Elem Elem;
        AssignedValue s0 = base.ctx->sum_with_constant({Elem(&p.x.limbs_le[0], Fr::one()), Elem(&p.x.limbs_le[1], shift)}, nullptr);
        AssignedValue s1 = base.ctx->sum_with_constant({Elem(&p.x.limbs_le[2], Fr::one()), Elem(&p.y.limbs_le[0], shift)}, nullptr);
        AssignedValue s2 = base.ctx->sum_with_constant({Elem(&p.y.limbs_le[1], Fr::one()), Elem(&p.y.limbs_le[2], shift)}, nullptr);
        return {s0, s1, s2};
    }
    // ecc_chip.rs:734-751
    void assign_cache_integer(const AssignedInteger& p, size_t sc, size_t g, size_t& offset) {
        if (p.times != 1) throw PanicError("assign_cache_integer: times != 1");
        for (size_t j = 0; j < (size_t)base.info->limbs; j++) {
            select_chip().assign_cache_value(p.limbs_le[j], offset, g, sc);
            offset += 1;
        }
        select_chip().assign_cache_value(p.native, offset, g, sc);
        offset += 1;
    }
    // ecc_chip.rs:753-777
    AssignedInteger assign_selected_integer(const AssignedInteger& p, const AssignedValue& sc, size_t g, size_t& offset) {
        std::vector<AssignedValue> limbs_le;
        for (size_t j = 0; j < (size_t)base.info->limbs; j++) {
            limbs_le.push_back(select_chip().assign_selected_value(p.limbs_le[j], offset, g, sc));
            offset += 1;
        }
        AssignedValue native = select_chip().assign_selected_value(p.native, offset, g, sc);
        offset += 1;
        return AssignedInteger(limbs_le, native, 1);
    }
    // ecc_chip.rs:779-788
    void assign_cache_point(const AssignedPointWithCurvature& p, size_t g, size_t sc) {
        size_t i = 0;
        assign_cache_integer(p.x, sc, g, i);
        assign_cache_integer(p.y, sc, g, i);
        select_chip().assign_cache_value(p.z.v, i, g, sc);
        i += 1;
        assign_cache_integer(p.curvature.v, sc, g, i);
        select_chip().assign_cache_value(p.curvature.z.v, i, g, sc);
    }
    // ecc_chip.rs:790-812
    AssignedPointWithCurvature assign_selected_point(const AssignedPointWithCurvature& p, const AssignedValue& sc, size_t g) {
        size_t i = 0;
        AssignedInteger x = assign_selected_integer(p.x, sc, g, i);
        AssignedInteger y = assign_selected_integer(p.y, sc, g, i);
        AssignedValue z = select_chip().assign_selected_value(p.z.v, i, g, sc);
        i += 1;
        AssignedInteger c_v = assign_selected_integer(p.curvature.v, sc, g, i);
        AssignedValue c_z = select_chip().assign_selected_value(p.curvature.z.v, i, g, sc);
        return AssignedPointWithCurvature{x, y, AssignedCondition(z), AssignedCurvature{c_v, AssignedCondition(c_z)}};
    }
    // ecc_chip.rs:814-838
    AssignedNonZeroPoint lambda_to_point_non_zero(const AssignedInteger& l, const AssignedNonZeroPoint& a,
                                                  const AssignedNonZeroPoint& b) {
        AssignedInteger l_square = base.int_square(l);
        AssignedInteger t = base.int_sub(l_square, a.x);
        AssignedInteger cx = base.int_sub(t, b.x);
        AssignedInteger t2 = base.int_sub(a.x, cx);
        t2 = base.int_mul(t2, l);
        AssignedInteger cy = base.int_sub(t2, a.y);
        return AssignedNonZeroPoint{cx, cy};
    }
    // ecc_chip.rs:840-858
    AssignedNonZeroPoint ecc_add_unsafe(const AssignedNonZeroPoint& a, const AssignedNonZeroPoint& b) {
        AssignedInteger diff_x = base.int_sub(a.x, b.x);
        AssignedInteger diff_y = base.int_sub(a.y, b.y);
        auto dv = base.int_div(diff_y, diff_x);
        bool succeed = base.ctx->try_assert_false(dv.first);
        AssignedNonZeroPoint res = lambda_to_point_non_zero(dv.second, a, b);
        if (!succeed) throw UnsafeError(UnsafeError::AddSameOrNegPoint);
        return res;
    }
    // ecc_chip.rs:860-882
    AssignedNonZeroPoint ecc_double_unsafe(const AssignedNonZeroPoint& a) {
        AssignedInteger x_square = base.int_square(a.x);
        AssignedInteger numerator = base.int_mul_small_constant(x_square, 3);
        AssignedInteger denominator = base.int_mul_small_constant(a.y, 2);
        auto zv = base.int_div(numerator, denominator);
        bool succeed = base.ctx->try_assert_false(zv.first);
        AssignedNonZeroPoint res = lambda_to_point_non_zero(zv.second, a, a);
        if (!succeed) throw UnsafeError(UnsafeError::AddIdentity);
        return res;
    }
    // ecc_chip.rs:884-889
    AssignedNonZeroPoint ecc_neg_non_zero(const AssignedNonZeroPoint& a) {
        AssignedInteger y = base.int_neg(a.y);
        return AssignedNonZeroPoint{a.x, y};
    }
    // ecc_chip.rs:891-899
    AssignedNonZeroPoint ecc_reduce_non_zero(const AssignedNonZeroPoint& a) {
        AssignedInteger x = base.reduce(a.x);
        AssignedInteger y = base.reduce(a.y);
        return AssignedNonZeroPoint{x, y};
    }
    // ecc_chip.rs:901-911
    AssignedNonZeroPoint ecc_bisec_non_zero_point(const AssignedCondition& cond, const AssignedNonZeroPoint& a,
                                                  const AssignedNonZeroPoint& b) {
        AssignedInteger x = base.bisec_int(cond, a.x, b.x);
        AssignedInteger y = base.bisec_int(cond, a.y, b.y);
        return AssignedNonZeroPoint{x, y};
    }
    // ecc_chip.rs:913-933
    AssignedNonZeroPoint bisec_candidate_non_zero(const std::vector<AssignedNonZeroPoint>& candidates,
                                                  const std::vector<AssignedCondition>& group_bits) {
        std::vector<AssignedNonZeroPoint> curr = candidates;
        for (auto& bit : group_bits) {
            std::vector<AssignedNonZeroPoint> next;
            for (size_t k = 0; k < curr.size(); k += 2) {
                if (k + 1 >= curr.size()) throw PanicError("bisec_candidate: odd chunk");  // it[1] out of bounds
                next.push_back(ecc_bisec_non_zero_point(bit, curr[k + 1], curr[k]));
            }
            curr = next;
        }
        if (curr.size() != 1) throw PanicError("bisec_candidate: size != 1");
        return curr[0];
    }
    // ecc_chip.rs:935-953
    std::pair<AssignedValue, AssignedNonZeroPoint> pick_candidate_non_zero(
        const std::vector<AssignedNonZeroPoint>& candidates, const std::vector<AssignedCondition>& group_bits) {
        std::vector<Context::Elem> index_vec;
        for (size_t i = 0; i < group_bits.size(); i++)
            index_vec.push_back(Context::Elem(&group_bits[i].v, Fr::from_u64(1ull << i)));
        AssignedValue index = base.ctx->sum_with_constant(index_vec, nullptr);
        uint64_t canon[4];
        index.val.to_canonical(canon);
        size_t index_i = (size_t)(canon[0] & 0xff);
        return std::make_pair(index, candidates.at(index_i));
    }
    // ecc_chip.rs:955-967
    AssignedNonZeroPoint assign_selected_point_non_zero(const AssignedNonZeroPoint& p, const AssignedValue& sc, size_t g) {
        size_t i = 0;
        AssignedInteger x = assign_selected_integer(p.x, sc, g, i);
        AssignedInteger y = assign_selected_integer(p.y, sc, g, i);
        return AssignedNonZeroPoint{x, y};
    }
    // ecc_chip.rs:969-973
    void assign_cache_point_non_zero(const AssignedNonZeroPoint& p, size_t g, size_t sc) {
        size_t i = 0;
        assign_cache_integer(p.x, sc, g, i);
        assign_cache_integer(p.y, sc, g, i);
    }
    // ecc_chip.rs:975-982
    void ecc_assert_equal_non_zero(const AssignedNonZeroPoint& a, const AssignedNonZeroPoint& b) {
        base.assert_int_equal(a.x, b.x);
        base.assert_int_equal(a.y, b.y);
    }
    // ecc_chip.rs:984-997
    AssignedPoint ecc_non_zero_point_downgrade(const AssignedNonZeroPoint& a) {
        AssignedValue zero = base.ctx->assign_constant(Fr::zero());
        return AssignedPoint{a.x, a.y, AssignedCondition(zero)};
    }
    // ecc_chip.rs:999-1008
    AssignedNonZeroPoint ecc_bisec_to_non_zero_point(const AssignedPoint& a, const AssignedNonZeroPoint& b) {
        AssignedInteger x = base.bisec_int(a.z, b.x, a.x);
        AssignedInteger y = base.bisec_int(a.z, b.y, a.y);
        return AssignedNonZeroPoint{x, y};
    }
};

// EccChipScalarOps with the scalar type as a template parameter (ecc_chip.rs:79-430)
template <class Derived, class AssignedScalar>
struct EccScalarOps : EccContext {
    using EccContext::EccContext;
    Derived& self() { return *static_cast<Derived*>(this); }

    int n_threads = 1;  // window-parallel region (rayon par_iter_mut, ecc_chip.rs:317-343)

    // ecc_chip.rs:223-371 (and :91-221 when select==false)
    AssignedPoint msm_batch_on_group_non_zero(bool with_select, const std::vector<AssignedNonZeroPoint>& points_in,
                                              const std::vector<AssignedScalar>& scalars, const NativePoint& r1,
                                              const NativePoint& r2) {
        if (with_select && !(points_in.size() <= MSM_PREFIX_OFFSET)) throw PanicError("msm: too many points");
        std::vector<AssignedNonZeroPoint> points;
        for (auto& p : points_in) points.push_back(ecc_reduce_non_zero(p));

        AssignedNonZeroPoint rand_acc_point = assign_non_zero_point(r1);
        AssignedNonZeroPoint rand_line_point = assign_non_zero_point(r2);
        AssignedNonZeroPoint rand_acc_point_neg = ecc_neg_non_zero(rand_acc_point);
        rand_acc_point_neg = ecc_reduce_non_zero(rand_acc_point_neg);
        AssignedNonZeroPoint rand_line_point_neg = ecc_neg_non_zero(rand_line_point);
        rand_line_point_neg = ecc_reduce_non_zero(rand_line_point_neg);

        size_t best_group_size = with_select ? 5 : 2;
        size_t n_group = (points.size() + best_group_size - 1) / best_group_size;
        size_t group_size = (points.size() + n_group - 1) / n_group;

        std::vector<std::vector<AssignedNonZeroPoint>> candidates;
        size_t group_prefix = with_select ? self().get_and_increase_msm_prefix() : 0;
        size_t n_chunks = (points.size() + group_size - 1) / group_size;
        for (size_t group_index = 0; group_index < n_chunks; group_index++) {
            size_t lo = group_index * group_size, hi = std::min(points.size(), lo + group_size);
            const AssignedNonZeroPoint& init = (group_index % 2 == 0) ? rand_line_point : rand_line_point_neg;
            candidates.push_back({init});
            if (with_select) assign_cache_point_non_zero(init, group_prefix + group_index, 0);
            std::vector<AssignedNonZeroPoint>& cl = candidates.back();
            for (uint32_t i = 1; i < (1u << (hi - lo)); i++) {
                uint32_t pos = __builtin_ctz(i);  // i.reverse_bits().leading_zeros()
                uint32_t other = i - (1u << pos);
                AssignedNonZeroPoint p = ecc_add_unsafe(cl[other], points[lo + pos]);
                p = ecc_reduce_non_zero(p);
                if (with_select) assign_cache_point_non_zero(p, group_prefix + group_index, i);
                cl.push_back(p);
            }
        }

        std::vector<std::vector<AssignedCondition>> bits;  // WINDOW_SIZE = 1
        for (auto& s : scalars) bits.push_back(self().decompose_scalar(s));
        size_t n_groups = (bits.size() + group_size - 1) / group_size;
        size_t windows = bits[0].size();

        auto window_body = [&](Derived& op, size_t wi) -> AssignedNonZeroPoint {
            AssignedNonZeroPoint acc = rand_acc_point_neg;
            for (size_t group_index = 0; group_index < n_groups; group_index++) {
                size_t lo = group_index * group_size, hi = std::min(bits.size(), lo + group_size);
                std::vector<AssignedCondition> group_bits;
                for (size_t j = lo; j < hi; j++) group_bits.push_back(bits[j][wi]);
                AssignedNonZeroPoint ci;
                if (with_select) {
                    auto pick = op.pick_candidate_non_zero(candidates[group_index], group_bits);
                    ci = op.assign_selected_point_non_zero(pick.second, pick.first, group_index + group_prefix);
                } else {
                    ci = op.bisec_candidate_non_zero(candidates[group_index], group_bits);
                }
                acc = op.ecc_add_unsafe(ci, acc);
            }
            return acc;
        };

        // predict_ops: window 0 on a clone to learn the per-window Offset (ecc_chip.rs:289-309)
        Derived predict_ops = self().clone_without_offset();
        Offset offset_before = predict_ops.offset();
        std::vector<AssignedNonZeroPoint> line_acc_arr;
        line_acc_arr.push_back(window_body(predict_ops, 0));
        Offset offset_after = predict_ops.offset();
        Offset offset_diff = offset_after - offset_before;
        self().merge(predict_ops);

        // Parallel setup on window (ecc_chip.rs:312-343)
        std::vector<Derived> cloned_ops;
        for (size_t i = 1; i < windows; i++) cloned_ops.push_back(self().clone_with_offset(offset_diff.scale(i)));
        {
            // every clone writes a disjoint row range of the shared arrays; pre-size them
            Offset end = offset_before;
            base.ctx->records.inner->reserve_rows(end.base_offset_diff + offset_diff.base_offset_diff * windows + 2,
                                                  end.range_offset_diff + offset_diff.range_offset_diff * windows + 4,
                                                  end.select_offset_diff + offset_diff.select_offset_diff * windows + 2);
        }
        line_acc_arr.resize(windows);
        self().run_windows(cloned_ops, [&](size_t k) {
            Derived& op = cloned_ops[k];
            size_t wi = k + 1;
            Offset ob = op.offset();
            line_acc_arr[wi] = window_body(op, wi);
            Offset oa = op.offset();
            if (!((oa - ob) == offset_diff)) throw PanicError("msm: per-window offset diff mismatch");
        });
        for (auto& op : cloned_ops) self().merge(op);

        self().apply_offset_diff(offset_diff.scale(windows));

        // Accumulate points of all windows (ecc_chip.rs:354-362)
        AssignedNonZeroPoint acc = rand_acc_point;
        for (size_t wi = 0; wi < windows; wi++) {
            acc = ecc_double_unsafe(acc);
            acc = ecc_add_unsafe(line_acc_arr[wi], acc);
            if (n_groups % 2 == 1) acc = ecc_add_unsafe(acc, rand_line_point_neg);
        }
        AssignedPoint accp = ecc_non_zero_point_downgrade(acc);
        AssignedPointWithCurvature accc = to_point_with_curvature(accp);
        AssignedPoint carry = ecc_non_zero_point_downgrade(rand_acc_point_neg);
        return ecc_add(accc, carry);
    }

    // ecc_chip.rs:373-408; r1, r2 are the blinding points `generator * Scalar::rand()` made explicit (quirk Q1)
    AssignedPoint msm_unsafe(const std::vector<AssignedPoint>& points, const std::vector<AssignedScalar>& scalars,
                             const NativePoint& r1, const NativePoint& r2) {
        std::vector<AssignedNonZeroPoint> non_zero_points;
        std::vector<AssignedScalar> normalized_scalars;
        AssignedNonZeroPoint non_zero_p = assign_non_zero_point(curve.generator);
        AssignedScalar s_zero = self().ecc_assign_constant_zero_scalar();
        for (size_t i = 0; i < points.size(); i++) {
            AssignedScalar s = self().ecc_bisec_scalar(points[i].z, s_zero, scalars[i]);
            AssignedNonZeroPoint p = ecc_bisec_to_non_zero_point(points[i], non_zero_p);
            non_zero_points.push_back(p);
            normalized_scalars.push_back(s);
        }
        return msm_batch_on_group_non_zero(has_select_chip(), non_zero_points, normalized_scalars, r1, r2);
    }
};

// NativeScalarEccContext (context.rs:190-207, native_scalar_ecc_chip.rs)
struct NativeScalarEccContext : EccScalarOps<NativeScalarEccContext, AssignedValue> {
    typedef EccScalarOps<NativeScalarEccContext, AssignedValue> Base;
    NativeScalarEccContext(const IntegerContext& ic, const CurveParams& cp, size_t prefix) : Base(ic, cp, prefix) {}
    static NativeScalarEccContext new_with_select_chip(const IntegerContext& ic, const CurveParams& cp) {
        return NativeScalarEccContext(ic, cp, 0);
    }
    static NativeScalarEccContext new_without_select_chip(const IntegerContext& ic, const CurveParams& cp) {
        return NativeScalarEccContext(ic, cp, (size_t)-1);
    }

    // ParallelClone (native_scalar_ecc_chip.rs:50-90)
    void apply_offset_diff(const Offset& d) {
        base.ctx->base_offset += d.base_offset_diff;
        base.ctx->range_offset += d.range_offset_diff;
        base.ctx->select_offset += d.select_offset_diff;
    }
    NativeScalarEccContext clone_with_offset(const Offset& d) const {
        auto c = std::make_shared<Context>(base.ctx->clone_without_permutation());
        c->base_offset += d.base_offset_diff;
        c->range_offset += d.range_offset_diff;
        c->select_offset += d.select_offset_diff;
        NativeScalarEccContext r(IntegerContext(c, base.info), curve, msm_prefix);
        r.n_threads = n_threads;
        return r;
    }
    NativeScalarEccContext clone_without_offset() const { return clone_with_offset(Offset()); }
    Offset offset() const {
        Offset o;
        o.base_offset_diff = base.ctx->base_offset;
        o.range_offset_diff = base.ctx->range_offset;
        o.select_offset_diff = base.ctx->select_offset;
        return o;
    }
    void merge(NativeScalarEccContext& other) {
        Records& record = base.ctx->records;
        Records& record_other = other.base.ctx->records;
        record.permutations.insert(record.permutations.end(), record_other.permutations.begin(),
                                   record_other.permutations.end());
        record_other.permutations.clear();
        record.base_height = std::max(record.base_height, record_other.base_height);
        record.range_height = std::max(record.select_height, record_other.range_height);  // sic (quirk Q3)
        record.select_height = std::max(record.select_height, record_other.select_height);
    }
    template <class Fn>
    void run_windows(std::vector<NativeScalarEccContext>& ops, Fn fn);

    // native_scalar_ecc_chip.rs:97-171 with WINDOW_SIZE = 1
    std::vector<AssignedCondition> decompose_scalar(const AssignedValue& s) {
        Fr one = Fr::one();
        Fr two = one + one;
        Fr four = two + two;
        std::vector<AssignedCondition> bits;
        BigUint s_bn = s.val.to_bn();
        AssignedValue v = s;
        uint64_t num_bits = curve.scalar_num_bits;
        for (uint64_t i = 0; i < num_bits / 2; i++) {
            Fr b0v = s_bn.bit(i * 2) ? Fr::one() : Fr::zero();
            Fr b1v = s_bn.bit(i * 2 + 1) ? Fr::one() : Fr::zero();
            AssignedCondition b0 = base.ctx->assign_bit(b0v);
            AssignedCondition b1 = base.ctx->assign_bit(b1v);
            Fr v_next = Fr::from_bn(s_bn >> (i * 2 + 2));
            auto cells = base.ctx->one_line_with_last({pr(v_next, four), pr(b1.v, two), pr(b0.v, one)}, pr(v, -one),
                                                      nullptr, {}, nullptr);
            v = cells.first[0];
            bits.push_back(b0);
            bits.push_back(b1);
        }
        if (num_bits % 2 == 1) {
            base.ctx->assert_bit(v);
            bits.push_back(AssignedCondition(v));
        } else {
            base.ctx->assert_constant(v, Fr::zero());
        }
        // WINDOW_SIZE == 1: NUM_BITS % 1 == 0, no padding
        std::vector<AssignedCondition> res(bits.rbegin(), bits.rend());
        return res;
    }
    size_t get_and_increase_msm_prefix() {  // :173-178
        size_t ret = msm_prefix;
        if (!(ret < MSM_LIMIT)) throw PanicError("msm prefix limit");
        msm_prefix += MSM_PREFIX_OFFSET;
        return ret;
    }
    AssignedValue ecc_bisec_scalar(const AssignedCondition& cond, const AssignedValue& a, const AssignedValue& b) {
        return base.ctx->bisec(cond, a, b);
    }
    AssignedValue ecc_assign_constant_zero_scalar() { return base.ctx->assign_constant(Fr::zero()); }
};

// GeneralScalarEccContext<C, N> (context.rs:215-239, circuit/general_scalar_ecc_chip.rs): two integer contexts - base
// field and scalar field of C - over one shared Context; AssignedScalar = AssignedInteger<C::Scalar, N>
struct GeneralScalarEccContext : EccScalarOps<GeneralScalarEccContext, AssignedInteger> {
    typedef EccScalarOps<GeneralScalarEccContext, AssignedInteger> Base;
    IntegerContext scalar;   // scalar_integer_ctx
    GeneralScalarEccContext(const IntegerContext& base_ic, const IntegerContext& scalar_ic, const CurveParams& cp, size_t prefix = 0)
        : Base(base_ic, cp, prefix), scalar(scalar_ic) {}

    // ParallelClone (general_scalar_ecc_chip.rs:43-91)
    void apply_offset_diff(const Offset& d) {
        base.ctx->base_offset += d.base_offset_diff;
        base.ctx->range_offset += d.range_offset_diff;
        base.ctx->select_offset += d.select_offset_diff;
    }
    GeneralScalarEccContext clone_with_offset(const Offset& d) const {
        auto c = std::make_shared<Context>(base.ctx->clone_without_permutation());
        c->base_offset += d.base_offset_diff;
        c->range_offset += d.range_offset_diff;
        c->select_offset += d.select_offset_diff;
        GeneralScalarEccContext r(IntegerContext(c, base.info), IntegerContext(c, scalar.info), curve, msm_prefix);
        r.n_threads = n_threads;
        return r;
    }
    GeneralScalarEccContext clone_without_offset() const { return clone_with_offset(Offset()); }
    Offset offset() const {
        Offset o;
        o.base_offset_diff = base.ctx->base_offset;
        o.range_offset_diff = base.ctx->range_offset;
        o.select_offset_diff = base.ctx->select_offset;
        return o;
    }
    void merge(GeneralScalarEccContext& other) {
        Records& record = base.ctx->records;
        Records& record_other = other.base.ctx->records;
        record.permutations.insert(record.permutations.end(), record_other.permutations.begin(), record_other.permutations.end());
        record_other.permutations.clear();
        record.base_height = std::max(record.base_height, record_other.base_height);
        record.range_height = std::max(record.select_height, record_other.range_height);  // sic (quirk Q3), general_scalar_ecc_chip.rs:88
        record.select_height = std::max(record.select_height, record_other.select_height);
    }
    template <class Fn>
    void run_windows(std::vector<GeneralScalarEccContext>& ops, Fn fn) {
        for (size_t k = 0; k < ops.size(); k++) fn(k);   // (the oracle keeps this context single-threaded)
    }

    // general_scalar_ecc_chip.rs:96-147 with WINDOW_SIZE = 1
    std::vector<AssignedCondition> decompose_scalar(const AssignedInteger& s_in) {
        Fr zero = Fr::zero(), one = Fr::one();
        Fr two = one + one;
        Fr two_inv = two.inv_or_zero();   // two.invert().unwrap()
        AssignedInteger s = scalar.reduce(s_in);
        std::vector<AssignedCondition> bits;
        for (auto& l : s.limbs_le) {
            BigUint v = l.val.to_bn();
            AssignedValue rest = l;
            for (uint64_t j = 0; j < scalar.info->limb_bits; j++) {
                AssignedCondition b = base.ctx->assign_bit(v.bit(j) ? Fr::one() : Fr::zero());
                Fr nv = (rest.val - b.v.val) * two_inv;
                rest = base.ctx->one_line_with_last({pr(rest, -one), pr(b.v, one)}, pr(nv, two), nullptr, {}, nullptr).second;
                bits.push_back(b);
            }
            base.ctx->assert_constant(rest, zero);
        }
        // WINDOW_SIZE == 1: no padding
        return std::vector<AssignedCondition>(bits.rbegin(), bits.rend());
    }
    size_t get_and_increase_msm_prefix() {  // :149-154
        size_t ret = msm_prefix;
        if (!(ret < MSM_LIMIT)) throw PanicError("msm prefix limit");
        msm_prefix += MSM_PREFIX_OFFSET;
        return ret;
    }
    AssignedInteger ecc_bisec_scalar(const AssignedCondition& cond, const AssignedInteger& a, const AssignedInteger& b) {
        return scalar.bisec_int(cond, a, b);   // :156-163
    }
    AssignedInteger ecc_assign_constant_zero_scalar() { return scalar.assign_int_constant(BigUint(0)); }   // :165-168
};

}  // namespace h2o

#include <thread>
#include <atomic>
#include <mutex>
namespace h2o {
template <class Fn>
void NativeScalarEccContext::run_windows(std::vector<NativeScalarEccContext>& ops, Fn fn) {
    if (n_threads <= 1 || ops.size() < 2) {
        for (size_t k = 0; k < ops.size(); k++) fn(k);
        return;
    }
    RecordsInner& inner = *base.ctx->records.inner;
    inner.frozen = true;
    std::atomic<size_t> next(0);
    std::atomic<bool> failed(false);
    std::vector<std::thread> th;
    std::string err;
    bool unsafe = false;
    UnsafeError::Kind ukind = UnsafeError::AddSameOrNegPoint;
    std::mutex* mu = new std::mutex();
    for (int t = 0; t < n_threads; t++)
        th.emplace_back([&]() {
            for (;;) {
                size_t k = next.fetch_add(1);
                if (k >= ops.size() || failed.load()) break;
                try {
                    fn(k);
                } catch (UnsafeError& e) {
                    std::lock_guard<std::mutex> g(*mu);
                    failed = true;
                    unsafe = true;
                    ukind = e.kind;
                } catch (std::exception& e) {
                    std::lock_guard<std::mutex> g(*mu);
                    failed = true;
                    err = e.what();
                }
            }
        });
    for (auto& t : th) t.join();
    delete mu;
    inner.frozen = false;
    if (failed) {
        if (unsafe) throw UnsafeError(ukind);
        throw PanicError(err);
    }
}
}  // namespace h2o
