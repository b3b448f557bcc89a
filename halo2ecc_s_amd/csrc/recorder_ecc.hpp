// Recording implementation of the reference's ECC / MSM surface:
//   EccChipBaseOps  src/circuit/ecc_chip.rs:438-1009
//   EccChipScalarOps (msm_batch_on_group_non_zero_with_select_chip, msm_unsafe)  src/circuit/ecc_chip.rs:223-416
//   NativeScalarEccContext glue  src/circuit/native_scalar_ecc_chip.rs:27-194, src/context.rs:190-207
// Same method names / argument meaning; values are replaced by input slots, handles carry cell refs.
//
// Where the reference iterates sequentially over independent items (points, scalars, candidate groups)
// or forks per MSM window (ecc_chip.rs:289-352), the recorder forks strands; the rows, fixed cells and
// permutations come out exactly as the sequential reference would have produced them because every
// strand consumes the same Offset.
#pragma once
#include "recorder.hpp"

namespace h2e {

static const size_t MSM_PREFIX_OFFSET = 1u << 20;               // ecc_chip.rs:20
static const size_t MSM_LIMIT = (1u << 8) * MSM_PREFIX_OFFSET;  // ecc_chip.rs:21

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
    HBig b;                    // C::b()
    HBig gen_x, gen_y;         // C::generator()
    uint32_t scalar_num_bits;  // PrimeField::NUM_BITS of C::Scalar
    bool scalar_modulus_odd;
};
inline CurveParams bn256_g1_params() { return CurveParams{HBig(3), HBig(1), HBig(2), 254, true}; }
inline CurveParams bls12_381_g1_params() {
    return CurveParams{HBig(4),
                       HBig::from_hex("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"),
                       HBig::from_hex("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1"),
                       255, true};
}

// A point entering from the instance input vector: slots (x, y, z-flag).
struct PointInput {
    uint32_t x_slot, y_slot, z_slot;
    bool strided;
};

// NativeScalarEccContext<C> (context.rs:190-207): integer context over C::Base + msm prefix
struct NativeScalarEccContext {
    Recorder& ctx;  // the shared Rc<RefCell<Context>> + IntegerContext
    CurveParams curve;
    size_t msm_prefix;  // usize::MAX => no select chip

    NativeScalarEccContext(Recorder& r, const CurveParams& c, size_t prefix = 0) : ctx(r), curve(c), msm_prefix(prefix) {}
    bool has_select_chip() const { return msm_prefix != (size_t)-1; }
    size_t get_and_increase_msm_prefix() {  // native_scalar_ecc_chip.rs:173-178
        size_t ret = msm_prefix;
        if (!(ret < MSM_LIMIT)) throw std::runtime_error("msm prefix limit");
        msm_prefix += MSM_PREFIX_OFFSET;
        return ret;
    }

    // ---- EccChipBaseOps ----
    // curve equation rows shared by assign_point / assign_non_zero_point (ecc_chip.rs:472-480, :502-510)
    void curve_rhs(const AssignedInteger& x, const AssignedInteger& y, AssignedInteger& y2, AssignedInteger& right) {
        AssignedInteger b = ctx.assign_int_constant(curve.b);
        y2 = ctx.int_square(y);
        AssignedInteger x2 = ctx.int_square(x);
        AssignedInteger x3 = ctx.int_mul(x2, x);
        right = ctx.int_add(x3, b);
    }
    // ecc_chip.rs:458-487
    AssignedPoint assign_point(const PointInput& in) {
        AssignedInteger x = ctx.assign_w(in.x_slot, in.strided);
        AssignedInteger y = ctx.assign_w(in.y_slot, in.strided);
        AssignedCondition z = ctx.assign_bit(in.z_slot, in.strided);
        AssignedInteger y2, right;
        curve_rhs(x, y, y2, right);
        AssignedCondition eq = ctx.is_int_equal(y2, right);
        AssignedCondition eq_or_identity = ctx.or_(eq, z);
        ctx.assert_true(eq_or_identity);
        return AssignedPoint{x, y, z};
    }
    // ecc_chip.rs:489-512
    AssignedNonZeroPoint assign_non_zero_point(uint32_t x_slot, uint32_t y_slot) {
        AssignedInteger x = ctx.assign_w(x_slot);
        AssignedInteger y = ctx.assign_w(y_slot);
        AssignedInteger y2, right;
        curve_rhs(x, y, y2, right);
        ctx.assert_int_equal(y2, right);
        return AssignedNonZeroPoint{x, y};
    }
    // ecc_chip.rs:531-560
    AssignedPoint bisec_point(const AssignedCondition& cond, const AssignedPoint& a, const AssignedPoint& b) {
        AssignedInteger x = ctx.bisec_int(cond, a.x, b.x);
        AssignedInteger y = ctx.bisec_int(cond, a.y, b.y);
        AssignedCondition z = ctx.bisec_cond(cond, a.z, b.z);
        return AssignedPoint{x, y, z};
    }
    AssignedCurvature bisec_curvature(const AssignedCondition& cond, const AssignedCurvature& a, const AssignedCurvature& b) {
        AssignedInteger v = ctx.bisec_int(cond, a.v, b.v);
        AssignedCondition z = ctx.bisec_cond(cond, a.z, b.z);
        return AssignedCurvature{v, z};
    }
    // ecc_chip.rs:580-604
    AssignedPoint lambda_to_point(const AssignedCurvature& lambda, const AssignedPoint& a, const AssignedPoint& b) {
        const AssignedInteger& l = lambda.v;
        AssignedInteger l_square = ctx.int_square(l);
        AssignedInteger t = ctx.int_sub(l_square, a.x);
        AssignedInteger cx = ctx.int_sub(t, b.x);
        AssignedInteger t2 = ctx.int_sub(a.x, cx);
        t2 = ctx.int_mul(t2, l);
        AssignedInteger cy = ctx.int_sub(t2, a.y);
        return AssignedPoint{cx, cy, lambda.z};
    }
    // ecc_chip.rs:606-628
    AssignedPoint ecc_add(const AssignedPointWithCurvature& a, const AssignedPoint& b) {
        AssignedInteger diff_x = ctx.int_sub(a.x, b.x);
        AssignedInteger diff_y = ctx.int_sub(a.y, b.y);
        auto dv = ctx.int_div(diff_y, diff_x);
        AssignedCondition x_eq = dv.first;
        AssignedCondition y_eq = ctx.is_int_zero(diff_y);
        AssignedCondition eq = ctx.and_(x_eq, y_eq);
        AssignedCurvature tangent{dv.second, x_eq};
        AssignedCurvature lambda = bisec_curvature(eq, a.curvature, tangent);
        AssignedPoint a_p = a.to_point();
        AssignedPoint p = lambda_to_point(lambda, a_p, b);
        p = bisec_point(a.z, b, p);
        p = bisec_point(b.z, a_p, p);
        return p;
    }
    // ecc_chip.rs:644-658
    void ecc_assert_equal(const AssignedPoint& a, const AssignedPoint& b) {
        AssignedCondition eq_x = ctx.is_int_equal(a.x, b.x);
        AssignedCondition eq_y = ctx.is_int_equal(a.y, b.y);
        AssignedCondition eq_z = ctx.xnor(a.z, b.z);
        AssignedCondition eq_xy = ctx.and_(eq_x, eq_y);
        AssignedCondition eq_xyz = ctx.and_(eq_xy, eq_z);
        AssignedCondition is_both_identity = ctx.and_(a.z, b.z);
        AssignedCondition eq = ctx.or_(eq_xyz, is_both_identity);
        ctx.assert_true(eq);
    }
    // ecc_chip.rs:695-708
    AssignedPointWithCurvature to_point_with_curvature(const AssignedPoint& a) {
        AssignedInteger x_square = ctx.int_square(a.x);
        AssignedInteger numerator = ctx.int_mul_small_constant(x_square, 3);
        AssignedInteger denominator = ctx.int_mul_small_constant(a.y, 2);
        auto zv = ctx.int_div(numerator, denominator);
        return AssignedPointWithCurvature{a.x, a.y, a.z, AssignedCurvature{zv.second, zv.first}};
    }
    // ---- the rest of the public EccChipBaseOps surface (not reached by the BASELINE configs) ----
    // ecc_chip.rs:441-456: a point known when the circuit is built (x, y canonical; identity -> (0, 0, z = 1))
    AssignedPoint assign_constant_point(const HBig& x, const HBig& y, bool is_identity) {
        AssignedInteger ax = ctx.assign_int_constant(is_identity ? HBig(0) : x);
        AssignedInteger ay = ctx.assign_int_constant(is_identity ? HBig(0) : y);
        AssignedValue z = ctx.assign_constant_u64(is_identity ? 1 : 0);
        return AssignedPoint{ax, ay, AssignedCondition{z}};
    }
    // ecc_chip.rs:514-529
    AssignedPointWithCurvature assign_identity() {
        AssignedInteger zero = ctx.assign_int_constant(HBig(0));
        AssignedValue one = ctx.assign_constant_u64(1);
        return AssignedPointWithCurvature{zero, zero, AssignedCondition{one}, AssignedCurvature{zero, AssignedCondition{one}}};
    }
    // ecc_chip.rs:562-578
    AssignedPointWithCurvature bisec_point_with_curvature(const AssignedCondition& cond, const AssignedPointWithCurvature& a,
                                                          const AssignedPointWithCurvature& b) {
        AssignedInteger x = ctx.bisec_int(cond, a.x, b.x);
        AssignedInteger y = ctx.bisec_int(cond, a.y, b.y);
        AssignedCondition z = ctx.bisec_cond(cond, a.z, b.z);
        AssignedCurvature c = bisec_curvature(cond, a.curvature, b.curvature);
        return AssignedPointWithCurvature{x, y, z, c};
    }
    // ecc_chip.rs:630-642
    AssignedPoint ecc_double(const AssignedPointWithCurvature& a) {
        AssignedPoint a_p = a.to_point();
        AssignedPoint p = lambda_to_point(a.curvature, a_p, a_p);
        p.z = ctx.bisec_cond(a.z, a.z, p.z);
        return p;
    }
    // ecc_chip.rs:660-666
    AssignedPoint ecc_neg(const AssignedPoint& a) { return AssignedPoint{a.x, ctx.int_neg(a.y), a.z}; }
    // ecc_chip.rs:668-675
    AssignedPoint ecc_reduce(const AssignedPoint& a) {
        AssignedInteger x = ctx.reduce(a.x);
        AssignedInteger y = ctx.reduce(a.y);
        AssignedPointWithCurvature identity = assign_identity();
        return bisec_point(a.z, identity.to_point(), AssignedPoint{x, y, a.z});
    }
    // ecc_chip.rs:677-693
    AssignedPointWithCurvature ecc_reduce_with_curvature(const AssignedPoint& a_in) {
        AssignedPoint a = ecc_reduce(a_in);
        AssignedInteger x_square = ctx.int_square(a.x);
        AssignedInteger numerator = ctx.int_mul_small_constant(x_square, 3);
        AssignedInteger denominator = ctx.int_mul_small_constant(a.y, 2);
        auto zv = ctx.int_div(numerator, denominator);
        AssignedInteger v = ctx.reduce(zv.second);
        return AssignedPointWithCurvature{a.x, a.y, a.z, AssignedCurvature{v, zv.first}};
    }
    // ecc_chip.rs:710-732: [x0 + x1 B, x2 + y0 B, y1 + y2 B] with B = 2^108 (3-limb fields)
    std::vector<AssignedValue> ecc_encode(const AssignedPoint& p_in) {
        AssignedPoint p = ecc_reduce(p_in);
        auto shift_add = [&](uint32_t a, uint32_t b) {
            H2EOp op = ctx.new_op(H2E_OP_SHIFT_ADD);
            op.refs[0] = a;
            op.refs[1] = b;
            ctx.push(op);
            size_t row = ctx.base_line({Recorder::A(a, ctx.id_one), Recorder::A(b, ctx.id_limb_coeff[1])}, Recorder::U(ctx.id_neg_one));
            return AssignedValue{ctx.mk(0, 4, row)};
        };
        AssignedValue s0 = shift_add(p.x.limbs_le[0], p.x.limbs_le[1]);
        AssignedValue s1 = shift_add(p.x.limbs_le[2], p.y.limbs_le[0]);
        AssignedValue s2 = shift_add(p.y.limbs_le[1], p.y.limbs_le[2]);
        return {s0, s1, s2};
    }
    void cache_cell(uint32_t ref, size_t offset, size_t g, size_t sc) {
        H2EOp op = ctx.new_op(H2E_OP_CACHE_INT, 1);
        op.refs[0] = ref;
        ctx.push(op);
        ctx.assign_cache_value(ref, offset, g, sc);
    }
    // ecc_chip.rs:779-788
    void assign_cache_point(const AssignedPointWithCurvature& p, size_t g, size_t sc) {
        size_t i = 0;
        assign_cache_integer(p.x, sc, g, i);
        assign_cache_integer(p.y, sc, g, i);
        cache_cell(p.z.v.ref, i, g, sc);
        i += 1;
        assign_cache_integer(p.curvature.v, sc, g, i);
        cache_cell(p.curvature.z.v.ref, i, g, sc);
    }
    // ecc_chip.rs:790-812.  The reference is handed the already chosen candidate (a value copy); here the candidate is
    // picked on the device by the value of the index cell `sc` from the candidates' cells.
    AssignedPointWithCurvature assign_selected_point(const std::vector<AssignedPointWithCurvature>& candidates, const AssignedValue& sc, size_t g) {
        const int L = ctx.fp.limbs;
        const uint32_t nc = 3 * (L + 1) + 2;
        if (nc > 255 || candidates.empty() || candidates.size() > 256) throw std::runtime_error("assign_selected_point: bad candidate table");
        uint32_t table = (uint32_t)ctx.aux.size();
        for (auto& q : candidates) {
            for (const AssignedInteger* a : {&q.x, &q.y}) {
                for (int j = 0; j < L; j++) ctx.aux.push_back(a->limbs_le[j]);
                ctx.aux.push_back(a->native);
            }
            ctx.aux.push_back(q.z.v.ref);
            for (int j = 0; j < L; j++) ctx.aux.push_back(q.curvature.v.limbs_le[j]);
            ctx.aux.push_back(q.curvature.v.native);
            ctx.aux.push_back(q.curvature.z.v.ref);
        }
        H2EOp op = ctx.new_op(H2E_OP_SELECT_POINT, table, (uint16_t)(nc << 8));
        op.refs[0] = sc.ref;
        ctx.push(op);
        size_t i = 0;
        auto sel_int = [&]() {
            AssignedInteger t;
            for (int j = 0; j < L; j++) t.limbs_le[j] = ctx.assign_selected_value(i++, g, sc.ref);
            t.native = ctx.assign_selected_value(i++, g, sc.ref);
            t.times = 1;
            return t;
        };
        AssignedInteger x = sel_int(), y = sel_int();
        AssignedValue z{ctx.assign_selected_value(i++, g, sc.ref)};
        AssignedInteger cv = sel_int();
        AssignedValue cz{ctx.assign_selected_value(i++, g, sc.ref)};
        return AssignedPointWithCurvature{x, y, AssignedCondition{z}, AssignedCurvature{cv, AssignedCondition{cz}}};
    }
    // ecc_chip.rs:975-982
    void ecc_assert_equal_non_zero(const AssignedNonZeroPoint& a, const AssignedNonZeroPoint& b) {
        ctx.assert_int_equal(a.x, b.x);
        ctx.assert_int_equal(a.y, b.y);
    }

    // ecc_chip.rs:734-751
    void assign_cache_integer(const AssignedInteger& p, size_t sc, size_t g, size_t& offset) {
        if (p.times != 1) throw std::runtime_error("assign_cache_integer: times != 1");
        H2EOp op = ctx.new_op(H2E_OP_CACHE_INT);
        ctx.put_int(op, 0, p);
        ctx.push(op);
        for (int j = 0; j < ctx.fp.limbs; j++) {
            ctx.assign_cache_value(p.limbs_le[j], offset, g, sc);
            offset += 1;
        }
        ctx.assign_cache_value(p.native, offset, g, sc);
        offset += 1;
    }
    // ecc_chip.rs:969-973
    void assign_cache_point_non_zero(const AssignedNonZeroPoint& p, size_t g, size_t sc) {
        size_t i = 0;
        assign_cache_integer(p.x, sc, g, i);
        assign_cache_integer(p.y, sc, g, i);
    }
    // ecc_chip.rs:814-838
    // (the expect() calls name the hint slot of each result's value while full value hints are on: recorder.hpp)
    AssignedNonZeroPoint lambda_to_point_non_zero(const AssignedInteger& l, const AssignedNonZeroPoint& a,
                                                  const AssignedNonZeroPoint& b) {
        ctx.expect(H2E_HINT_LAMBDA2);
        AssignedInteger l_square = ctx.int_square(l);
        AssignedInteger t = ctx.int_sub(l_square, a.x);
        ctx.expect(H2E_HINT_XC);
        AssignedInteger cx = ctx.int_sub(t, b.x);
        ctx.expect(H2E_HINT_T2);
        AssignedInteger t2 = ctx.int_sub(a.x, cx);
        ctx.expect(H2E_HINT_T2L);
        t2 = ctx.int_mul(t2, l);
        ctx.expect(H2E_HINT_YC);
        AssignedInteger cy = ctx.int_sub(t2, a.y);
        return AssignedNonZeroPoint{cx, cy};
    }
    // ecc_chip.rs:840-858 — a failing instance reports H2E_STATUS_RETRY_ADD_SAME_OR_NEG
    AssignedNonZeroPoint ecc_add_unsafe(const AssignedNonZeroPoint& a, const AssignedNonZeroPoint& b) {
        ctx.begin_ecc_op();
        ctx.expect(H2E_HINT_AUX0);
        AssignedInteger diff_x = ctx.int_sub(a.x, b.x);
        AssignedInteger diff_y = ctx.int_sub(a.y, b.y);
        auto dv = ctx.int_div(diff_y, diff_x);
        ctx.try_assert_false(dv.first, H2E_FLAG_UNSAFE_ADD);
        AssignedNonZeroPoint r = lambda_to_point_non_zero(dv.second, a, b);
        ctx.end_ecc_op();
        return r;
    }
    // ecc_chip.rs:860-882 — a failing instance reports H2E_STATUS_RETRY_ADD_IDENTITY
    AssignedNonZeroPoint ecc_double_unsafe(const AssignedNonZeroPoint& a) {
        ctx.begin_ecc_op();
        ctx.expect(H2E_HINT_AUX0);
        AssignedInteger x_square = ctx.int_square(a.x);
        AssignedInteger numerator = ctx.int_mul_small_constant(x_square, 3);
        ctx.expect(H2E_HINT_AUX1);
        AssignedInteger denominator = ctx.int_mul_small_constant(a.y, 2);
        auto zv = ctx.int_div(numerator, denominator);
        ctx.try_assert_false(zv.first, H2E_FLAG_UNSAFE_DBL);
        AssignedNonZeroPoint r = lambda_to_point_non_zero(zv.second, a, a);
        ctx.end_ecc_op();
        return r;
    }
    AssignedNonZeroPoint ecc_neg_non_zero(const AssignedNonZeroPoint& a) {  // :884-889
        return AssignedNonZeroPoint{a.x, ctx.int_neg(a.y)};
    }
    AssignedNonZeroPoint ecc_reduce_non_zero(const AssignedNonZeroPoint& a) {  // :891-899
        AssignedInteger x = ctx.reduce(a.x);
        AssignedInteger y = ctx.reduce(a.y);
        return AssignedNonZeroPoint{x, y};
    }
    // ecc_chip.rs:984-997
    AssignedPoint ecc_non_zero_point_downgrade(const AssignedNonZeroPoint& a) {
        AssignedValue zero = ctx.assign_constant_u64(0);
        return AssignedPoint{a.x, a.y, AssignedCondition{zero}};
    }
    // ecc_chip.rs:999-1008
    AssignedNonZeroPoint ecc_bisec_to_non_zero_point(const AssignedPoint& a, const AssignedNonZeroPoint& b) {
        AssignedInteger x = ctx.bisec_int(a.z, b.x, a.x);
        AssignedInteger y = ctx.bisec_int(a.z, b.y, a.y);
        return AssignedNonZeroPoint{x, y};
    }
    // pick_candidate_non_zero's in-circuit part (ecc_chip.rs:941-948): index = sum bit_i 2^i
    AssignedValue pick_index(const std::vector<AssignedCondition>& group_bits, bool preselected = false) {
        size_t k = group_bits.size();
        if (k > 5) throw std::runtime_error("pick_index: more than 5 bits");
        H2EOp op = ctx.new_op(H2E_OP_PICK_INDEX, (uint32_t)k, preselected ? H2E_FLAG_PRESELECTED : 0);
        for (size_t i = 0; i < k; i++) op.refs[i] = group_bits[i].v.ref;
        ctx.push(op);
        Recorder::Col cols[4];
        if (k < 5) {  // sum_with_constant_in_one_line (base_chip.rs:110-132)
            for (size_t i = 0; i < k; i++) cols[i] = Recorder::A(group_bits[i].v.ref, ctx.id_pow2[i]);
            return AssignedValue{ctx.mk(0, 4, ctx.base_line(cols, (int)k, Recorder::U(ctx.id_neg_one), 0, 0, 0, 0))};
        }
        for (size_t i = 0; i < 4; i++) cols[i] = Recorder::A(group_bits[i].v.ref, ctx.id_pow2[i]);
        uint32_t acc = ctx.mk(0, 4, ctx.base_line(cols, 4, Recorder::U(ctx.id_neg_one), 0, 0, 0, 0));
        size_t row = ctx.base_line({Recorder::A(group_bits[4].v.ref, ctx.id_pow2[4]), Recorder::A(acc, ctx.id_one)},
                                   Recorder::U(ctx.id_neg_one));
        return AssignedValue{ctx.mk(0, 4, row)};
    }
    // pick_candidate_non_zero + assign_selected_point_non_zero (ecc_chip.rs:935-967): the candidate is
    // picked *by value* on the device; `table_aux` is the aux offset of the group's candidate ref table.
    // sel_entry >= 0: a select pre-kernel picks this point into selection-buffer entry sel_entry (+ strand * stride)
    AssignedNonZeroPoint pick_and_select(uint32_t table_aux, const std::vector<AssignedCondition>& group_bits, size_t g,
                                         int64_t sel_entry = -1) {
        AssignedValue index = pick_index(group_bits, sel_entry >= 0);
        H2EOp op = ctx.new_op(H2E_OP_SELECT_POINT, table_aux, sel_entry >= 0 ? H2E_FLAG_PRESELECTED : 0);
        op.refs[0] = index.ref;
        if (sel_entry >= 0) op.refs[1] = (uint32_t)sel_entry;
        ctx.push(op);
        AssignedNonZeroPoint r;
        size_t i = 0;
        for (int which = 0; which < 2; which++) {
            AssignedInteger& t = which == 0 ? r.x : r.y;
            for (int j = 0; j < ctx.fp.limbs; j++) t.limbs_le[j] = ctx.assign_selected_value(i++, g, index.ref);
            t.native = ctx.assign_selected_value(i++, g, index.ref);
            t.times = 1;
        }
        return r;
    }

    // ---- EccChipScalarOps for the native scalar field ----
    // decompose_scalar::<1> (native_scalar_ecc_chip.rs:97-171): returns bit cells MSB first
    std::vector<AssignedCondition> decompose_scalar(const AssignedValue& s) {
        uint32_t nbits = curve.scalar_num_bits;
        H2EOp op = ctx.new_op(H2E_OP_DECOMPOSE_NATIVE, nbits);
        op.refs[0] = s.ref;
        ctx.push(op);
        std::vector<AssignedCondition> bits;
        uint32_t v = s.ref;
        for (uint32_t i = 0; i < nbits / 2; i++) {
            size_t r0 = ctx.base_line({Recorder::U(ctx.id_one), Recorder::U(ctx.id_zero)}, Recorder::none(), ctx.id_neg_one);
            size_t r1 = ctx.base_line({Recorder::U(ctx.id_one), Recorder::U(ctx.id_zero)}, Recorder::none(), ctx.id_neg_one);
            uint32_t b0 = ctx.mk(0, 0, r0), b1 = ctx.mk(0, 0, r1);
            size_t r2 = ctx.base_line({Recorder::U(ctx.id_four), Recorder::A(b1, ctx.id_two), Recorder::A(b0, ctx.id_one)},
                                      Recorder::A(v, ctx.id_neg_one));
            v = ctx.mk(0, 0, r2);
            bits.push_back(AssignedCondition{AssignedValue{b0}});
            bits.push_back(AssignedCondition{AssignedValue{b1}});
        }
        if (nbits % 2 == 1) {  // assert_bit(v) (base_chip.rs:381-390)
            ctx.base_line({Recorder::A(v, ctx.id_one), Recorder::A(v, ctx.id_zero)}, Recorder::none(), ctx.id_neg_one);
            bits.push_back(AssignedCondition{AssignedValue{v}});
        } else {  // assert_constant(v, 0); the engine op already checked the value
            ctx.base_line({Recorder::A(v, ctx.id_neg_one)}, Recorder::none(), 0, 0, 0, ctx.id_zero);
        }
        return std::vector<AssignedCondition>(bits.rbegin(), bits.rend());
    }
    AssignedValue ecc_bisec_scalar(const AssignedCondition& cond, const AssignedValue& a, const AssignedValue& b) {
        return ctx.bisec(cond, a, b);
    }
    AssignedValue ecc_assign_constant_zero_scalar() { return ctx.assign_constant_u64(0); }

    // ---- EccChipScalarOps for a non-native scalar field (GeneralScalarEccContext, general_scalar_ecc_chip.rs:93-168) ----
    // scalar_field >= 0: AssignedScalar = AssignedInteger<C::Scalar, N> of the *other* integer context (context.rs:215-239);
    // -1: AssignedValue (NativeScalarEccContext), carried in AssignedInteger::native so that one MSM body serves both.
    int scalar_field = -1;
    int scalar_limbs() const { return scalar_field < 0 ? 0 : field_pair_of(scalar_field).limbs; }
    AssignedInteger scalar_param(const AssignedInteger& s) {
        AssignedInteger r = s;
        for (int i = 0; i < scalar_limbs(); i++) r.limbs_le[i] = ctx.param(s.limbs_le[i]);
        r.native = ctx.param(s.native);
        return r;
    }
    AssignedInteger scalar_zero() {   // ecc_assign_constant_zero_scalar (native :188-192, general :163-167)
        AssignedInteger r;
        if (scalar_field < 0) {
            r.native = ecc_assign_constant_zero_scalar().ref;
            return r;
        }
        int base_field = ctx.fp.id;
        ctx.use_field(scalar_field);
        r = ctx.assign_int_constant(HBig(0));
        ctx.use_field(base_field);
        return r;
    }
    AssignedInteger scalar_bisec(const AssignedCondition& cond, const AssignedInteger& a, const AssignedInteger& b) {
        AssignedInteger r;
        if (scalar_field < 0) {
            r.native = ecc_bisec_scalar(cond, AssignedValue{a.native}, AssignedValue{b.native}).ref;
            return r;
        }
        return ctx.bisec_int_limbs(cond, a, b, scalar_limbs());   // scalar_integer_ctx.bisec_int (general :154-161)
    }
    // decompose_scalar::<1>: bit cells, MSB (window 0) first
    std::vector<AssignedCondition> scalar_decompose(const AssignedInteger& s) {
        if (scalar_field < 0) return decompose_scalar(AssignedValue{s.native});
        // general_scalar_ecc_chip.rs:96-147: reduce (a no-op for the bisected scalars: times == 1), then limb by limb
        if (s.times != 1) throw std::runtime_error("decompose_scalar: scalar not reduced (times != 1)");
        std::vector<AssignedCondition> bits;
        for (int l = 0; l < scalar_limbs(); l++) {
            std::vector<AssignedCondition> b = ctx.decompose_limb(s.limbs_le[l], LIMB_BITS);
            bits.insert(bits.end(), b.begin(), b.end());
        }
        return std::vector<AssignedCondition>(bits.rbegin(), bits.rend());
    }

    void push_point_refs(std::vector<uint32_t>& v, const AssignedNonZeroPoint& p) const {
        for (int j = 0; j < ctx.fp.limbs; j++) v.push_back(p.x.limbs_le[j]);
        v.push_back(p.x.native);
        for (int j = 0; j < ctx.fp.limbs; j++) v.push_back(p.y.limbs_le[j]);
        v.push_back(p.y.native);
    }

    struct MsmInputs {
        uint32_t r1_x, r1_y, r2_x, r2_y;  // blinding points `generator * Scalar::rand()` made explicit (quirk Q1)
    };

    // The reference's test body (src/tests/native_scalar_ecc_chip.rs:34-47): assign_point x n, assign x n,
    // msm_unsafe, with forks where the reference loops over independent items.
    // Input layout: slots [0, 3n) = (x, y, z) per point, [3n, 4n) = scalars, then r1, r2 as given.
    // false: msm_batch_on_group_non_zero_without_select_chip (ecc_chip.rs:91-221): groups of 2 points, the candidate is
    // chosen by a bisection tree (bisec_candidate_non_zero, :913-933) instead of a select-chip lookup, no cache rows
    bool with_select = true;
    // ecc_chip.rs:901-911
    AssignedNonZeroPoint ecc_bisec_non_zero_point(const AssignedCondition& cond, const AssignedNonZeroPoint& a,
                                                  const AssignedNonZeroPoint& b) {
        AssignedInteger x = ctx.bisec_int(cond, a.x, b.x);
        AssignedInteger y = ctx.bisec_int(cond, a.y, b.y);
        return AssignedNonZeroPoint{x, y};
    }
    // ecc_chip.rs:913-933
    AssignedNonZeroPoint bisec_candidate_non_zero(const std::vector<AssignedNonZeroPoint>& candidates,
                                                  const std::vector<AssignedCondition>& group_bits) {
        std::vector<AssignedNonZeroPoint> curr = candidates;
        for (auto& bit : group_bits) {
            std::vector<AssignedNonZeroPoint> next;
            for (size_t k = 0; k < curr.size(); k += 2) {
                if (k + 1 >= curr.size()) throw std::runtime_error("bisec_candidate: odd chunk");
                next.push_back(ecc_bisec_non_zero_point(bit, curr[k + 1], curr[k]));
            }
            curr = next;
        }
        if (curr.size() != 1) throw std::runtime_error("bisec_candidate: size != 1");
        return curr[0];
    }
    // points.iter().map(|x| ctx.assign_point(x)) as a fork over the points: slots (x, y, z) per point from first_slot
    std::vector<AssignedPoint> assign_points_from_inputs(uint32_t n, uint32_t first_slot) {
        Recorder& c = ctx;
        AssignedPoint p0;
        c.fork(n, 3, [&](uint32_t k) {
            AssignedPoint p = assign_point(PointInput{first_slot + 0, first_slot + 1, first_slot + 2, true});
            if (k == 0) p0 = p;
        });
        Segment seg = c.segments[c.segments.size() - 2];
        std::vector<AssignedPoint> out;
        for (uint32_t k = 0; k < n; k++)
            out.push_back(AssignedPoint{c.strand_int(p0.x, seg, k), c.strand_int(p0.y, seg, k),
                                        AssignedCondition{AssignedValue{c.strand_ref(p0.z.v.ref, seg, k)}}});
        return out;
    }
    // the scalars of the test bodies: native `ctx.assign(x)` (tests/native_scalar_ecc_chip.rs:40-43) or
    // `scalar_integer_ctx.assign_w(x)` (tests/general_scalar_ecc_chip.rs:38-41), one input slot each from first_slot
    std::vector<AssignedInteger> assign_scalars_from_inputs(uint32_t n, uint32_t first_slot) {
        Recorder& c = ctx;
        AssignedInteger s0;
        int base_field = c.fp.id;
        if (scalar_field >= 0) c.use_field(scalar_field);
        c.fork(n, 1, [&](uint32_t k) {
            AssignedInteger s;
            if (scalar_field < 0) s.native = c.assign(first_slot, true).ref;
            else s = c.assign_w(first_slot, true);
            if (k == 0) s0 = s;
        });
        Segment seg = c.segments[c.segments.size() - 2];
        if (scalar_field >= 0) c.use_field(base_field);
        std::vector<AssignedInteger> out;
        for (uint32_t k = 0; k < n; k++) {
            AssignedInteger s = s0;
            for (int i = 0; i < scalar_limbs(); i++) s.limbs_le[i] = c.strand_ref(s0.limbs_le[i], seg, k);
            s.native = c.strand_ref(s0.native, seg, k);
            out.push_back(s);
        }
        return out;
    }
    AssignedPoint msm_unsafe_from_inputs(uint32_t n, uint32_t first_slot, const MsmInputs& mi, uint32_t gen_x_slot,
                                         uint32_t gen_y_slot) {
        std::vector<AssignedPoint> points = assign_points_from_inputs(n, first_slot);
        std::vector<AssignedInteger> scalars = assign_scalars_from_inputs(n, first_slot + 3 * n);
        return msm_unsafe(points, scalars, mi, gen_x_slot, gen_y_slot);
    }
    // EccChipScalarOps::msm_unsafe (ecc_chip.rs:373-408) on assigned points / scalars (handles = absolute cell refs).
    // The blinding points r1, r2 the reference draws inside (quirk Q1) and the generator are instance inputs.
    AssignedPoint msm_unsafe(const std::vector<AssignedPoint>& points, const std::vector<AssignedInteger>& scalars,
                             const MsmInputs& mi, uint32_t gen_x_slot, uint32_t gen_y_slot) {
        Recorder& c = ctx;
        int L = c.fp.limbs;
        uint32_t n = (uint32_t)points.size();
        if (n == 0 || scalars.size() != n) throw std::runtime_error("msm_unsafe: points / scalars mismatch");
        AssignedNonZeroPoint non_zero_p = assign_non_zero_point(gen_x_slot, gen_y_slot);
        AssignedInteger s_zero = scalar_zero();
        AssignedInteger ns0;
        AssignedNonZeroPoint np0;
        c.fork(n, 0, [&](uint32_t k) {
            AssignedPoint pk{c.param(points[k].x), c.param(points[k].y), AssignedCondition{c.param(points[k].z.v)}};
            AssignedInteger sk = scalar_param(scalars[k]);
            AssignedInteger s = scalar_bisec(pk.z, s_zero, sk);
            AssignedNonZeroPoint p = ecc_bisec_to_non_zero_point(pk, non_zero_p);
            if (k == 0) {
                ns0 = s;
                np0 = p;
            }
        });
        Segment seg_norm = c.segments[c.segments.size() - 2];
        auto point_k = [&](uint32_t k) {
            return AssignedNonZeroPoint{c.strand_int(np0.x, seg_norm, k), c.strand_int(np0.y, seg_norm, k)};
        };
        auto scalar_k = [&](uint32_t k) {
            AssignedInteger r = ns0;
            for (int i = 0; i < scalar_limbs(); i++) r.limbs_le[i] = c.strand_ref(ns0.limbs_le[i], seg_norm, k);
            r.native = c.strand_ref(ns0.native, seg_norm, k);
            return r;
        };

        // ---- msm_batch_on_group_non_zero_with_select_chip (ecc_chip.rs:223-371) ----
        if (with_select && !(n <= MSM_PREFIX_OFFSET)) throw std::runtime_error("msm: too many points");
        // ecc_reduce_non_zero(points): bisec_int results have times == 1, so no rows (ecc_chip.rs:233-236)
        AssignedNonZeroPoint rand_acc_point = assign_non_zero_point(mi.r1_x, mi.r1_y);
        AssignedNonZeroPoint rand_line_point = assign_non_zero_point(mi.r2_x, mi.r2_y);
        AssignedNonZeroPoint rand_acc_point_neg = ecc_reduce_non_zero(ecc_neg_non_zero(rand_acc_point));
        AssignedNonZeroPoint rand_line_point_neg = ecc_reduce_non_zero(ecc_neg_non_zero(rand_line_point));

        size_t best_group_size = with_select ? 5 : 2;
        size_t n_group = (n + best_group_size - 1) / best_group_size;
        size_t group_size = (n + n_group - 1) / n_group;
        size_t group_prefix = with_select ? get_and_increase_msm_prefix() : 0;
        size_t n_chunks = (n + group_size - 1) / group_size;
        size_t n_full = n / group_size;  // groups with exactly group_size points

        // candidate tables: aux[table(g) + idx * 2(L+1) + j] = absolute ref of cell j of candidate idx
        const uint32_t NC = 2 * (L + 1);
        std::vector<uint32_t> table_aux(n_chunks);
        auto write_table_entry = [&](uint32_t at, const AssignedNonZeroPoint& p) {
            for (int j = 0; j < L; j++) c.aux[at + j] = p.x.limbs_le[j];
            c.aux[at + L] = p.x.native;
            for (int j = 0; j < L; j++) c.aux[at + L + 1 + j] = p.y.limbs_le[j];
            c.aux[at + 2 * L + 1] = p.y.native;
        };
        for (size_t g = 0; g < n_chunks; g++) {
            size_t sz = std::min<size_t>(group_size, n - g * group_size);
            table_aux[g] = (uint32_t)c.aux.size();
            c.aux.resize(c.aux.size() + ((size_t)NC << sz), H2E_NO_REF);
        }
        // one group's candidate list (ecc_chip.rs:255-274); `pts` = the group's points
        auto build_group = [&](size_t group_index, const std::vector<AssignedNonZeroPoint>& pts,
                               std::vector<AssignedNonZeroPoint>& cl, const AssignedNonZeroPoint& init_even,
                               const AssignedNonZeroPoint& init_odd, bool parity_param) {
            (void)parity_param;
            const AssignedNonZeroPoint& init = (group_index % 2 == 0) ? init_even : init_odd;
            cl.clear();
            cl.push_back(init);
            if (with_select) assign_cache_point_non_zero(init, group_prefix + group_index, 0);
            for (uint32_t i = 1; i < (1u << pts.size()); i++) {
                uint32_t pos = __builtin_ctz(i);
                uint32_t other = i - (1u << pos);
                AssignedNonZeroPoint p = ecc_add_unsafe(cl[other], pts[pos]);
                p = ecc_reduce_non_zero(p);
                if (with_select) assign_cache_point_non_zero(p, group_prefix + group_index, i);
                cl.push_back(p);
                ctx.cut();
            }
        };
        // full groups: strands (the init point alternates r2 / -r2 with the group parity -> a parameter)
        std::vector<AssignedNonZeroPoint> cl0;
        auto add_candidates_pre = [&](uint32_t n_lanes, uint32_t sz, uint32_t hint_base, uint32_t params_begin, uint32_t n_params) {
            PreKernel pk;
            std::memset(&pk, 0, sizeof(pk));
            pk.early_after_segment = -1;
            pk.k.kind = H2E_PRE_MSM_CANDIDATES;
            pk.k.n_lanes = n_lanes;
            pk.k.hint_base = hint_base;
            pk.k.hints_per_lane = (1u << sz) - 1;
            pk.k.args_begin = (uint32_t)c.pre_args.size();
            c.pre_args.push_back(sz);
            pk.k.n_params = n_params;
            pk.k.params_begin = params_begin;
            pk.k.scratch_begin = c.n_jac_slots;
            c.n_jac_slots += n_lanes << sz;
            return pk;
        };
        int32_t full_seg_index = -1;
        if (n_full > 0) {
            uint32_t hbase = c.n_hint_slots;
            c.n_hint_slots += (uint32_t)n_full * ((1u << group_size) - 1);
            uint32_t seg_index = (uint32_t)c.segments.size();  // index the fork segment will get
            full_seg_index = (int32_t)seg_index;
            c.begin_hints(hbase);
            c.fork((uint32_t)n_full, 0, [&](uint32_t g) {
                std::vector<AssignedNonZeroPoint> pts;
                for (size_t j = 0; j < group_size; j++) {
                    AssignedNonZeroPoint pk = point_k((uint32_t)(g * group_size + j));
                    pts.push_back(AssignedNonZeroPoint{c.param(pk.x), c.param(pk.y)});
                }
                const AssignedNonZeroPoint& init_abs = (g % 2 == 0) ? rand_line_point : rand_line_point_neg;
                AssignedNonZeroPoint init{c.param(init_abs.x), c.param(init_abs.y)};
                std::vector<AssignedNonZeroPoint> cl;
                build_group(g, pts, cl, init, init, true);
                if (g == 0) cl0 = cl;
            });
            c.end_hints();
            if (n_chunks > n_full) c.segments[c.segments.size() - 2].expand_after_next = true;   // the remainder group's one-lane chain follows
            Segment seg_groups = c.segments[c.segments.size() - 2];
            {
                PreKernel pk = add_candidates_pre((uint32_t)n_full, (uint32_t)group_size, hbase, seg_groups.params_begin, seg_groups.n_params);
                pk.before_segment = seg_index;
                c.pre_kernels.push_back(pk);
            }
            for (size_t g = 0; g < n_full; g++) {
                const AssignedNonZeroPoint& init_abs = (g % 2 == 0) ? rand_line_point : rand_line_point_neg;
                write_table_entry(table_aux[g], init_abs);
                for (size_t i = 1; i < cl0.size(); i++)
                    write_table_entry(table_aux[g] + (uint32_t)i * NC,
                                      AssignedNonZeroPoint{c.strand_int(cl0[i].x, seg_groups, (uint32_t)g),
                                                           c.strand_int(cl0[i].y, seg_groups, (uint32_t)g)});
            }
        }
        // remainder group (fewer points): in the main context
        for (size_t g = n_full; g < n_chunks; g++) {
            std::vector<AssignedNonZeroPoint> pts;
            for (size_t j = g * group_size; j < n; j++) pts.push_back(point_k((uint32_t)j));
            std::vector<AssignedNonZeroPoint> cl;
            {   // one-lane predictor for the remainder group: its ref table lives in the parameter array
                uint32_t hbase = c.n_hint_slots;
                c.n_hint_slots += (1u << pts.size()) - 1;
                uint32_t pbegin = (uint32_t)c.params.size();
                std::vector<uint32_t> refs;
                for (auto& p : pts) push_point_refs(refs, p);
                push_point_refs(refs, (g % 2 == 0) ? rand_line_point : rand_line_point_neg);
                c.params.insert(c.params.end(), refs.begin(), refs.end());
                PreKernel pk = add_candidates_pre(1, (uint32_t)pts.size(), hbase, pbegin, (uint32_t)refs.size());
                pk.before_segment = (uint32_t)c.segments.size() - 1;  // the current main segment
                // it only reads the points and the random line point, like the full groups' predictor: side stream
                pk.early_after_segment = full_seg_index;
                c.pre_kernels.push_back(pk);
                c.begin_hints(hbase);
            }
            build_group(g, pts, cl, rand_line_point, rand_line_point_neg, false);
            c.end_hints();
            for (size_t i = 0; i < cl.size(); i++) write_table_entry(table_aux[g] + (uint32_t)i * NC, cl[i]);
        }

        // decompose_scalar per scalar (ecc_chip.rs:277-280)
        std::vector<AssignedCondition> bits0;
        c.fork(n, 0, [&](uint32_t k) {
            AssignedInteger sk = scalar_param(scalar_k(k));
            std::vector<AssignedCondition> b = scalar_decompose(sk);
            if (k == 0) bits0 = b;
        });
        Segment seg_bits = c.segments[c.segments.size() - 2];
        size_t windows = bits0.size();
        size_t n_groups = n_chunks;

        // windows (ecc_chip.rs:289-352): predict_ops + clones == strands 0..windows-1
        AssignedNonZeroPoint line_acc0;
        // full value hints: a block of 8 slots per ecc_add_unsafe + one block for the chain's initial point
        const uint32_t win_hints_per_lane = H2E_ECC_HINT_SLOTS * ((uint32_t)n_groups + 1);
        c.n_hint_slots = (c.n_hint_slots + H2E_ECC_HINT_SLOTS - 1) / H2E_ECC_HINT_SLOTS * H2E_ECC_HINT_SLOTS;   // blocks are 8-aligned
        uint32_t win_hbase = c.n_hint_slots;
        c.n_hint_slots += (uint32_t)windows * win_hints_per_lane;
        uint32_t win_seg_index = (uint32_t)c.segments.size();
        uint32_t win_jac = c.n_jac_slots;
        c.n_jac_slots += (uint32_t)windows;
        uint32_t win_sel = c.n_sel_slots;
        c.n_sel_slots += (uint32_t)(windows * n_groups);
        c.begin_hints(win_hbase, 2);
        c.fork((uint32_t)windows, 0, [&](uint32_t wi) {
            AssignedNonZeroPoint acc = rand_acc_point_neg;
            for (size_t group_index = 0; group_index < n_groups; group_index++) {
                size_t lo = group_index * group_size, hi = std::min<size_t>(n, lo + group_size);
                std::vector<AssignedCondition> group_bits;
                for (size_t j = lo; j < hi; j++)
                    group_bits.push_back(AssignedCondition{c.param(AssignedValue{c.strand_ref(bits0[wi].v.ref, seg_bits, (uint32_t)j)})});
                AssignedNonZeroPoint ci;
                if (with_select) {
                    ci = pick_and_select(table_aux[group_index], group_bits, group_index + group_prefix, (int64_t)(win_sel + group_index));
                } else {
                    // the candidates as handles (their cells are listed in the group's table); the value chain follows
                    // the bisection tree itself (BISEC_INT ops), the predictor uses the select pre-kernel's pick
                    std::vector<AssignedNonZeroPoint> cands;
                    for (uint32_t i = 0; i < (1u << (hi - lo)); i++) {
                        const uint32_t* t = &c.aux[table_aux[group_index] + i * NC];
                        AssignedNonZeroPoint q;
                        for (int j = 0; j < L; j++) {
                            q.x.limbs_le[j] = t[j];
                            q.y.limbs_le[j] = t[L + 1 + j];
                        }
                        q.x.native = t[L];
                        q.y.native = t[2 * L + 1];
                        cands.push_back(q);
                    }
                    ci = bisec_candidate_non_zero(cands, group_bits);
                }
                acc = ecc_add_unsafe(ci, acc);
                c.cut();
            }
            if (wi == 0) line_acc0 = acc;
        }, true);
        c.end_hints();
        c.segments[c.segments.size() - 2].sel_stride = (uint32_t)n_groups;
        Segment seg_windows = c.segments[c.segments.size() - 2];
        uint32_t win_args = (uint32_t)c.pre_args.size();
        {   // select pre-kernel: one lane per (window, group) picks the candidate -> selection buffer (same arguments)
            PreKernel pk;
            std::memset(&pk, 0, sizeof(pk));
            pk.early_after_segment = -1;
            pk.k.kind = H2E_PRE_MSM_SELECT;
            pk.k.n_lanes = (uint32_t)(windows * n_groups);
            pk.k.args_begin = win_args;
            pk.k.n_params = seg_windows.n_params;
            pk.k.params_begin = seg_windows.params_begin;
            pk.k.sel_begin = win_sel;
            pk.before_segment = win_seg_index;
            pk.early_after_segment = -1;
            c.pre_kernels.push_back(pk);
        }
        {
            PreKernel pk;
            std::memset(&pk, 0, sizeof(pk));
            pk.early_after_segment = -1;
            pk.k.kind = H2E_PRE_MSM_WINDOWS;
            pk.k.sel_begin = win_sel;
            pk.k.n_lanes = (uint32_t)windows;
            pk.k.hint_base = win_hbase;
            pk.k.hints_per_lane = win_hints_per_lane;
            pk.k.ecc_ops = (uint32_t)n_groups;
            pk.k.pattern_len = 1;
            pk.k.pattern = H2E_ECC_ADD_EXT_PREV;
            pk.k.args_begin = (uint32_t)c.pre_args.size();
            c.pre_args.push_back((uint32_t)n_groups);
            c.pre_args.push_back((uint32_t)group_size);
            c.pre_args.push_back(n);
            push_point_refs(c.pre_args, rand_acc_point_neg);
            for (size_t g = 0; g < n_groups; g++) c.pre_args.push_back(table_aux[g]);
            pk.k.n_params = seg_windows.n_params;
            pk.k.params_begin = seg_windows.params_begin;
            pk.k.scratch_begin = win_jac;
            pk.k.scan_begin = c.n_jac_slots;
            c.n_jac_slots += H2E_WIN_SCAN_SLOTS((uint32_t)windows);
            pk.before_segment = win_seg_index;
            c.pre_kernels.push_back(pk);
        }
        {   // tail predictor: runs before the main segment that holds the accumulation loop
            PreKernel pk;
            std::memset(&pk, 0, sizeof(pk));
            pk.early_after_segment = -1;
            pk.k.kind = H2E_PRE_MSM_TAIL;
            pk.k.n_lanes = 1;
            pk.k.hint_base = c.n_hint_slots;
            pk.k.ecc_ops = (uint32_t)(windows * (2 + (n_groups % 2)));
            pk.k.hints_per_lane = H2E_ECC_HINT_SLOTS * (pk.k.ecc_ops + 1);
            pk.k.pattern_len = 2 + (uint32_t)(n_groups % 2);
            pk.k.pattern = H2E_ECC_DBL | (H2E_ECC_ADD_EXT_PREV << 2) | (H2E_ECC_ADD_PREV_EXT << 4);
            pk.k.args_begin = (uint32_t)c.pre_args.size();
            c.pre_args.push_back((uint32_t)windows);
            c.pre_args.push_back((uint32_t)(n_groups % 2));
            push_point_refs(c.pre_args, rand_acc_point);
            push_point_refs(c.pre_args, rand_line_point_neg);
            c.pre_args.push_back(win_jac);
            pk.k.scan_begin = c.n_jac_slots;
            c.n_jac_slots += H2E_TAIL_SCAN_SLOTS((uint32_t)windows);
            pk.before_segment = (uint32_t)c.segments.size() - 1;
            pk.early_after_segment = (int32_t)win_seg_index;
            c.pre_kernels.push_back(pk);
            c.begin_hints(c.n_hint_slots, 2);
            c.n_hint_slots += pk.k.hints_per_lane;
        }

        // accumulate windows (ecc_chip.rs:354-362)
        AssignedNonZeroPoint acc = rand_acc_point;
        for (size_t wi = 0; wi < windows; wi++) {
            acc = ecc_double_unsafe(acc);
            c.cut();
            AssignedNonZeroPoint line{c.strand_int(line_acc0.x, seg_windows, (uint32_t)wi), c.strand_int(line_acc0.y, seg_windows, (uint32_t)wi)};
            acc = ecc_add_unsafe(line, acc);
            c.cut();
            if (n_groups % 2 == 1) {
                acc = ecc_add_unsafe(acc, rand_line_point_neg);
                c.cut();
            }
        }
        c.end_hints();
        // the rest of the main context (final curvature + carry add, then the caller's ecc_assert_equal) holds two
        // unhinted divisions: a segment of its own (the loop above then needs no replay: every value that escapes one of its
        // sub-ranges is a combination of hints), its expansion split finely, or one lane carries both inversions (3 ms of a 4 ms kernel)
        c.cut();
        c.split_segment();
        c.auto_cut_every = 4;
        AssignedPoint accp = ecc_non_zero_point_downgrade(acc);
        AssignedPointWithCurvature accc = to_point_with_curvature(accp);
        AssignedPoint carry = ecc_non_zero_point_downgrade(rand_acc_point_neg);
        return ecc_add(accc, carry);
    }
};

}  // namespace h2e
