"""Operator API (include/h2e.h): chip ops called one after the other on a device-resident Context - operands are
handles to rows written earlier, every op starts at the Context's current offsets (and msm prefix) - against the oracle running
the same op sequence in one context of its own.  All calls go through the C ABI."""
import numpy as np
import pytest

import oracle_lib
from halo2ecc_s_amd import Records, synth
from halo2ecc_s_amd import engine as E
from parity import expand_fixed

pytestmark = pytest.mark.gpu


def _rows_of(engine, rec, orun, n_inst):
    """compare every advice cell (exported to the reference's row-major layout, masked by the accumulated flags), the flags,
    offsets / heights, the permutation list in order and the fixed cells of a finished records object with an oracle run"""
    t = engine.torch
    t.cuda.synchronize()
    arrs = rec.arrays()
    sh = rec.shape()
    i = orun.info
    assert i.status == 0, orun.error
    assert (sh.base_offset, sh.range_offset, sh.select_offset) == (i.base_offset, i.range_offset, i.select_offset)
    assert (sh.base_height, sh.range_height, sh.select_height) == (i.base_height, i.range_height, i.select_height)
    assert sh.n_permutations == i.n_permutations
    perms = E._view(sh.permutations, sh.n_permutations * 2, np.uint32).reshape(-1, 2)
    assert np.array_equal(perms, orun.permutations()), "permutation list differs"
    d = E._view(sh.dict, sh.n_dict * 4, np.uint64).reshape(-1, 4)
    fixp = (sh.base_fix, sh.range_fix, sh.select_fix)
    flp = (sh.base_flags, sh.range_flags, sh.select_flags)
    for region in range(3):
        rows, cols, fcols = rec.rows[region], E.COLS[region], (9, 2, 2)[region]
        flags = E._view(flp[region], rows * cols, np.uint8).reshape(rows, cols)
        ovals, oflags = orun.adv(region, rows)
        assert np.array_equal(flags, oflags), f"flags differ in region {region}"
        ids = E._view(fixp[region], rows * fcols, np.uint32).reshape(rows, fcols)
        ofix, opresent = orun.fix(region, rows)
        assert np.array_equal((ids != 0).astype(np.uint8), opresent), f"fixed presence differs in region {region}"
        assert np.array_equal(d[ids], ofix), f"fixed values differ in region {region}"
        got = arrs[region].permute(3, 0, 1, 2, 4).reshape(n_inst, rows, cols, 4)
        assigned = t.from_numpy((flags & 1).astype(bool)).to(got.device)
        return_vals = (got * assigned[None, :, :, None]).cpu().numpy().view(np.uint64)
        yield region, return_vals, ovals


def test_ops_msm_twice_in_one_context(engine, oracle):
    """assign_point x n, assign x n, int_mul / int_add / int_sub / reduce / int_div on assigned integers, msm_unsafe, a second
    msm_unsafe on the same handles (offsets deep inside the arrays, msm prefix 2^20: its select rows encode group + 2^20,
    src/circuit/native_scalar_ecc_chip.rs:173-178), ecc_assert_equal of the two results - nine separate ops on the GPU equal
    the oracle calling the same ops on one context, cell for cell"""
    n, n_inst = 6, 3
    ins = [synth.msm_bn256_tile_inputs(n, tile=800 + k)[0] for k in range(n_inst)]
    oruns = [oracle_lib.run_ops_msm_twice(n, inp) for inp in ins]
    i = oruns[0].info
    rows = (max(i.base_height, i.base_offset) + 1, max(i.range_height, i.range_offset) + 1, max(i.select_height, i.select_offset) + 1)
    rec = Records(engine, E.FIELD_BN256_FQ, n_inst, rows)
    a = np.stack(ins)                                            # [inst][4n+9][4]
    pts = rec.assign_points(n, a[:, 0:3 * n])
    scs = rec.assign_scalars(n, a[:, 3 * n:4 * n])
    m = rec.int_op(E.INT_MUL, pts[0].x, pts[0].y)
    s = rec.int_op(E.INT_ADD, m, m)
    s2 = rec.int_op(E.INT_SUB, s, pts[0].x)
    assert (m.times, s.times, s2.times) == (1, 2, 4)
    rd = rec.int_op(E.INT_REDUCE, s2)
    q, cond = rec.int_op(E.INT_DIV, rd, pts[0].y)
    g, r1, r2 = a[:, 4 * n:4 * n + 2], a[:, 4 * n + 2:4 * n + 4], a[:, 4 * n + 4:4 * n + 6]
    res1 = rec.msm_unsafe(pts, scs, np.concatenate([g, r1, r2], axis=1))
    res2 = rec.msm_unsafe(pts, scs, np.concatenate([g, r2, r1], axis=1))
    rec.ecc_assert_equal(res1, res2)
    engine.torch.cuda.synchronize()
    assert (rec.arrays()[3].cpu().numpy() == 0).all()
    for k, orun in enumerate(oruns):
        for region, got, ovals in _rows_of(engine, rec, orun, n_inst):
            assert np.array_equal(got[k], ovals), f"instance {k}: advice differs in region {region}"
    rec.close()


def test_ops_msm_in_a_context_without_the_select_chip(engine, oracle):
    """NativeScalarEccContext::new_without_select_chip (src/context.rs:201-205; the reference's
    test_native_ecc_chip_without_select_chip, src/tests/native_scalar_ecc_chip.rs:63-110): a device-resident context created
    without the select chip dispatches msm_unsafe to the bisection form (ecc_chip.rs:373-408 -> :91-221: groups of two points,
    no cache / select rows).  assign_point x n, assign x n, msm_unsafe, assign_point(expected), ecc_assert_equal as five ops ==
    the oracle's no-select test body, cell for cell, flags, fixed cells and permutations included."""
    n, n_inst = 5, 2
    ins = [synth.msm_bn256_tile_inputs(n, tile=830 + k)[0] for k in range(n_inst)]
    oruns = [oracle_lib.run_msm_bn256_tile(n, inp, with_select=False) for inp in ins]
    i = oruns[0].info
    assert i.status == 0, oruns[0].error
    assert i.select_height == 0 and i.select_offset == 0          # no select rows at all
    rows = (max(i.base_height, i.base_offset) + 1, max(i.range_height, i.range_offset) + 1, 8)
    rec = Records(engine, E.FIELD_BN256_FQ, n_inst, rows, select_chip=False)
    a = np.stack(ins)
    pts = rec.assign_points(n, a[:, 0:3 * n])
    scs = rec.assign_scalars(n, a[:, 3 * n:4 * n])
    res = rec.msm_unsafe(pts, scs, a[:, 4 * n:4 * n + 6])
    exp = rec.assign_points(1, a[:, 4 * n + 6:4 * n + 9])
    rec.ecc_assert_equal(res, exp[0])
    engine.torch.cuda.synchronize()
    assert (rec.arrays()[3].cpu().numpy() == 0).all()
    for k, orun in enumerate(oruns):
        for region, got, ovals in _rows_of(engine, rec, orun, n_inst):
            assert np.array_equal(got[k], ovals), f"instance {k}: advice differs in region {region}"
    rec.close()


def test_ops_check_pairing_on_assigned_terms(engine, oracle):
    """PairingChipOps::check_pairing (src/circuit/pairing_chip.rs:173-176) as an op on terms assigned by earlier ops: the G2
    constants, two assign_point ops, then the check - the same rows as the reference's test body
    (src/tests/native_scalar_pairing_chip.rs:67-97)"""
    n_inst = 2
    ins = [synth.pairing_check_bn256_inputs(instance=810 + k) for k in range(n_inst)]
    oruns = [oracle_lib.run_pairing_check_bn256(inp) for inp in ins]
    i = oruns[0].info
    rows = (max(i.base_height, i.base_offset) + 1, max(i.range_height, i.range_offset) + 1, 1)
    rec = Records(engine, E.FIELD_BN256_FQ, n_inst, rows, emit_shape=False)
    a = np.stack(ins)
    b = rec.assign_g2_constant(a[:, 0:4])
    neg_a = rec.assign_points(1, a[:, 4:7])[0]
    pa = rec.assign_points(1, a[:, 7:10])[0]
    rec.check_pairing([pa, neg_a], [b, b])
    engine.torch.cuda.synchronize()
    arrs = rec.arrays()
    assert (arrs[3].cpu().numpy() == 0).all()
    sh = rec.shape()
    assert (sh.base_offset, sh.range_offset) == (i.base_offset, i.range_offset)
    for k, orun in enumerate(oruns):
        for region in range(2):
            ovals, _ = orun.adv(region, rows[region])
            got = arrs[region][:, :, :, k, :].cpu().numpy().view(np.uint64).reshape(rows[region], E.COLS[region], 4)
            assert np.array_equal(got, ovals), f"instance {k}: advice differs in region {region}"
    rec.close()


def test_ops_complete_addition_surface(engine, oracle):
    """the public EccChipBaseOps surface no BASELINE config reaches, as ops on assigned handles -
    ecc_reduce_with_curvature (ecc_reduce, assign_identity, bisec_point), ecc_double, to_point_with_curvature, the complete
    ecc_add, ecc_neg, ecc_encode, ecc_mul, assign_constant_point, bisec_point_with_curvature, assign_cache_point /
    assign_selected_point on points with curvature (picked on the device by the index cell's value), ecc_assert_equal -
    sixteen ops equal the oracle calling the same methods (src/circuit/ecc_chip.rs:441-812) on one context"""
    n_inst = 3
    ins = [synth.ops_ecc_surface_inputs(instance=k) for k in range(n_inst)]
    oruns = [oracle_lib.run_ops_ecc_surface(inp) for inp in ins]
    i = oruns[0].info
    rows = (max(i.base_height, i.base_offset) + 1, max(i.range_height, i.range_offset) + 1, max(i.select_height, i.select_offset) + 1)
    rec = Records(engine, E.FIELD_BN256_FQ, n_inst, rows)
    a = np.stack(ins)
    P = rec.assign_points(1, a[:, 0:3])[0]
    Q = rec.assign_points(1, a[:, 3:6])[0]
    s = rec.assign_scalars(1, a[:, 6:7])[0]
    idx = rec.assign(a[:, 7:8])
    Pc = rec.ecc_reduce_with_curvature(P)
    D = rec.ecc_double(Pc)
    Qc = rec.to_point_with_curvature(Q)
    S = rec.ecc_add(Qc, D)
    N = rec.ecc_neg(S)
    enc = rec.ecc_encode(N)
    assert len(enc) == 3
    rec.ecc_mul(P, s, a[:, 8:14])
    g = synth.bn_g1_gen()
    Cp = rec.assign_constant_point(g[0].a, g[1].a)
    Cc = rec.to_point_with_curvature(Cp)
    rec.bisec_point_with_curvature(P.z, Pc, Cc)
    rec.assign_cache_point(Pc, 7, 0)
    rec.assign_cache_point(Cc, 7, 1)
    Sel = rec.assign_selected_point([Pc, Cc], idx, 7)
    rec.ecc_assert_equal(Sel.p, Cp)
    engine.torch.cuda.synchronize()
    assert (rec.arrays()[3].cpu().numpy() == 0).all()
    for k, orun in enumerate(oruns):
        for region, got, ovals in _rows_of(engine, rec, orun, n_inst):
            assert np.array_equal(got[k], ovals), f"instance {k}: advice differs in region {region}"
    rec.close()


def test_ops_attach_splices_into_caller_arrays(engine, oracle):
    """The splice seam (ParallelClone, src/circuit/ecc_chip.rs:64-77, :289-352; native_scalar_ecc_chip.rs:50-90,
    :173-178): the first half of the nine-op MSM scenario runs on a records object that owns its arrays; a second records
    object is then *attached* to those arrays at the first one's cursors and msm prefix (h2e_records_attach: caller-allocated
    arrays, starting offsets deep inside them, prefix 2^20) and runs the second msm_unsafe + ecc_assert_equal.  The arrays
    then equal the oracle's single context cell for cell, and the attached object reports the offsets / heights the caller's
    apply_offset_diff / merge need."""
    n, n_inst = 6, 3
    ins = [synth.msm_bn256_tile_inputs(n, tile=820 + k)[0] for k in range(n_inst)]
    oruns = [oracle_lib.run_ops_msm_twice(n, inp) for inp in ins]
    i = oruns[0].info
    rows = (max(i.base_height, i.base_offset) + 1, max(i.range_height, i.range_offset) + 1, max(i.select_height, i.select_offset) + 1)
    rec = Records(engine, E.FIELD_BN256_FQ, n_inst, rows, emit_shape=False)
    a = np.stack(ins)
    pts = rec.assign_points(n, a[:, 0:3 * n])
    scs = rec.assign_scalars(n, a[:, 3 * n:4 * n])
    m = rec.int_op(E.INT_MUL, pts[0].x, pts[0].y)
    s = rec.int_op(E.INT_ADD, m, m)
    s2 = rec.int_op(E.INT_SUB, s, pts[0].x)
    rd = rec.int_op(E.INT_REDUCE, s2)
    rec.int_op(E.INT_DIV, rd, pts[0].y)
    g, r1, r2 = a[:, 4 * n:4 * n + 2], a[:, 4 * n + 2:4 * n + 4], a[:, 4 * n + 4:4 * n + 6]
    res1 = rec.msm_unsafe(pts, scs, np.concatenate([g, r1, r2], axis=1))
    sh = rec.shape()
    off = (sh.base_offset, sh.range_offset, sh.select_offset)
    assert all(o > 0 for o in off)
    arrs = rec.arrays()
    fork = Records.attach(engine, E.FIELD_BN256_FQ, arrs, off, msm_prefix=1 << 20, emit_shape=False)   # MSM_PREFIX_OFFSET: one msm so far
    res2 = fork.msm_unsafe(pts, scs, np.concatenate([g, r2, r1], axis=1))
    fork.ecc_assert_equal(res1, res2)
    engine.torch.cuda.synchronize()
    assert (arrs[3].cpu().numpy() == 0).all()
    fs = fork.shape()
    assert (fs.base_offset, fs.range_offset, fs.select_offset) == (i.base_offset, i.range_offset, i.select_offset)
    for k, orun in enumerate(oruns):
        for region in range(3):
            ovals, oflags = orun.adv(region, rows[region])
            got = arrs[region][:, :, :, k, :].cpu().numpy().view(np.uint64).reshape(rows[region], E.COLS[region], 4)
            mask = (oflags & 1).astype(bool)
            assert np.array_equal(got[mask], ovals[mask]), f"instance {k}: advice differs in region {region}"
    fork.close()
    rec.close()


@pytest.mark.parametrize("curve", [0, 1])
def test_ops_integer_and_tower_surface(engine, oracle, curve):
    """The rest of IntegerChipOps (int_neg, int_square, int_unsafe_invert, is_int_zero, is_int_equal, assign_int_constant,
    int_mul_small_constant, bisec_int, assert_int_equal: src/circuit/integer_chip.rs:15-70) and the Fq2 / Fq6 / Fq12 surface
    (src/circuit/fq12.rs:24-459: add, sub, mul, square, neg, double, conjugate, unsafe_invert, mul_by_nonresidue, frobenius_map,
    cyclotomic_square, reduce, assert_equal) as ops on assigned elements, bn256 and bls12_381 towers - ~70 ops equal the oracle
    calling the same methods on one context.  A second records object then repeats the sequence: every op comes from the
    context's op-program cache (no host-side recording) and the arrays are the same."""
    n_inst = 2
    fp = E.FIELD_BN256_FQ if curve == 0 else E.FIELD_BLS12_381_FQ
    w = synth.W_MODULUS[curve]
    sw = synth.SLOT_WORDS[curve]
    ins = []
    for k in range(n_inst):
        rng = synth.SplitMix64(synth.SEED0 + 77 + curve + 1000003 * k)
        ins.append(synth.pack([rng.below(w) for _ in range(26)], sw))
    oruns = [oracle_lib.run_ops_int_tower(curve, inp) for inp in ins]
    i = oruns[0].info
    assert i.status == 0, oruns[0].error
    rows = (max(i.base_height, i.base_offset) + 1, max(i.range_height, i.range_offset) + 1, 1)
    a = np.stack(ins)

    def scenario(rec):
        A, B = rec.assign_w(a[:, 0:1]), rec.assign_w(a[:, 1:2])
        X = [rec.assign_w(a[:, 2 + j:3 + j]) for j in range(12)]
        Y = [rec.assign_w(a[:, 14 + j:15 + j]) for j in range(12)]
        rec.int_unary(E.INT_NEG, A)
        rec.int_unary(E.INT_SQUARE, A)
        inv = rec.int_unary(E.INT_UNSAFE_INVERT, B)
        z = rec.int_unary(E.INT_IS_ZERO, A)
        rec.int_unary(E.INT_IS_EQUAL, A, B)
        rec.assign_int_constant(w - 5)
        rec.int_mul_small_constant(A, 5)
        rec.bisec_int(z, A, B)
        t = rec.int_op(E.INT_MUL, inv, B)
        one = rec.assign_int_constant(1)
        rec.int_unary(E.INT_ASSERT_EQUAL, t, one)
        for deg in (2, 6, 12):
            x, y = X[:deg], Y[:deg]
            s = rec.fq(deg, E.FQ_ADD, x, y)
            rec.fq(deg, E.FQ_SUB, x, y)
            m = rec.fq(deg, E.FQ_MUL, x, y)
            rec.fq(deg, E.FQ_SQUARE, x)
            rec.fq(deg, E.FQ_NEG, x)
            rec.fq(deg, E.FQ_DOUBLE, x)
            if deg == 2:
                rec.fq(deg, E.FQ_CONJUGATE, x)
                rec.fq(deg, E.FQ_UNSAFE_INVERT, y)
                rec.fq(deg, E.FQ_MUL_BY_NONRESIDUE, m)
                rec.fq(deg, E.FQ_FROBENIUS_MAP, x, imm=1)
            elif deg == 6:
                rec.fq(deg, E.FQ_UNSAFE_INVERT, y)
                rec.fq(deg, E.FQ_MUL_BY_NONRESIDUE, m)
                rec.fq(deg, E.FQ_FROBENIUS_MAP, x, imm=1)
            else:
                rec.fq(deg, E.FQ_CONJUGATE, x)
                rec.fq(deg, E.FQ_FROBENIUS_MAP, x, imm=1)
                rec.fq(deg, E.FQ_CYCLOTOMIC_SQUARE, x)
                rec.fq(deg, E.FQ_UNSAFE_INVERT, y)
            r = rec.fq(deg, E.FQ_REDUCE, s)
            rec.fq(deg, E.FQ_ASSERT_EQUAL, r, s)

    rec = Records(engine, fp, n_inst, rows)
    scenario(rec)
    engine.torch.cuda.synchronize()
    assert (rec.arrays()[3].cpu().numpy() == 0).all(), rec.arrays()[3].cpu().numpy()
    for k, orun in enumerate(oruns):
        for region, got, ovals in _rows_of(engine, rec, orun, n_inst):
            assert np.array_equal(got[k], ovals), f"instance {k}: advice differs in region {region}"
    first = [x.clone() for x in rec.arrays()[:2]]
    hits0, miss0 = engine.get_stat(E.STAT_OP_CACHE_HITS), engine.get_stat(E.STAT_OP_CACHE_MISSES)
    rec2 = Records(engine, fp, n_inst, rows)
    scenario(rec2)
    engine.torch.cuda.synchronize()
    assert engine.get_stat(E.STAT_OP_CACHE_MISSES) == miss0 and engine.get_stat(E.STAT_OP_CACHE_HITS) > hits0
    for region in range(2):
        assert engine.torch.equal(rec2.arrays()[region], first[region])
    sh1, sh2 = rec.shape(), rec2.shape()
    assert (sh1.base_offset, sh1.range_offset, sh1.n_permutations) == (sh2.base_offset, sh2.range_offset, sh2.n_permutations)
    rec2.close()
    # a bounded cache (H2E_OPT_OP_CACHE_CAP): with room for four programs the scenario's ops evict each other; the third pass re-records
    # what was let go and still writes the same arrays
    if curve == 0:
        ev0 = engine.get_stat(E.STAT_OP_CACHE_EVICTIONS)
        engine.set_option(E.OPT_OP_CACHE_CAP, 4)
        try:
            assert engine.get_stat(E.STAT_OP_CACHE_SIZE) <= 4 and engine.get_stat(E.STAT_OP_CACHE_EVICTIONS) > ev0
            miss1 = engine.get_stat(E.STAT_OP_CACHE_MISSES)
            rec3 = Records(engine, fp, n_inst, rows)
            scenario(rec3)
            engine.torch.cuda.synchronize()
            assert engine.get_stat(E.STAT_OP_CACHE_MISSES) > miss1 and engine.get_stat(E.STAT_OP_CACHE_SIZE) <= 4
            for region in range(2):
                assert engine.torch.equal(rec3.arrays()[region], first[region])
            rec3.close()
        finally:
            engine.set_option(E.OPT_OP_CACHE_CAP, 4096)
    rec.close()


def test_ops_pairing_on_assigned_terms(engine, oracle):
    """PairingChipOps::pairing (src/circuit/pairing_chip.rs:157-171) as an op on assigned terms - the G2 constants, the G1
    points, then pairing() returning the Fq12 result's twelve integers as handles - equals the oracle running
    pairing(terms) in one context (src/tests/native_scalar_pairing_chip.rs:20-65 without the final constant comparison);
    one more op on the returned handles (fq12 square) checks that they are the result's cells"""
    n_inst = 2
    ins = [synth.pairing_inputs(0, 1, instance=830 + k) for k in range(n_inst)]
    oruns = [oracle_lib.run_pairing(0, 1, False, x) for x in ins]
    i = oruns[0].info
    assert i.status == 0, oruns[0].error
    rows = (max(i.base_height, i.base_offset) + 1 + 4096, max(i.range_height, i.range_offset) + 1 + 4096, 1)
    rec = Records(engine, E.FIELD_BN256_FQ, n_inst, rows, emit_shape=False)
    a = np.stack(ins)
    b = rec.assign_g2_constant(a[:, 0:4])
    g1 = rec.assign_points(1, a[:, 4:7])[0]
    res = rec.pairing([g1], [b])
    assert len(res) == 12
    sh = rec.shape()
    assert (sh.base_offset, sh.range_offset) == (i.base_offset, i.range_offset)
    rec.fq(12, E.FQ_SQUARE, res)            # the handles are usable operands
    engine.torch.cuda.synchronize()
    arrs = rec.arrays()
    assert (arrs[3].cpu().numpy() == 0).all()
    for k, orun in enumerate(oruns):
        for region in range(2):
            nrow = (i.base_offset, i.range_offset)[region]
            ovals, _ = orun.adv(region, rows[region])
            got = arrs[region][:, :, :, k, :].cpu().numpy().view(np.uint64).reshape(rows[region], E.COLS[region], 4)
            assert np.array_equal(got[:nrow], ovals[:nrow]), f"instance {k}: advice differs in region {region}"
    rec.close()
