"""Device-side constraint check (include/h2e.h h2e_check) = the reference's own acceptance criterion, `MockProver::verify() == Ok`
(src/tests/mod.rs:117-132: base gate src/circuit/base_chip.rs:50-69, range gates + lookups range_chip.rs:119-220, select lookup
select_chip.rs:71-88, copy constraints context.rs:523-541), evaluated on the ENGINE's arrays for every instance of a run.

Small cases: verdicts and per-class failure counts equal the oracle's checker (oracle/checker.hpp) on the same, identically corrupted
Records; the arrays into 0xFF-poisoned memory still pass (unassigned cells count as zero).  BASELINE batch sizes: all 64 x 1024-point
tiles, all 64 bn256 / 16 bls12_381 checks pass, and one flipped cell fails on that instance only."""
import numpy as np
import pytest

import oracle_lib
from halo2ecc_s_amd import Program, synth
from halo2ecc_s_amd import engine as E

pytestmark = pytest.mark.gpu


def _run(engine, prog, inputs_list, fill=0):
    d_in = engine.upload_inputs(prog, np.stack(inputs_list))
    base, rng, sel, status = engine.alloc(prog, len(inputs_list), fill=fill)
    engine.run(prog, d_in, base, rng, sel, status)
    engine.torch.cuda.synchronize()
    assert int(status.abs().max()) == 0, status
    return d_in, base, rng, sel


def _fail(engine, prog, d_in, base, rng, sel, classes=0):
    out = engine.check(prog, d_in, base, rng, sel, classes=classes)
    engine.torch.cuda.synchronize()
    return out.cpu().numpy()


def _assigned_cell(prog, region, want_permute=None, start=0):
    """(row, col) of an assigned advice cell of that region (optionally one that is / is not in a copy constraint)"""
    flags = (prog.base_flags(), prog.range_flags(), prog.select_flags())[region]
    ok = (flags & 1) != 0
    if want_permute is not None:
        ok &= ((flags & 2) != 0) == want_permute
    rc = np.argwhere(ok)
    rc = rc[rc[:, 0] >= start]
    return int(rc[0][0]), int(rc[0][1])


CASES = [
    ("integer_chip_st_bn256", lambda: Program.integer_chip_st(0), lambda k: synth.integer_chip_st_inputs(0, seed_index=40 + k),
     lambda inp: oracle_lib.run_integer_chip_st(0, inp)),
    ("integer_chip_st_bls_fq", lambda: Program.integer_chip_st(1), lambda k: synth.integer_chip_st_inputs(1, seed_index=40 + k),
     lambda inp: oracle_lib.run_integer_chip_st(1, inp)),
    ("msm_tile_6", lambda: Program.msm_bn256_tile(6), lambda k: synth.msm_bn256_tile_inputs(6, tile=50 + k)[0],
     lambda inp: oracle_lib.run_msm_bn256_tile(6, inp)),
    ("msm_tile_no_select_5", lambda: Program.msm_bn256_tile(5, with_select=False), lambda k: synth.msm_bn256_tile_inputs(5, tile=60 + k)[0],
     lambda inp: oracle_lib.run_msm_bn256_tile(5, inp, with_select=False)),
    ("pairing_check_bn256", Program.pairing_check_bn256, lambda k: synth.pairing_check_bn256_inputs(instance=870 + k),
     oracle_lib.run_pairing_check_bn256),
]


@pytest.mark.parametrize("name,make,gen,orun", CASES, ids=[c[0] for c in CASES])
def test_check_small_cases_match_the_oracles_checker(engine, oracle, name, make, gen, orun):
    prog = make()
    ins = [gen(k) for k in range(3)]
    d_in, base, rng, sel = _run(engine, prog, ins, fill=0xFF)   # poisoned arrays: only assigned cells count
    f = _fail(engine, prog, d_in, base, rng, sel)
    assert (f[:, :E.CHECK_CLASSES] == 0).all(), f
    assert (f[:, E.CHECK_CLASSES:] == -1).all(), f
    o = orun(ins[1])
    assert o.check()[0]
    # the same corruption on both sides: instance 1, one assigned cell per region (+ 1 on the value, as oracle_corrupt_adv does)
    arrs = (base, rng, sel)
    picks = []
    for region in range(3):
        if (prog.base_rows, prog.range_rows, prog.select_rows)[region] == 0 or not ((prog.base_flags(), prog.range_flags(), prog.select_flags())[region] & 1).any():
            continue
        for want in (True, False):
            try:
                row, col = _assigned_cell(prog, region, want_permute=want, start=3)
            except IndexError:
                continue
            picks.append((region, row, col))
    assert picks
    n_caught = 0
    for region, row, col in picks:
        o = orun(ins[1])
        o.corrupt(region, row, col)
        want = o.check_counts().astype(np.int64)   # (all zero for a cell no gate constrains: the second copy of an assign_bit, quirk Q2)
        n_caught += int(want.sum() > 0)
        cell = int(arrs[region][row, col, 0, 1, 0])   # (a Python int: indexing gives a view that would follow the write)
        arrs[region][row, col, 0, 1, 0] = cell + 1
        f = _fail(engine, prog, d_in, base, rng, sel)
        arrs[region][row, col, 0, 1, 0] = cell
        assert (f[0, :E.CHECK_CLASSES] == 0).all() and (f[2, :E.CHECK_CLASSES] == 0).all(), (region, row, col, f)
        assert np.array_equal(f[1, :E.CHECK_CLASSES], want), (name, region, row, col, f[1], want)
    assert n_caught > 0
    # a single class on request
    f = _fail(engine, prog, d_in, base, rng, sel, classes=1 << E.CHECK_BASE_GATE)
    assert (f[:, :E.CHECK_CLASSES] == 0).all()


def test_check_needs_the_shape_and_the_inputs(engine):
    prog = Program.pairing_check_bn256()
    d_in, base, rng, sel = _run(engine, prog, [synth.pairing_check_bn256_inputs(instance=880)])
    with pytest.raises(E.H2EError):
        engine.check(prog, None, base, rng, sel)            # fixed cells made from the G2 inputs
    bare = Program.integer_chip_st(0, emit_shape=False)
    ins = [synth.integer_chip_st_inputs(0, seed_index=1)]
    d2, b2, r2, s2 = _run(engine, bare, ins)
    with pytest.raises(E.H2EError):
        engine.check(bare, d2, b2, r2, s2)


def test_check_a_wrong_g2_constant_fails_the_base_gate(engine):
    """the constants made from instance inputs are part of the criterion: the same arrays against another instance's inputs fail"""
    prog = Program.pairing_check_bn256()
    ins = [synth.pairing_check_bn256_inputs(instance=890 + k) for k in range(2)]
    d_in, base, rng, sel = _run(engine, prog, ins)
    swapped = engine.upload_inputs(prog, np.stack(ins[::-1]))
    f = _fail(engine, prog, swapped, base, rng, sel)
    assert (f[:, E.CHECK_BASE_GATE] > 0).all(), f


def _flip_and_check(engine, prog, d_in, base, rng, sel, n, victims):
    f = _fail(engine, prog, d_in, base, rng, sel)
    assert (f[:, :E.CHECK_CLASSES] == 0).all(), f[(f[:, :E.CHECK_CLASSES] != 0).any(axis=1)]
    arrs = (base, rng, sel)
    for region, inst in victims:
        if arrs[region].shape[0] == 0:
            continue
        row, col = _assigned_cell(prog, region, start=arrs[region].shape[0] // 2)
        cell = int(arrs[region][row, col, 0, inst, 0])   # (a Python int: indexing gives a view that would follow the write)
        arrs[region][row, col, 0, inst, 0] = cell ^ 1
        f = _fail(engine, prog, d_in, base, rng, sel)
        arrs[region][row, col, 0, inst, 0] = cell
        bad = (f[:, :E.CHECK_CLASSES] != 0).any(axis=1)
        assert bad[inst] and bad.sum() == 1, (region, inst, np.nonzero(bad)[0])


def test_check_64_tiles_full_size(engine):
    """configs[1]: all 64 x 1024-point tiles satisfy the reference's constraint system; one flipped cell per region fails on that tile only"""
    n, tiles = 1024, 64
    ins = [synth.msm_bn256_tile_inputs(n, tile=300 + t, cheap_points=True, with_expected=False)[0] for t in range(tiles)]
    prog = Program.msm_bn256_tile(n)
    d_in = engine.upload_inputs(prog, np.stack(ins))
    base, rng, sel, status = engine.alloc(prog, tiles)
    engine.run(prog, d_in, base, rng, sel, status)
    engine.torch.cuda.synchronize()
    # (`expected` was not supplied: the in-circuit ecc_assert_equal of the test body would fail - feed the MSM result back, as
    # the reference test does with the native library's sum, src/tests/native_scalar_ecc_chip.rs:40-47)
    refs = prog.outputs()
    exp = np.zeros((tiles, 3, 4), dtype=np.uint64)
    for k in range(tiles):
        xs = [engine.read_cell(base, r, k) for r in refs[0:3]]
        ys = [engine.read_cell(base, r, k) for r in refs[4:7]]
        assert engine.read_cell(base, refs[8], k) == 0
        exp[k] = synth.pack([sum(v << (108 * i) for i, v in enumerate(xs)) % synth.BN_Q, sum(v << (108 * i) for i, v in enumerate(ys)) % synth.BN_Q, 0], 4)
    d_in[:, 4 * n + 6:4 * n + 9, :] = engine.torch.from_numpy(exp.view(np.int64)).to(d_in.device)
    status.zero_()
    engine.run(prog, d_in, base, rng, sel, status)
    engine.torch.cuda.synchronize()
    assert int(status.abs().max()) == 0
    _flip_and_check(engine, prog, d_in, base, rng, sel, tiles, [(0, 5), (1, 37), (2, 63)])


@pytest.mark.parametrize("curve,units", [("bn256", 64), ("bls12_381", 16)])
def test_check_pairing_batches_full_size(engine, curve, units):
    """configs[3] / configs[4]: every check of the batch satisfies the constraint system"""
    if curve == "bn256":
        prog, gen = Program.pairing_check_bn256(), synth.pairing_check_bn256_inputs
    else:
        prog, gen = Program.pairing_check_bls12_381(), synth.pairing_check_bls12_381_inputs
    ins = [gen(instance=700 + k) for k in range(units)]
    d_in, base, rng, sel = _run(engine, prog, ins)
    _flip_and_check(engine, prog, d_in, base, rng, sel, units, [(0, 3), (1, units - 1)])
