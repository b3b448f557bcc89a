"""GPU parity tests: every advice cell the HIP engine writes must equal the oracle's, bit for bit
(integer work: no tolerance).  All calls go through the C ABI (libh2e.so)."""
import numpy as np
import pytest

import oracle_lib
from halo2ecc_s_amd import Program, synth
from parity import compare_advice, compare_shape

pytestmark = pytest.mark.gpu


def _rows(engine, prog, arrs):
    """batch-interleaved advice arrays -> per-instance row-major [inst][rows][cols][4] (the reference's Records layout);
    programs recorded with their shape export unassigned cells as zero whatever the buffer held"""
    out = tuple(engine.export(prog, region, a) for region, a in enumerate(arrs))
    engine.torch.cuda.synchronize()
    return out


def _run(engine, prog, inputs_list, fill=0):
    inputs = np.stack(inputs_list)
    d_in = engine.upload_inputs(prog, inputs)
    base, rng, sel, status = engine.alloc(prog, len(inputs_list), fill=fill)
    engine.run(prog, d_in, base, rng, sel, status)
    engine.torch.cuda.synchronize()
    base, rng, sel = _rows(engine, prog, (base, rng, sel))
    return base, rng, sel, status.cpu().numpy()


@pytest.mark.parametrize("fp", [0, 1, 2])
def test_int_mul_batch(engine, oracle, fp):
    n = 70  # more than one wave of strands
    ins = [synth.int_mul_batch_inputs(fp, n, seed_index=10 + k) for k in range(3)]
    prog = Program.int_mul_batch(fp, n)
    base, rng, sel, status = _run(engine, prog, ins)
    assert (status == 0).all()
    for k, inp in enumerate(ins):
        orun = oracle_lib.run_int_mul_batch(fp, n, inp)
        compare_advice(prog, orun, base, rng, sel, instance=k)


@pytest.mark.parametrize("fp", [0, 1, 2])
def test_integer_chip_st(engine, oracle, fp):
    ins = [synth.integer_chip_st_inputs(fp, seed_index=20 + k) for k in range(4)]
    prog = Program.integer_chip_st(fp)
    base, rng, sel, status = _run(engine, prog, ins)
    assert (status == 0).all(), status
    for k, inp in enumerate(ins):
        orun = oracle_lib.run_integer_chip_st(fp, inp)
        assert orun.info.status == 0
        compare_advice(prog, orun, base, rng, sel, instance=k)


def test_int_mul_edge_values(engine, oracle):
    """operands 0, 1, w-1 and values just below 2^w_ceil_bits"""
    for fp in (0, 1, 2):
        w = synth.W_MODULUS[fp]
        top = (1 << (w.bit_length())) - 1
        vals = [0, 0, 1, w - 1, w - 1, w - 1, top, top, w, 1, 0, w - 1]
        n = len(vals) // 2
        inp = synth.pack(vals, synth.SLOT_WORDS[fp])
        prog = Program.int_mul_batch(fp, n)
        base, rng, sel, status = _run(engine, prog, [inp])
        assert (status == 0).all()
        orun = oracle_lib.run_int_mul_batch(fp, n, inp)
        assert orun.info.status == 0
        compare_advice(prog, orun, base, rng, sel)


@pytest.mark.parametrize("n", [1, 5, 6, 10, 12, 33])   # 5, 10: no remainder group; 6: even group count; 33: several full groups + remainder
def test_msm_tile(engine, oracle, n):
    ins = [synth.msm_bn256_tile_inputs(n, tile=t)[0] for t in range(2)]
    prog = Program.msm_bn256_tile(n)
    base, rng, sel, status = _run(engine, prog, ins)
    assert (status == 0).all(), status
    for k, inp in enumerate(ins):
        orun = oracle_lib.run_msm_bn256_tile(n, inp)
        assert orun.info.status == 0, orun.error
        compare_advice(prog, orun, base, rng, sel, instance=k)


def test_msm_tile_full_size(engine, oracle):
    """BASELINE configs[1]'s tile - 1024 points, 38.1 M advice cells - cell for cell against the oracle (which takes a
    few seconds over all host cores), on the last of several tiles so that the multi-wave paths are exercised.  That last tile
    is the one tests/golden/pyref/msm_bn256_tile_n1024.json was traced on by the independent Python restatement: the oracle's
    digests of its three advice arrays must equal the fixture's (three texts, one witness - at BASELINE's size)"""
    import json
    import os
    n, tiles = 1024, 3
    ins = [synth.msm_bn256_tile_inputs(n, tile=100 + t, cheap_points=True)[0] for t in (2, 1, 0)]
    prog = Program.msm_bn256_tile(n)
    base, rng, sel, status = _run(engine, prog, ins)
    assert (status == 0).all(), status
    orun = oracle_lib.run_msm_bn256_tile(n, ins[tiles - 1], threads=os.cpu_count())
    assert orun.info.status == 0, orun.error
    compare_advice(prog, orun, base, rng, sel, instance=tiles - 1)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pyref", "msm_bn256_tile_n1024.json")) as f:
        fx = json.load(f)["pyref"]
    i = orun.info
    assert fx["offsets"] == [i.base_offset, i.range_offset, i.select_offset] and fx["heights"] == [i.base_height, i.range_height, i.select_height]
    assert fx["n_permutations"] == i.n_permutations and fx["n_advice_cells"] == i.n_advice_cells
    for region in range(3):
        assert fx["adv_digest"][region] == [int(x) for x in orun.digest(region)], f"oracle != pyref fixture in region {region}"


def test_msm_alternating_inputs_reuse_buffers(engine, oracle):
    """Runs with different inputs into the same arrays and workspace: nothing may survive from the previous run (a
    missing cross-stream dependency once passed every same-input repetition)."""
    n, tiles = 96, 4
    prog = Program.msm_bn256_tile(n, emit_shape=False)
    sets = [[synth.msm_bn256_tile_inputs(n, tile=10 * k + t, cheap_points=True)[0] for t in range(tiles)] for k in range(2)]
    d_in = [engine.upload_inputs(prog, np.stack(s)) for s in sets]
    base, rng, sel, status = engine.alloc(prog, tiles)
    for rep in range(4):
        k = rep % 2
        status.zero_()
        engine.run(prog, d_in[k], base, rng, sel, status)
        engine.torch.cuda.synchronize()
        assert (status.cpu().numpy() == 0).all(), (rep, status.cpu().numpy())
    orun = oracle_lib.run_msm_bn256_tile(n, sets[1][tiles - 1])
    assert orun.info.status == 0, orun.error
    base, rng, sel = _rows(engine, prog, (base, rng, sel))
    compare_advice(prog, orun, base, rng, sel, instance=tiles - 1)


BN_R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def _to_ints(a):
    """[..., 4] uint64 words -> flat list of Python ints"""
    a = a.reshape(-1, 4)
    return [sum(int(a[i, k]) << (64 * k) for k in range(4)) for i in range(a.shape[0])]


@pytest.mark.parametrize("n_inst", [1, 3, 33, 70])
def test_export_layouts(engine, n_inst):
    """h2e_export (device half) on random cells: batch-interleaved [rows][cols][half][inst][2] ->
    per-instance rows / columns, bit exact against a torch permute; instance counts around the 32-instance tile,
    row counts around the 8-row tile (emit_shape = 0: no flags, cells copied as they are)"""
    t = engine.torch
    prog = Program.int_mul_batch(0, 5, emit_shape=False)
    for region, (rows, cols) in enumerate(((prog.base_rows, 5), (prog.range_rows, 3), (prog.select_rows, 2))):
        g = t.Generator(device="cuda").manual_seed(rows + cols + n_inst)
        x = t.randint(-2**62, 2**62, (rows, cols, 2, n_inst, 2), dtype=t.int64, device="cuda", generator=g)
        want_rows = x.permute(3, 0, 1, 2, 4).reshape(n_inst, rows, cols, 4)
        got = engine.export(prog, region, x, layout=0)
        got_c = engine.export(prog, region, x, layout=1)
        t.cuda.synchronize()
        assert t.equal(got, want_rows)
        assert t.equal(got_c, want_rows.permute(0, 2, 1, 3).contiguous())


def test_export_columns_and_montgomery_of_a_witness(engine, oracle):
    """the exported columns of an MSM tile equal the oracle's Records columns (unassigned cells zero although the run
    wrote into 0xFF-poisoned arrays), and the Montgomery-form export is x * 2^256 mod n of the canonical one
    (the in-memory form of halo2's Fr: src/utils.rs:10-17 costs one such conversion per cell on the host)"""
    n = 6
    inp, _ = synth.msm_bn256_tile_inputs(n, tile=5)
    prog = Program.msm_bn256_tile(n)
    d_in = engine.upload_inputs(prog, np.stack([inp, inp]))
    arrs = engine.alloc(prog, 2, fill=0xFF)
    engine.run(prog, d_in, *arrs)
    engine.torch.cuda.synchronize()
    assert (arrs[3].cpu().numpy() == 0).all()
    orun = oracle_lib.run_msm_bn256_tile(n, inp)
    for region in range(3):
        cols = engine.export(prog, region, arrs[region], layout=1)
        mont = engine.export(prog, region, arrs[region], layout=0, form=1)
        engine.torch.cuda.synchronize()
        ovals, _ = orun.adv(region, arrs[region].shape[0])
        for inst in range(2):
            got = cols[inst].cpu().numpy().view(np.uint64)
            assert np.array_equal(got, np.ascontiguousarray(ovals.transpose(1, 0, 2)))
        rows = min(ovals.shape[0], 300)   # Python-int check of the Montgomery form on the first rows
        want = [(v << 256) % BN_R for v in _to_ints(ovals[:rows])]
        assert _to_ints(mont[1].cpu().numpy().view(np.uint64)[:rows]) == want


def test_msm_value_chain_does_not_depend_on_expansion(engine, oracle):
    """The expansion of a cut segment runs on its own stream, concurrently with the value chain of the following
    segments, so the value chain may only read cells the value chain itself stored.  With the expansion of every
    cut segment but the last left out (test hook), the last segment - what follows the MSM's accumulation loop, whose
    sums come from the windows' sums - must still come out exactly right.  (Regression: the tail's references into the
    windows were not counted.)"""
    n = 24
    inp, _ = synth.msm_bn256_tile_inputs(n, tile=3)
    prog = Program.msm_bn256_tile(n)
    from halo2ecc_s_amd import engine as E
    engine.set_option(E.OPT_TEST_SKIP_EXPANSION, -1)
    try:
        base, rng, sel, status = _run(engine, prog, [inp])
    finally:
        engine.set_option(E.OPT_TEST_SKIP_EXPANSION, E.OPT_OFF)
    assert (status == E.ST_TEST_HOOK).all(), status   # the hook marks the run: its arrays are not a witness
    orun = oracle_lib.run_msm_bn256_tile(n, inp)
    assert orun.info.status == 0, orun.error
    # the last launch: what follows the accumulation loop (final curvature + carry add + the test body's ecc_assert_equal); its
    # value chain reads the loop's last sum, which the loop's value chain made from the windows' sums, which ...
    launches = prog.launches()
    assert launches[-1]["n_strands"] == 1 and launches[-2]["n_strands"] == 1 and launches[-3]["n_strands"] == 254
    last0 = launches[-1]["base0"]
    ovals, _ = orun.adv(0, prog.base_rows)
    got = base[0].cpu().numpy().view(np.uint64)
    assert np.array_equal(got[last0:], ovals[last0:]), "the last segment differs when the expansions before it are left out"
    assert not np.array_equal(got[:last0], ovals[:last0]), "the hook left nothing out"


@pytest.mark.parametrize("splits", [None, "2"], ids=["two_launches", "four_launches"])
@pytest.mark.parametrize("curve", ["bn256", "bls12_381"])
def test_pairing_value_chain_does_not_depend_on_expansion(engine, oracle, curve, splits):
    """A pairing check runs as several launches (Miller loop | final exponentiation; H2E_PAIRING_SPLITS=2: also after each
    exponentiation by x).  A later launch's field chain loads hint slots an earlier chain exported (H2E_F_FROM_HINTS) and its hint
    store reads the earlier launches' integers from their CELLS (H2E_SX_CELLS) - while the earlier launch's expansion runs on
    another stream.  So every such cell must have been stored by a value chain (the compiler's cross-segment escape analysis):
    with the expansion of every launch but the last left out (test hook), the last launch's rows must still equal the oracle's.
    A missed escape would otherwise be a race the parity tests pass by luck."""
    import os
    from halo2ecc_s_amd import engine as E
    make_inputs = synth.pairing_check_bn256_inputs if curve == "bn256" else synth.pairing_check_bls12_381_inputs
    ins = [make_inputs(instance=70 + k) for k in range(2)]
    old = os.environ.get("H2E_PAIRING_SPLITS")
    if splits is not None:
        os.environ["H2E_PAIRING_SPLITS"] = splits
    try:
        prog = Program.pairing_check_bn256(emit_shape=False) if curve == "bn256" else Program.pairing_check_bls12_381(emit_shape=False)
    finally:
        if splits is not None:
            if old is None:
                del os.environ["H2E_PAIRING_SPLITS"]
            else:
                os.environ["H2E_PAIRING_SPLITS"] = old
    launches = prog.launches()
    assert len(launches) == (2 if splits is None else 4), launches
    engine.set_option(E.OPT_TEST_SKIP_EXPANSION, -1)
    try:
        # poisoned arrays: a cell the last launch's chains read without anybody having stored it reads 0xFF.., not a lucky zero
        d_in = engine.upload_inputs(prog, np.stack(ins))
        arrs = engine.alloc(prog, len(ins), fill=0xFF)
        engine.run(prog, d_in, *arrs)
        engine.torch.cuda.synchronize()
    finally:
        engine.set_option(E.OPT_TEST_SKIP_EXPANSION, E.OPT_OFF)
    assert (arrs[3].cpu().numpy() == E.ST_TEST_HOOK).all(), arrs[3].cpu().numpy()
    base0, range0, _ = prog.launch_rows(len(launches) - 1)
    assert base0 == launches[-1]["base0"] and base0 > 0 and range0 > 0
    k = 1
    orun = (oracle_lib.run_pairing_check_bn256 if curve == "bn256" else oracle_lib.run_pairing_check_bls12_381)(ins[k])
    assert orun.info.status == 0, orun.error
    # batch-interleaved arrays [row][col][half][inst][2] -> this instance's cells [row][col][4]
    for region, first in ((0, base0), (1, range0)):
        ovals, oflags = orun.adv(region, (prog.base_rows, prog.range_rows)[region])
        got = arrs[region][:, :, :, k, :].contiguous().cpu().numpy().view(np.uint64)
        got = got.reshape(got.shape[0], got.shape[1], 4)
        assigned = (oflags[first:] & 1).astype(bool)
        assert np.array_equal(got[first:][assigned], ovals[first:][assigned]), f"{curve}: the last launch's rows differ in region {region} when the expansions before it are left out"
        before = (oflags[:first] & 1).astype(bool)
        assert not np.array_equal(got[:first][before], ovals[:first][before]), "the hook left nothing out"
    prog.close()


@pytest.mark.parametrize("pct", [10, 45, 90])
def test_msm_split_expansion(engine, oracle, pct):
    """A big expansion goes out as two launches over a prefix / the rest of its sub-ranges, with the inverse fix-up of
    the first part in between (h2e_capi.cpp `expand`; at BASELINE's size: the MSM windows).  Forced here at a small
    size: every cell, the is_zero inverses of both parts included, must still equal the oracle's."""
    n, tiles = 96, 2
    ins = [synth.msm_bn256_tile_inputs(n, tile=40 + t, cheap_points=True)[0] for t in range(tiles)]
    prog = Program.msm_bn256_tile(n)
    from halo2ecc_s_amd import engine as E
    engine.set_option(E.OPT_X_SPLIT_MIN_LANES, 0)
    engine.set_option(E.OPT_X_SPLIT_PCT, pct)
    try:
        base, rng, sel, status = _run(engine, prog, ins)
        assert engine.get_stat(E.STAT_LAST_SPLIT_SEGMENTS) >= 1, "the split path was not taken"
    finally:
        engine.set_option(E.OPT_X_SPLIT_MIN_LANES, 1 << 21)
        engine.set_option(E.OPT_X_SPLIT_PCT, 45)
    assert (status == 0).all(), status
    for k, inp in enumerate(ins):
        orun = oracle_lib.run_msm_bn256_tile(n, inp)
        assert orun.info.status == 0, orun.error
        compare_advice(prog, orun, base, rng, sel, instance=k)


@pytest.mark.parametrize("mask", [1, 2, 4, 7])
def test_msm_scan_predictor_fallback(engine, oracle, mask):
    """The MSM chains are predicted as scans (engine.hip "scan predictors"): chunk sums, a short serial pass, then every
    chunk's real operations from the chunk's start value.  A start value the scan could not make (a scan-only addition of
    equal / opposite points: Z = 0) is replaced by walking the real chain.  Forced here on every other chunk / window
    (H2E_OPT_TEST_SCAN_FALLBACK): same witness.  n = 96: 20 groups -> 7 window chunks; 254 windows -> 16 tail chunks."""
    n, tiles = 96, 2
    ins = [synth.msm_bn256_tile_inputs(n, tile=70 + t, cheap_points=True)[0] for t in range(tiles)]
    prog = Program.msm_bn256_tile(n)
    from halo2ecc_s_amd import engine as E
    before = engine.get_stat(E.STAT_SCAN_FALLBACKS)
    base0, rng0, sel0, status0 = _run(engine, prog, ins)
    assert engine.get_stat(E.STAT_SCAN_FALLBACKS) == before, "the scan fell back although nothing was degenerate"
    engine.set_option(E.OPT_TEST_SCAN_FALLBACK, mask)
    try:
        base, rng, sel, status = _run(engine, prog, ins)
    finally:
        engine.set_option(E.OPT_TEST_SCAN_FALLBACK, 0)
    assert engine.get_stat(E.STAT_SCAN_FALLBACKS) > before, "the fallback path was not taken"
    assert (status == 0).all() and (status0 == 0).all(), status
    for k, inp in enumerate(ins):
        orun = oracle_lib.run_msm_bn256_tile(n, inp)
        assert orun.info.status == 0, orun.error
        compare_advice(prog, orun, base, rng, sel, instance=k)


@pytest.mark.parametrize("n", [1, 5, 12])
def test_msm_tile_no_select(engine, oracle, n):
    """MSM without the select chip (ecc_chip.rs:91-221, candidates by bisection trees)"""
    ins = [synth.msm_bn256_tile_inputs(n, tile=60 + t)[0] for t in range(2)]
    prog = Program.msm_bn256_tile(n, with_select=False)
    base, rng, sel, status = _run(engine, prog, ins)
    assert (status == 0).all(), status
    for k, inp in enumerate(ins):
        orun = oracle_lib.run_msm_bn256_tile(n, inp, with_select=False)
        assert orun.info.status == 0, orun.error
        compare_advice(prog, orun, base, rng, sel, instance=k)


@pytest.mark.parametrize("n", [2, 7, 50])   # 50 = the reference's test size (src/tests/general_scalar_ecc_chip.rs:22): 10 groups of 5
def test_msm_bls12_381_tile_general_scalars(engine, oracle, n):
    """general-scalar MSM (GeneralScalarEccContext<bls12_381::G1Affine, bn256::Fr>,
    src/circuit/general_scalar_ecc_chip.rs:93-168): points over the 4-limb bls12_381 Fq, scalars as 3-limb integers of the
    second integer context decomposed limb by limb into 324 windows - two W fields in one program"""
    tiles = 2 if n < 50 else 1
    ins = [synth.msm_bls12_381_tile_inputs(n, tile=700 + t, cheap_points=n >= 50)[0] for t in range(tiles)]
    prog = Program.msm_bls12_381_tile(n)
    base, rng, sel, status = _run(engine, prog, ins)
    assert (status == 0).all(), status
    for k, inp in enumerate(ins):
        orun = oracle_lib.run_msm_bls12_381_tile(n, inp)
        assert orun.info.status == 0, orun.error
        compare_advice(prog, orun, base, rng, sel, instance=k)


def test_pairing_result_paths(engine, oracle):
    """pairing(terms) == expected (first block of the reference's pairing tests: src/tests/native_scalar_pairing_chip.rs:20-65
    with one bn256 pair, general_scalar_pairing_chip.rs:20-72 with the product of two bls12_381 pairs): run without the
    expected constant, read the Fq12 result from the program's output cells, feed it back as the constant the in-circuit
    fq12_assert_eq compares with - status 0 - and compare every cell with the oracle"""
    t = engine.torch
    for curve, n_pairs, Q in ((0, 1, synth.BN_Q), (1, 2, synth.BLS_Q)):
        L = 3 if curve == 0 else 4
        inp = synth.pairing_inputs(curve, n_pairs, instance=1)
        prog = Program.pairing(curve, n_pairs, False, emit_shape=False)
        d_in = engine.upload_inputs(prog, np.stack([inp]))
        arrs = engine.alloc(prog, 1)
        engine.run(prog, d_in, *arrs)
        t.cuda.synchronize()
        assert int(arrs[3][0]) == 0
        refs = prog.outputs()
        assert len(refs) == 12 * (L + 1)

        def cell(ref):
            region, col, row = ref >> 30, (ref >> 27) & 7, ref & 0x3FFFFFF
            w = arrs[region][row, col, :, 0, :].reshape(4).cpu().numpy().view(np.uint64)
            return sum(int(w[k]) << (64 * k) for k in range(4))
        exp = [sum(cell(r) << (108 * j) for j, r in enumerate(refs[i * (L + 1):i * (L + 1) + L])) % Q for i in range(12)]
        inp2 = synth.pairing_inputs(curve, n_pairs, instance=1, expected=exp)
        prog2 = Program.pairing(curve, n_pairs, True, emit_shape=False)
        base, rng, sel, status = _run(engine, prog2, [inp2])
        assert (status == 0).all(), status
        orun = oracle_lib.run_pairing(curve, n_pairs, True, inp2)
        assert orun.info.status == 0, orun.error
        compare_advice(prog2, orun, base, rng, sel)
        # a wrong expected value must fail the in-circuit assert
        bad = list(exp)
        bad[5] = (bad[5] + 1) % Q
        _, _, _, st_bad = _run(engine, prog2, [synth.pairing_inputs(curve, n_pairs, instance=1, expected=bad)])
        assert st_bad[0] & 1


def test_msm_tile_with_identity_inputs(engine, oracle):
    """identity points go through ecc_bisec_to_non_zero_point / ecc_bisec_scalar (quirk Q9)"""
    n = 6
    inp, _ = synth.msm_bn256_tile_inputs(n, seed_index=7, identity_at=(1, 4))
    prog = Program.msm_bn256_tile(n)
    base, rng, sel, status = _run(engine, prog, [inp])
    assert (status == 0).all(), status
    orun = oracle_lib.run_msm_bn256_tile(n, inp)
    assert orun.info.status == 0, orun.error
    compare_advice(prog, orun, base, rng, sel)


def test_msm_wrong_expected_sets_status(engine):
    n = 3
    inp, _ = synth.msm_bn256_tile_inputs(n, with_expected=False)
    prog = Program.msm_bn256_tile(n, emit_shape=False)
    base, rng, sel, status = _run(engine, prog, [inp])
    assert status[0] & 1  # ASSERT_FAILED, like the reference's assert_true panic


@pytest.mark.parametrize("n", [7, 33])
def test_msm_unsafe_error_is_a_retry_status(engine, oracle, n):
    """UnsafeError (src/circuit/ecc_chip.rs:23-34): ecc_add_unsafe returns Err(AddSameOrNegPoint) when the two x coordinates
    are equal (:850-857) and the reference's test draws new blinding points and retries
    (src/tests/native_scalar_ecc_chip.rs:52-57).  Here: the middle instance's line blinding point r2 equals its first input
    point, so the first candidate addition of group 0 (rand_line_point + P_0, :265-268) is P + P.
    The engine reports H2E_ST_RETRY_ADD_SAME_OR_NEG_POINT on that instance only (its rows are garbage, as after a failed
    reference run) and the neighbours stay bit-exact.  The oracle, which restates the reference as written, *panics* on that
    input: `try_assert_false` (base_chip.rs:497-500) calls `assert_constant` first, whose `assert_eq!(a.val, b)`
    (base_chip.rs:377) fires before the bool is formed - at v0.3.2 the Err branch of ecc_add_unsafe is unreachable.  The
    engine's status names the reference's intent (a retryable failure), the oracle documents what the crate does."""
    from halo2ecc_s_amd.engine import ST_RETRY_ADD_SAME_OR_NEG_POINT
    ins = [synth.msm_bn256_tile_inputs(n, tile=300 + t)[0] for t in range(3)]
    bad = ins[1].copy()
    bad[4 * n + 4] = bad[0]   # r2.x = P_0.x
    bad[4 * n + 5] = bad[1]   # r2.y = P_0.y
    ins[1] = bad
    prog = Program.msm_bn256_tile(n)
    base, rng, sel, status = _run(engine, prog, ins)
    assert status[0] == 0 and status[2] == 0, status
    assert status[1] & ST_RETRY_ADD_SAME_OR_NEG_POINT, status
    orun_bad = oracle_lib.run_msm_bn256_tile(n, bad)
    assert orun_bad.info.status == 1 and "assert_constant" in orun_bad.error, (orun_bad.info.status, orun_bad.error)
    for k in (0, 2):
        orun = oracle_lib.run_msm_bn256_tile(n, ins[k])
        assert orun.info.status == 0, orun.error
        compare_advice(prog, orun, base, rng, sel, instance=k)


def test_pairing_check_bn256(engine, oracle):
    """config 4 unit: check_pairing([(a,b),(-a,b)]) — 6.17M advice cells per instance, bit-exact"""
    ins = [synth.pairing_check_bn256_inputs(instance=k) for k in range(2)]
    prog = Program.pairing_check_bn256(emit_shape=False)
    base, rng, sel, status = _run(engine, prog, ins)
    assert (status == 0).all(), status
    orun = oracle_lib.run_pairing_check_bn256(ins[1])
    assert orun.info.status == 0, orun.error
    assert (prog.base_offset, prog.range_offset) == (orun.info.base_offset, orun.info.range_offset)
    compare_advice(prog, orun, base, rng, sel, instance=1)


@pytest.mark.parametrize("curve", ["bn256", "bls12_381"])
@pytest.mark.parametrize("knob", [("H2E_NO_FIELD_CHAIN", "1"), ("H2E_FIELD_NO_SINKS", "1"), ("H2E_FIELD_NO_INLINE", "1"),
                                  ("H2E_FIELD_NO_REBALANCE", "1"), ("H2E_FIELD_NO_LONG", "1"), ("H2E_FIELD_NO_PAIRS", "1"), ("H2E_FIELD_STEP", "54")],
                         ids=["level_parallel_replay", "sinks_in_chain", "no_inlining", "no_rebalancing", "no_long_combinations", "no_product_pairs", "one_pass_rounds"])
def test_pairing_value_chain_variants(engine, oracle, curve, knob):
    """The pairing checks' value chain has a default form (field-domain program, digit-parallel kernel, hint-only combinations
    computed after the chain, combinations inlined and depth-balanced with long records, two products per row for the eight-digit
    fields, two passes of rows per round) and fall-backs a program is compiled to when a knob says so or the field-domain compiler
    meets an op outside its vocabulary: the level-parallel replay of the integer-chip ops, the chain
    with its sinks inside, the chain without inlining, and the round-4 forms of what round 5 changed (no re-association, no long
    combinations, one product per row, one pass per round).  Every form must give the same cells.
    (The knobs are read when the program is recorded.)"""
    import os
    name, value = knob
    make_inputs = synth.pairing_check_bn256_inputs if curve == "bn256" else synth.pairing_check_bls12_381_inputs
    ins = [make_inputs(instance=40 + k) for k in range(3)]
    old = os.environ.get(name)
    os.environ[name] = value
    try:
        prog = Program.pairing_check_bn256(emit_shape=False) if curve == "bn256" else Program.pairing_check_bls12_381(emit_shape=False)
    finally:
        if old is None:
            del os.environ[name]
        else:
            os.environ[name] = old
    base, rng, sel, status = _run(engine, prog, ins)
    assert (status == 0).all(), status
    orun = (oracle_lib.run_pairing_check_bn256 if curve == "bn256" else oracle_lib.run_pairing_check_bls12_381)(ins[2])
    assert orun.info.status == 0, orun.error
    compare_advice(prog, orun, base, rng, sel, instance=2)


def test_pairing_check_bls12_381(engine, oracle):
    """config 5 unit: check_pairing([(ac,b),(-a,bc)]) over the 4-limb bls12_381 Fq — 7.95M cells, bit-exact"""
    ins = [synth.pairing_check_bls12_381_inputs(instance=k) for k in range(2)]
    prog = Program.pairing_check_bls12_381(emit_shape=False)
    base, rng, sel, status = _run(engine, prog, ins)
    assert (status == 0).all(), status
    orun = oracle_lib.run_pairing_check_bls12_381(ins[0])
    assert orun.info.status == 0, orun.error
    compare_advice(prog, orun, base, rng, sel, instance=0)


def test_poisoned_buffers_only_assigned_cells_are_written(engine, oracle):
    """The engine writes assigned cells only (include/h2e.h): after a run into 0xFF-filled arrays every assigned cell
    equals the oracle's and every other word is still 0xFF - rows after the last op, holes inside rows and the
    unused select array of an integer-only program included (trailing rows are owned by no op)."""
    cases = [(Program.integer_chip_st(0), [synth.integer_chip_st_inputs(0, seed_index=77 + k) for k in range(3)],
              lambda inp: oracle_lib.run_integer_chip_st(0, inp)),
             (Program.msm_bn256_tile(5), [synth.msm_bn256_tile_inputs(5, tile=70 + k)[0] for k in range(3)],
              lambda inp: oracle_lib.run_msm_bn256_tile(5, inp))]
    for prog, ins, orun_of in cases:
        d_in = engine.upload_inputs(prog, np.stack(ins))
        base, rng, sel, status = engine.alloc(prog, len(ins), fill=0xFF)
        engine.run(prog, d_in, base, rng, sel, status)
        engine.torch.cuda.synchronize()
        assert (status.cpu().numpy() == 0).all()
        flags = (prog.base_flags(), prog.range_flags(), prog.select_flags())
        for k, inp in enumerate(ins):
            orun = orun_of(inp)
            for region, arr in enumerate((base, rng, sel)):
                rows, cols = arr.shape[0], arr.shape[1]
                got = arr[:, :, :, k, :].cpu().numpy().view(np.uint64).reshape(rows, cols, 4)   # raw, not exported
                ovals, oflags = orun.adv(region, rows)
                assigned = (flags[region] & 1).astype(bool)
                assert np.array_equal(assigned, (oflags & 1).astype(bool))
                assert np.array_equal(got[assigned], ovals[assigned])
                assert (got[~assigned] == np.uint64(0xFFFFFFFFFFFFFFFF)).all(), f"region {region}: an unassigned cell was written"


def test_digest_matches_oracle(engine, oracle):
    """h2e_digest (the on-device consumer of a streaming job) of every array of every instance == the oracle's digest of
    its Records (same definition, include/h2e.h), also when the run went into poisoned arrays"""
    n = 12
    ins = [synth.msm_bn256_tile_inputs(n, tile=80 + k)[0] for k in range(5)]
    prog = Program.msm_bn256_tile(n)
    d_in = engine.upload_inputs(prog, np.stack(ins))
    arrs = engine.alloc(prog, len(ins), fill=0xFF)
    engine.run(prog, d_in, *arrs)
    dg = [engine.digest(prog, region, arrs[region]) for region in range(3)]
    engine.torch.cuda.synchronize()
    assert (arrs[3].cpu().numpy() == 0).all()
    for k, inp in enumerate(ins):
        orun = oracle_lib.run_msm_bn256_tile(n, inp)
        for region in range(3):
            assert np.array_equal(dg[region][k].cpu().numpy().view(np.uint64), orun.digest(region)), (k, region)


@pytest.mark.parametrize("workload", ["msm", "msm_no_select", "pairing_bn256", "integer_chip"])
def test_stream_digest_matches_oracle(engine, oracle, workload):
    """h2e_run_digest / h2e_submit_digest: the digest the expansion (and the inverse fix-up) accumulate while they store ==
    the oracle's stream digest of its Records (same definition, include/h2e.h) for every array of every instance - i.e. every
    assigned cell went into it exactly once, with its position - and the arrays themselves are what a plain run writes.
    MSM tiles (forks, candidate tables, the tail), the no-select variant, a pairing check (field chain + hint store) and the
    integer-chip test body (is_zero rows: fix-up cells)."""
    t = engine.torch
    if workload == "msm":
        n, prog = 33, Program.msm_bn256_tile(33)
        ins = [synth.msm_bn256_tile_inputs(n, tile=90 + k)[0] for k in range(5)]
        oracle_run = lambda inp: oracle_lib.run_msm_bn256_tile(n, inp)   # noqa: E731
    elif workload == "msm_no_select":
        n, prog = 12, Program.msm_bn256_tile(12, with_select=False)
        ins = [synth.msm_bn256_tile_inputs(n, tile=95 + k)[0] for k in range(3)]
        oracle_run = lambda inp: oracle_lib.run_msm_bn256_tile(n, inp, with_select=False)   # noqa: E731
    elif workload == "pairing_bn256":
        prog = Program.pairing_check_bn256(emit_shape=False)
        ins = [synth.pairing_check_bn256_inputs(instance=40 + k) for k in range(3)]
        oracle_run = oracle_lib.run_pairing_check_bn256
    else:
        prog = Program.integer_chip_st(1)
        ins = [synth.integer_chip_st_inputs(1, seed_index=60 + k) for k in range(4)]
        oracle_run = lambda inp: oracle_lib.run_integer_chip_st(1, inp)   # noqa: E731
    d_in = engine.upload_inputs(prog, np.stack(ins))
    arrs = engine.alloc(prog, len(ins))
    dg = engine.run_digest(prog, d_in, *arrs)
    t.cuda.synchronize()
    assert (arrs[3].cpu().numpy() == 0).all(), arrs[3].cpu().numpy()
    for k, inp in enumerate(ins):
        orun = oracle_run(inp)
        assert orun.info.status == 0, orun.error
        for region in range(3):
            assert np.array_equal(dg[region, k].cpu().numpy().view(np.uint64), orun.stream_digest(region)), (workload, k, region)
        orun.close()
    # pipelined: the same digests through h2e_submit_digest into a second set of arrays
    arrs2 = engine.alloc(prog, len(ins))
    dg2 = t.full((3, len(ins), 4), -1, dtype=t.int64, device=dg.device)
    job = engine.submit_digest(prog, d_in, *arrs2, dg2)
    engine.wait(job)
    t.cuda.synchronize()
    assert t.equal(dg, dg2)
    for region in range(3):
        assert t.equal(arrs[region], arrs2[region])


def test_pipelined_submit_matches_run(engine, oracle):
    """h2e_submit / h2e_wait: runs queued back to back into a ring of two output-buffer sets (the value chain of run
    k + 1 overlaps the expansion of run k; each run owns its workspace and instance table) give exactly the arrays of
    one-after-the-other h2e_run calls.  Six runs over three different input batches, digests compared per run, the last
    run of each ring slot compared cell for cell with the oracle."""
    t = engine.torch
    n, tiles = 96, 3
    prog = Program.msm_bn256_tile(n)
    sets = [[synth.msm_bn256_tile_inputs(n, tile=200 + 10 * k + i, cheap_points=True)[0] for i in range(tiles)] for k in range(3)]
    d_in = [engine.upload_inputs(prog, np.stack(s_)) for s_ in sets]
    ring = [engine.alloc(prog, tiles) for _ in range(2)]
    want = []
    for k in range(3):   # reference digests from plain h2e_run
        ring[0][3].zero_()
        engine.run(prog, d_in[k], *ring[0])
        want.append(t.stack([engine.digest(prog, region, ring[0][region]) for region in range(3)]).clone())
    t.cuda.synchronize()
    got, jobs = [], []
    for step in range(6):
        slot = step % 2
        if len(jobs) == 2:   # the slot's previous run must be consumed before its arrays are overwritten
            j, s_, k_ = jobs.pop(0)
            engine.wait(j)
            got.append((k_, t.stack([engine.digest(prog, region, ring[s_][region]) for region in range(3)]).clone(), ring[s_][3].clone()))
        ring[slot][3].zero_()
        jobs.append((engine.submit(prog, d_in[step % 3], *ring[slot]), slot, step % 3))
    for j, s_, k_ in jobs:
        engine.wait(j)
        got.append((k_, t.stack([engine.digest(prog, region, ring[s_][region]) for region in range(3)]).clone(), ring[s_][3].clone()))
    t.cuda.synchronize()
    assert len(got) == 6
    for k_, dg, st in got:
        assert (st.cpu().numpy() == 0).all()
        assert t.equal(dg, want[k_]), f"pipelined run of batch {k_} differs from h2e_run"
    for slot, k_ in ((0, 4 % 3), (1, 5 % 3)):
        base, rng, sel = _rows(engine, prog, ring[slot][:3])
        orun = oracle_lib.run_msm_bn256_tile(n, sets[k_][tiles - 1])
        compare_advice(prog, orun, base, rng, sel, instance=tiles - 1)


def test_pipelined_submit_of_different_programs(engine, oracle):
    """Runs of *different* programs in flight at the same time on one context (an MSM tile batch over bn256 Fq, a bls12_381
    pairing check over the 4-limb field, a general-scalar MSM with two W fields): every job slot has its own workspace
    with its own slot width, instance table and streams - each run must come out as if it had run alone."""
    from halo2ecc_s_amd import engine as E
    t = engine.torch
    engine.set_option(E.OPT_PIPELINE_DEPTH, 3)
    try:
        n = 33
        progs = [Program.msm_bn256_tile(n), Program.pairing_check_bls12_381(), Program.msm_bls12_381_tile(7)]
        ins = [[synth.msm_bn256_tile_inputs(n, tile=300 + k)[0] for k in range(3)],
               [synth.pairing_check_bls12_381_inputs(instance=300 + k) for k in range(2)],
               [synth.msm_bls12_381_tile_inputs(7, tile=300 + k)[0] for k in range(3)]]
        d_in = [engine.upload_inputs(pg, np.stack(i_)) for pg, i_ in zip(progs, ins)]
        arrs = [engine.alloc(pg, len(i_), fill=0xFF) for pg, i_ in zip(progs, ins)]
        for rnd in range(2):   # second round: the slots' workspaces are reused by a different program than before
            order = (0, 1, 2) if rnd == 0 else (1, 2, 0)
            for a in arrs:
                a[3].zero_()
            jobs = [engine.submit(progs[k], d_in[k], *arrs[k]) for k in order]
            for j in jobs:
                engine.wait(j)
            t.cuda.synchronize()
            for k in range(3):
                assert (arrs[k][3].cpu().numpy() == 0).all(), (rnd, k, arrs[k][3].cpu().numpy())
        oracles = [lambda inp: oracle_lib.run_msm_bn256_tile(n, inp), oracle_lib.run_pairing_check_bls12_381,
                   lambda inp: oracle_lib.run_msm_bls12_381_tile(7, inp)]
        for k in range(3):
            base, rng, sel = _rows(engine, progs[k], arrs[k][:3])
            inst = len(ins[k]) - 1
            orun = oracles[k](ins[k][inst])
            assert orun.info.status == 0, orun.error
            compare_advice(progs[k], orun, base, rng, sel, instance=inst)
    finally:
        engine.set_option(E.OPT_PIPELINE_DEPTH, 2)


def _named_call(engine, fn, *args):
    from halo2ecc_s_amd.engine import _check, lib
    _check(getattr(lib(), fn)(engine._h, *args, engine.torch.cuda.current_stream().cuda_stream))


def test_named_entry_points(engine, oracle):
    """the named C entry points of SURVEY 8(b) (build-or-reuse the program for the shape, then run it) produce the same
    cells as the oracle: h2e_int_mul_batch, h2e_msm_bn256_tile, h2e_pairing_check_bn256 / _bls12_381"""
    t = engine.torch
    # int_mul batch
    fp, n = 0, 9
    prog = Program.int_mul_batch(fp, n)
    ins = [synth.int_mul_batch_inputs(fp, n, seed_index=300 + k) for k in range(2)]
    d_in = engine.upload_inputs(prog, np.stack(ins))
    arrs = engine.alloc(prog, 2)
    _named_call(engine, "h2e_int_mul_batch", fp, n, 2, d_in.data_ptr(), *(a.data_ptr() for a in arrs))
    t.cuda.synchronize()
    assert (arrs[3].cpu().numpy() == 0).all()
    compare_advice(prog, oracle_lib.run_int_mul_batch(fp, n, ins[1]), *_rows(engine, prog, arrs[:3]), instance=1)
    # MSM tile
    n = 7
    prog = Program.msm_bn256_tile(n)
    ins = [synth.msm_bn256_tile_inputs(n, tile=310 + k)[0] for k in range(2)]
    d_in = engine.upload_inputs(prog, np.stack(ins))
    arrs = engine.alloc(prog, 2)
    _named_call(engine, "h2e_msm_bn256_tile", n, 2, d_in.data_ptr(), *(a.data_ptr() for a in arrs))
    t.cuda.synchronize()
    assert (arrs[3].cpu().numpy() == 0).all()
    compare_advice(prog, oracle_lib.run_msm_bn256_tile(n, ins[0]), *_rows(engine, prog, arrs[:3]), instance=0)
    # pairing checks
    for name, mk, gen, orun_of in (("h2e_pairing_check_bn256", Program.pairing_check_bn256, synth.pairing_check_bn256_inputs, oracle_lib.run_pairing_check_bn256),
                                   ("h2e_pairing_check_bls12_381", Program.pairing_check_bls12_381, synth.pairing_check_bls12_381_inputs, oracle_lib.run_pairing_check_bls12_381)):
        prog = mk(emit_shape=False)
        ins = [gen(instance=320 + k) for k in range(2)]
        d_in = engine.upload_inputs(prog, np.stack(ins))
        arrs = engine.alloc(prog, 2)
        _named_call(engine, name, 2, d_in.data_ptr(), *(a.data_ptr() for a in arrs))
        t.cuda.synchronize()
        assert (arrs[3].cpu().numpy() == 0).all()
        compare_advice(prog, orun_of(ins[1]), *_rows(engine, prog, arrs[:3]), instance=1)


def test_pairing_bn256_batch_64(engine, oracle):
    """BASELINE configs[3] at its batch size: 64 x bn256 check_pairing with distinct inputs in one run (the grid of the
    level-parallel replay and the multi-wave expansion paths differ from the 2-instance unit test): every status word 0,
    first / a middle / last instance cell for cell against the oracle, digests of all 64 pairwise distinct"""
    n_inst = 64
    ins = [synth.pairing_check_bn256_inputs(instance=400 + k) for k in range(n_inst)]
    prog = Program.pairing_check_bn256(emit_shape=False)
    d_in = engine.upload_inputs(prog, np.stack(ins))
    arrs = engine.alloc(prog, n_inst)
    engine.run(prog, d_in, *arrs)
    dg = engine.digest(prog, 0, arrs[0])
    engine.torch.cuda.synchronize()
    assert (arrs[3].cpu().numpy() == 0).all(), arrs[3].cpu().numpy()
    assert len({tuple(r) for r in dg.cpu().numpy().tolist()}) == n_inst
    for k in (0, 37, n_inst - 1):
        orun = oracle_lib.run_pairing_check_bn256(ins[k])
        assert orun.info.status == 0, orun.error
        rows = [engine.export(prog, region, arrs[region])[k:k + 1] for region in range(3)]
        engine.torch.cuda.synchronize()
        compare_advice(prog, orun, *rows, instance=0)
        del rows


def test_pairing_bls12_381_batch_16(engine, oracle):
    """BASELINE configs[4] at its batch size: 16 x bls12_381 check_pairing, distinct inputs, one run"""
    n_inst = 16
    ins = [synth.pairing_check_bls12_381_inputs(instance=500 + k) for k in range(n_inst)]
    prog = Program.pairing_check_bls12_381(emit_shape=False)
    d_in = engine.upload_inputs(prog, np.stack(ins))
    arrs = engine.alloc(prog, n_inst)
    engine.run(prog, d_in, *arrs)
    engine.torch.cuda.synchronize()
    assert (arrs[3].cpu().numpy() == 0).all(), arrs[3].cpu().numpy()
    rows = _rows(engine, prog, arrs[:3])
    for k in (0, 9, n_inst - 1):
        orun = oracle_lib.run_pairing_check_bls12_381(ins[k])
        assert orun.info.status == 0, orun.error
        compare_advice(prog, orun, *rows, instance=k)


@pytest.mark.parametrize("curve,n_inst", [("bn256", 3), ("bn256", 8), ("bls12_381", 2), ("bls12_381", 5)])
def test_pairing_small_batches_packed_expansion(engine, oracle, curve, n_inst):
    """batches smaller than half a wave - one GPU's share of configs[3] / configs[4] at 8 GPUs (8 / 2 checks) and ragged shares (3, 5:
    the packed expansion's groups are padded to 4 / 8 lanes) - go through h2e_run_tape_packed and its order tables: every status 0,
    first and last instance cell for cell against the oracle, and the run submitted through the pipeline gives the same arrays"""
    if curve == "bn256":
        prog, gen, orun_of = Program.pairing_check_bn256(), synth.pairing_check_bn256_inputs, oracle_lib.run_pairing_check_bn256
    else:
        prog, gen, orun_of = Program.pairing_check_bls12_381(), synth.pairing_check_bls12_381_inputs, oracle_lib.run_pairing_check_bls12_381
    ins = [gen(instance=900 + k) for k in range(n_inst)]
    d_in = engine.upload_inputs(prog, np.stack(ins))
    arrs = engine.alloc(prog, n_inst, fill=0xFF)   # (the shape's flags mask the cells nobody assigns)
    engine.run(prog, d_in, *arrs)
    engine.torch.cuda.synchronize()
    assert (arrs[3].cpu().numpy() == 0).all(), arrs[3].cpu().numpy()
    rows = _rows(engine, prog, arrs[:3])
    for k in sorted({0, n_inst - 1}):
        orun = orun_of(ins[k])
        assert orun.info.status == 0, orun.error
        compare_advice(prog, orun, *rows, instance=k)
    arrs2 = engine.alloc(prog, n_inst, fill=0xFF)
    engine.wait(engine.submit(prog, d_in, *arrs2))
    engine.torch.cuda.synchronize()
    dg = [engine.digest(prog, region, arrs[region]).cpu().numpy() for region in range(2)]
    dg2 = [engine.digest(prog, region, arrs2[region]).cpu().numpy() for region in range(2)]
    assert all(np.array_equal(a, b) for a, b in zip(dg, dg2))


@pytest.mark.parametrize("curve,n_inst", [("bn256", 8), ("bls12_381", 2)])
def test_pairing_small_batches_sixteen_runs_in_flight(engine, oracle, curve, n_inst):
    """what a host does with one GPU's share of configs[3] / configs[4]: sixteen job slots (H2E_OPT_PIPELINE_DEPTH), 24 submissions of
    DIFFERENT batches back to back without a host wait - every run in its own chain stream, expansions and fix-ups behind the
    chains, `done` per slot, slots re-used by the last eight.  Every batch must come out as the same batch through h2e_run alone does
    (32-byte digests of the three arrays per instance), every status 0, and one run of the re-used slots cell for cell as the oracle's."""
    if curve == "bn256":
        prog, gen, orun_of = Program.pairing_check_bn256(), synth.pairing_check_bn256_inputs, oracle_lib.run_pairing_check_bn256
    else:
        prog, gen, orun_of = Program.pairing_check_bls12_381(), synth.pairing_check_bls12_381_inputs, oracle_lib.run_pairing_check_bls12_381
    t = engine.torch
    depth, n_runs, n_batches = 16, 24, 3
    batches = [[gen(instance=1200 + 50 * b + k) for k in range(n_inst)] for b in range(n_batches)]
    d_in = [engine.upload_inputs(prog, np.stack(ins)) for ins in batches]
    # the reference: each batch alone through h2e_run
    want = []
    ref = engine.alloc(prog, n_inst, fill=0xFF)
    for b in range(n_batches):
        ref[3].zero_()
        engine.run(prog, d_in[b], *ref)
        t.cuda.synchronize()
        assert (ref[3].cpu().numpy() == 0).all()
        want.append([engine.digest(prog, region, ref[region]).cpu().numpy() for region in range(3)])
    old_depth = engine.get_stat(3)   # H2E_STAT_PIPELINE_DEPTH
    engine.set_option(4, depth)      # H2E_OPT_PIPELINE_DEPTH
    try:
        bufs = [engine.alloc(prog, n_inst, fill=0xFF) for _ in range(depth)]
        got, pending = {}, []
        for k in range(n_runs):
            slot = k % depth
            if len(pending) >= depth:    # the slot's previous run: consume it (on the stream, not on the host) before its arrays are reused
                k0, job0 = pending.pop(0)
                engine.wait(job0)
                got[k0] = ([engine.digest(prog, region, bufs[k0 % depth][region]) for region in range(3)], bufs[k0 % depth][3].clone())
            bufs[slot][3].zero_()
            pending.append((k, engine.submit(prog, d_in[k % n_batches], *bufs[slot])))
        keep = None
        for k0, job0 in pending:
            engine.wait(job0)
            got[k0] = ([engine.digest(prog, region, bufs[k0 % depth][region]) for region in range(3)], bufs[k0 % depth][3].clone())
            if k0 == n_runs - 3:
                keep = (k0, _rows(engine, prog, bufs[k0 % depth][:3]))
        t.cuda.synchronize()
    finally:
        engine.set_option(4, old_depth)
    for k in range(n_runs):
        dg, status = got[k]
        assert (status.cpu().numpy() == 0).all(), (k, status.cpu().numpy())
        for region in range(3):
            assert np.array_equal(dg[region].cpu().numpy(), want[k % n_batches][region]), (k, region)
    k0, rows = keep
    orun = orun_of(batches[k0 % n_batches][n_inst - 1])
    assert orun.info.status == 0, orun.error
    compare_advice(prog, orun, *rows, instance=n_inst - 1)
    orun.close()


@pytest.mark.parametrize("curve", ["bn256", "bls12_381"])
def test_pairing_soak_statuses(engine, curve):
    """Every hint the value chain produces is checked by the expansion that consumes it (a differing hint sets H2E_ST_ARITH) and
    the circuit itself asserts e(a, b) e(-a, b) = 1: four more batches of fresh inputs through the digit-parallel chain - 256
    bn256 / 128 bls12-381 checks, ~2 x 10^7 residue operations per batch - must come back with every status word 0."""
    n_inst = 64 if curve == "bn256" else 32
    make = synth.pairing_check_bn256_inputs if curve == "bn256" else synth.pairing_check_bls12_381_inputs
    prog = Program.pairing_check_bn256(emit_shape=False) if curve == "bn256" else Program.pairing_check_bls12_381(emit_shape=False)
    arrs = engine.alloc(prog, n_inst)
    for batch in range(4):
        ins = [make(instance=7000 + 100 * batch + k) for k in range(n_inst)]
        arrs[3].zero_()
        engine.run(prog, engine.upload_inputs(prog, np.stack(ins)), *arrs)
        engine.torch.cuda.synchronize()
        assert (arrs[3].cpu().numpy() == 0).all(), (batch, arrs[3].cpu().numpy())


def test_msm_batch_64_tiles_full_size(engine, oracle):
    """BASELINE configs[1] at its batch size: 64 tiles x 1024 points in one run (110 GB of advice arrays): every status
    word 0 once each tile's expected result is fed back, and the last tile cell for cell against the oracle"""
    import os
    t = engine.torch
    n, tiles = 1024, 64
    prog = Program.msm_bn256_tile(n, emit_shape=False)
    ins = np.stack([synth.msm_bn256_tile_inputs(n, tile=600 + k, cheap_points=True, with_expected=False)[0] for k in range(tiles)])
    d_in = engine.upload_inputs(prog, ins)
    arrs = engine.alloc(prog, tiles)
    from halo2ecc_s_amd import engine as E
    fallbacks = engine.get_stat(E.STAT_SCAN_FALLBACKS)
    engine.run(prog, d_in, *arrs)
    t.cuda.synchronize()
    # 64 x 254 windows x 8 chunks + 64 x 254 tail windows with random blinding: the scan predictors never need their fallback
    assert engine.get_stat(E.STAT_SCAN_FALLBACKS) == fallbacks, "scan predictor fell back on random inputs (a performance bug, not a correctness one)"
    refs = prog.outputs()
    Qm = synth.BN_Q
    exp = np.zeros((tiles, 3, 4), dtype=np.uint64)
    for k in range(tiles):
        xs = [engine.read_cell(arrs[0], r, k) for r in refs[0:3]]
        ys = [engine.read_cell(arrs[0], r, k) for r in refs[4:7]]
        z = engine.read_cell(arrs[0], refs[8], k)
        assert z == 0
        exp[k] = synth.pack([sum(v << (108 * i) for i, v in enumerate(xs)) % Qm, sum(v << (108 * i) for i, v in enumerate(ys)) % Qm, 0], 4)
    d_in[:, 4 * n + 6:4 * n + 9, :] = t.from_numpy(exp.view(np.int64)).to(d_in.device)
    arrs[3].zero_()
    engine.run(prog, d_in, *arrs)
    t.cuda.synchronize()
    assert (arrs[3].cpu().numpy() == 0).all(), arrs[3].cpu().numpy()
    k = tiles - 1
    final_in = d_in[k].cpu().numpy().view(np.uint64)
    orun = oracle_lib.run_msm_bn256_tile(n, final_in, threads=os.cpu_count())
    assert orun.info.status == 0, orun.error
    for region in range(3):
        rows = arrs[region].shape[0]
        got = arrs[region][:, :, :, k, :].contiguous().cpu().numpy().view(np.uint64).reshape(rows, arrs[region].shape[1], 4)
        ovals, _ = orun.adv(region, rows)
        assert np.array_equal(got, ovals), f"region {region} differs"
    del arrs
    t.cuda.empty_cache()


@pytest.mark.parametrize("workload", ["msm", "msm_bls12_381", "pairing_bn256"])
def test_unit_records_kernel_matches_reference_indexing(engine, oracle, workload):
    """h2e_unit_records (one kernel behind the C ABI: the per-unit records of a job's one collective) against
    halo2ecc_s_amd.parallel.unit_records, the torch indexing chain the CPU gloo tests run - and against the oracle's result
    point and Offset.  3-limb and 4-limb result points, a workload without a result point, with and without digests, into a
    row block of a wider table (leading index column untouched)."""
    from halo2ecc_s_amd import parallel
    t = engine.torch
    if workload == "msm":
        n = 7
        prog = Program.msm_bn256_tile(n)
        ins = [synth.msm_bn256_tile_inputs(n, tile=300 + k)[0] for k in range(5)]
        orun = oracle_lib.run_msm_bn256_tile(n, ins[3])
    elif workload == "msm_bls12_381":
        n = 2
        prog = Program.msm_bls12_381_tile(n)
        ins = [synth.msm_bls12_381_tile_inputs(n, tile=310 + k)[0] for k in range(3)]
        orun = oracle_lib.run_msm_bls12_381_tile(n, ins[2])
    else:
        prog = Program.pairing_check_bn256()
        ins = [synth.pairing_check_bn256_inputs(instance=320 + k) for k in range(3)]
        orun = None
    units = len(ins)
    d_in = engine.upload_inputs(prog, np.stack(ins))
    arrs = engine.alloc(prog, units)
    dg = engine.run_digest(prog, d_in, *arrs)
    t.cuda.synchronize()
    refs = prog.outputs() if workload != "pairing_bn256" else []
    L = (len(refs) - 3) // 2 if refs else 3
    R = parallel.record_words(L)
    offsets = t.tensor([prog.base_offset, prog.range_offset, prog.select_offset], dtype=t.int64, device=arrs[0].device)
    want = parallel.unit_records(arrs[3], offsets, arrs[0], refs, dg, limbs=L)
    got = engine.unit_records(prog, arrs[0], arrs[3], dg)
    assert got.shape == (units, R) and t.equal(got, want)
    want0 = parallel.unit_records(arrs[3], offsets, arrs[0], refs, None, limbs=L)
    table = t.full((units + 2, R + 3), -7, dtype=t.int64, device=arrs[0].device)
    engine.unit_records(prog, arrs[0], arrs[3], None, out=table[1:1 + units], col0=1)
    assert t.equal(table[1:1 + units, 1:1 + R], want0)
    assert int((table[:, 0] != -7).sum()) == 0 and int((table[0] != -7).sum()) == 0 and int((table[-1] != -7).sum()) == 0 and int((table[:, 1 + R:] != -7).sum()) == 0
    if orun is not None:   # the record's Offset and result point are the oracle's
        k = 3 if workload == "msm" else 2
        assert orun.info.status == 0, orun.error
        rec = [int(x) & (2**64 - 1) for x in got[k].cpu().tolist()]
        assert rec[0] == 0 and rec[1:4] == [orun.info.base_offset, orun.info.range_offset, orun.info.select_offset]
        ovals, _ = orun.adv(0, prog.base_rows)
        for i, ref in enumerate(list(refs[0:L]) + list(refs[L + 1:2 * L + 1])):
            assert rec[4 + 2 * i:6 + 2 * i] == [int(x) for x in ovals[ref & 0x3FFFFFF, (ref >> 27) & 7, 0:2]]
        z = refs[2 * L + 2]
        assert rec[4 + 4 * L] == int(ovals[z & 0x3FFFFFF, (z >> 27) & 7, 0])
        orun.close()
    prog.close()


import golden_util  # noqa: E402


@pytest.mark.parametrize("doc", golden_util.load_all(), ids=lambda d: d["name"])
def test_engine_matches_golden_fixtures(engine, doc):
    """committed fixtures (inputs + SHA-256 of each advice array) reproduced by the HIP engine"""
    k, p = doc["kind"], doc["params"]
    prog = {"int_mul_batch": lambda: Program.int_mul_batch(p["field_pair"], p["n"]),
            "integer_chip_st": lambda: Program.integer_chip_st(p["field_pair"]),
            "msm_bn256_tile": lambda: Program.msm_bn256_tile(p["n"], with_select=p.get("with_select", True))}[k]()
    assert [prog.base_rows, prog.range_rows, prog.select_rows] == doc["rows"]
    assert prog.n_advice_cells == doc["n_advice_cells"] and prog.n_permutations == doc["n_permutations"]
    base, rng, sel, status = _run(engine, prog, [golden_util.inputs_of(doc)])
    assert (status == 0).all()
    for t, name in ((base, "base"), (rng, "range"), (sel, "select")):
        assert golden_util.sha(t[0].cpu().numpy().view(np.uint64)) == doc[name + "_adv_sha256"]
    flags = (prog.base_flags(), prog.range_flags(), prog.select_flags())
    for f, name in zip(flags, ("base", "range", "select")):
        assert golden_util.sha(f) == doc[name + "_flags_sha256"]


@pytest.mark.parametrize("what,k,n_each", [("bls12_381", 8, 2), ("bn256", 8, 8), ("bn256", 3, 5), ("msm", 4, 16), ("msm", 2, 3), ("int_mul", 16, 1)])
def test_several_caller_batches_as_one_run(engine, oracle, what, k, n_each):
    """h2e_run_batches / h2e_submit_batches: k caller batches - own inputs, own batch-interleaved arrays over n_each instances, own
    status words - executed as ONE run of k x n_each instances (a stream of small batches costs runs, not instances: one GPU's share of
    configs[3] / configs[4] at 8 GPUs is 8 / 2 checks per step).  Every batch's three arrays must be what h2e_run of that batch alone
    writes (32-byte digests per instance; arrays 0xFF-poisoned: nothing but the assigned cells may change), every status 0, and one
    instance of the last batch cell for cell as the oracle's - through the packed expansion (pairing batches of <= 32 instances in
    all), the plain one (64 MSM instances) and ragged sizes."""
    if what == "bn256":
        prog, gen, orun_of = Program.pairing_check_bn256(), synth.pairing_check_bn256_inputs, oracle_lib.run_pairing_check_bn256
        make = lambda i: gen(instance=i)   # noqa: E731
    elif what == "bls12_381":
        prog, gen, orun_of = Program.pairing_check_bls12_381(), synth.pairing_check_bls12_381_inputs, oracle_lib.run_pairing_check_bls12_381
        make = lambda i: gen(instance=i)   # noqa: E731
    elif what == "msm":
        prog = Program.msm_bn256_tile(12)
        make = lambda i: synth.msm_bn256_tile_inputs(12, tile=i)[0]   # noqa: E731
        orun_of = lambda inp: oracle_lib.run_msm_bn256_tile(12, inp)   # noqa: E731
    else:
        prog = Program.int_mul_batch(1, 3)
        make = lambda i: synth.int_mul_batch_inputs(1, 3, seed_index=i)   # noqa: E731
        orun_of = lambda inp: oracle_lib.run_int_mul_batch(1, 3, inp)   # noqa: E731
    t = engine.torch
    ins = [[make(3000 + 40 * b + j) for j in range(n_each)] for b in range(k)]
    d_in = [engine.upload_inputs(prog, np.stack(x)) for x in ins]
    want = []
    ref = engine.alloc(prog, n_each, fill=0xFF)
    for b in range(k):
        ref[3].zero_()
        engine.run(prog, d_in[b], *ref)
        t.cuda.synchronize()
        assert (ref[3].cpu().numpy() == 0).all(), ref[3].cpu().numpy()
        want.append([engine.digest(prog, region, ref[region]).cpu().numpy() for region in range(3)])
    raw_ref = [a.clone() for a in ref[:3]]   # batch k - 1 alone, poison included
    for mode in ("run", "submit"):
        bufs = [engine.alloc(prog, n_each, fill=0xFF) for _ in range(k)]
        for b in range(k):
            bufs[b][3].zero_()
        batches = [(d_in[b],) + tuple(bufs[b]) for b in range(k)]
        if mode == "run":
            engine.run_batches(prog, batches)
        else:
            engine.wait(engine.submit_batches(prog, batches))
        t.cuda.synchronize()
        for b in range(k):
            assert (bufs[b][3].cpu().numpy() == 0).all(), (mode, b, bufs[b][3].cpu().numpy())
            for region in range(3):
                got = engine.digest(prog, region, bufs[b][region]).cpu().numpy()
                assert np.array_equal(got, want[b][region]), (mode, b, region)
        for region in range(3):   # bytes, not only assigned cells: the poison around them is untouched exactly as in the single run
            assert t.equal(bufs[k - 1][region], raw_ref[region]), (mode, region)
    orun = orun_of(ins[k - 1][n_each - 1])
    assert orun.info.status == 0, orun.error
    compare_advice(prog, orun, *_rows(engine, prog, bufs[k - 1][:3]), instance=n_each - 1)
    orun.close()


def test_batches_argument_checks(engine):
    prog = Program.int_mul_batch(0, 1)
    d_in = engine.upload_inputs(prog, np.stack([synth.int_mul_batch_inputs(0, 1)]))
    a, b = engine.alloc(prog, 1), engine.alloc(prog, 1)
    from halo2ecc_s_amd.engine import H2EError
    with pytest.raises(H2EError, match="share an output array"):
        engine.run_batches(prog, [(d_in,) + tuple(a), (d_in,) + tuple(a)])
    with pytest.raises(H2EError, match="n_batches"):
        engine.run_batches(prog, [(d_in,) + tuple(engine.alloc(prog, 1)) for _ in range(17)])
    engine.run_batches(prog, [(d_in,) + tuple(a), (d_in,) + tuple(b)])
    engine.torch.cuda.synchronize()
    assert engine.torch.equal(a[0], b[0]) and int(a[3][0]) == 0


@pytest.mark.parametrize("what,n_inst,depth", [("msm", 5, 3), ("msm", 64, 3), ("msm", 3, 4)])
def test_ring_three_runs_in_flight_share_the_big_launch_rows(engine, oracle, what, n_inst, depth):
    """h2e_ring: `depth` runs in flight whose biggest launch's rows (the MSM's window strands) are backed by TWO physical copies - run k
    and run k + 2 write the same physical rows - with everything of run k that writes them fenced behind the completion of run k - 2.
    Nine runs of DIFFERENT batches back to back, `depth` of them in flight, consumed the way the ring is made for: the STREAM DIGEST of
    every run (all three arrays, every assigned cell, accumulated by the expansion while it stores - the shared rows of run k may be
    gone once run k + 2 is submitted) must equal the stream digest of the same batch through h2e_run_digest into plain arrays, every
    status 0, and the last run - nothing submitted after it - cell for cell as the oracle's; and the aliasing is real (set k and set
    k + 2 share addresses in the big launch's rows, and only there)."""
    from halo2ecc_s_amd import Ring
    if what == "msm":
        prog = Program.msm_bn256_tile(33)
        make = lambda i: synth.msm_bn256_tile_inputs(33, tile=i)[0]   # noqa: E731
        orun_of = lambda inp: oracle_lib.run_msm_bn256_tile(33, inp)   # noqa: E731
    else:
        prog = Program.pairing_check_bn256()
        make = lambda i: synth.pairing_check_bn256_inputs(instance=i)   # noqa: E731
        orun_of = oracle_lib.run_pairing_check_bn256
    t = engine.torch
    n_runs, n_batches = 9, 4
    ins = [[make(5000 + 70 * b + j) for j in range(n_inst)] for b in range(n_batches)]
    d_in = [engine.upload_inputs(prog, np.stack(x)) for x in ins]
    want = []
    ref = engine.alloc(prog, n_inst, fill=0xFF)
    for b in range(n_batches):
        ref[3].zero_()
        want.append(engine.run_digest(prog, d_in[b], *ref).cpu().numpy().copy())
        t.cuda.synchronize()
        assert (ref[3].cpu().numpy() == 0).all()
    del ref
    old_depth = engine.get_stat(3)
    engine.set_option(4, depth)
    ring = None
    try:
        ring = Ring(engine, prog, n_inst, depth)
        info = ring.info
        assert info["depth"] == depth and info["virtual_sets"] == (depth if depth % 2 == 0 else 2 * depth)
        full = sum(info["set_bytes"])
        assert info["physical_bytes"] == 2 * sum(info["shared_bytes"]) + depth * (full - sum(info["shared_bytes"])) < depth * full
        if what == "msm" and n_inst == 64:
            assert sum(info["shared_bytes"]) > 0.5 * full       # the window strands own most rows (pieces meet at 2 MB boundaries: small shapes share less)
        rows = prog.launch_rows(info["shared_launch"])
        a0, a1, a2 = ring.arrays(0)[0], ring.arrays(1)[0], ring.arrays(2)[0]
        if info["shared_bytes"][0] > 0:
            # the aliasing: a cell of the shared launch written through set 0 reads back through set 2 (and not through set 1); a cell
            # outside it does not (a row 2 MB into the launch's rows: inside the shared piece wherever its boundary was rounded to)
            row_bytes = 5 * 32 * n_inst
            r_in, r_out = int(rows[0]) + (2 << 20) // row_bytes + 1, 0
            for a in (a0, a1, a2):
                a[r_in].zero_()
                a[r_out].zero_()
            a0[r_in] += 7
            a0[r_out] += 9
            t.cuda.synchronize()
            assert int(a2[r_in].flatten()[0]) == 7 and int(a1[r_in].flatten()[0]) == 0
            assert int(a2[r_out].flatten()[0]) == 0 and int(a1[r_out].flatten()[0]) == 0
        for v in range(info["virtual_sets"]):
            for a in ring.arrays(v):
                a.fill_(-1)                              # poison: only assigned cells may change
        status = [t.zeros((n_inst,), dtype=t.int32, device=a0.device) for _ in range(depth)]
        dgs = [t.zeros((3, n_inst, 4), dtype=t.int64, device=a0.device) for _ in range(depth)]
        got, pending = {}, []

        def retire(k0, job0):
            engine.wait(job0)
            got[k0] = (dgs[k0 % depth].clone(), status[k0 % depth].clone())

        for k in range(n_runs):
            while len(pending) >= depth:
                retire(*pending.pop(0))
            status[k % depth].zero_()
            pending.append((k, ring.submit(k, d_in[k % n_batches], status[k % depth], digests=dgs[k % depth])))
            while len(pending) > depth - 1:              # as bench.py: consume the oldest while the newer ones are in flight
                retire(*pending.pop(0))
        for k0, job0 in pending:
            retire(k0, job0)
        t.cuda.synchronize()
        for k in range(n_runs):
            dg, st = got[k]
            assert (st.cpu().numpy() == 0).all(), (k, st.cpu().numpy())
            assert np.array_equal(dg.cpu().numpy(), want[k % n_batches]), k
        last = n_runs - 1
        orun = orun_of(ins[last % n_batches][n_inst - 1])
        assert orun.info.status == 0, orun.error
        compare_advice(prog, orun, *_rows(engine, prog, ring.arrays(last)), instance=n_inst - 1)
        orun.close()
        from halo2ecc_s_amd.engine import H2EError
        with pytest.raises(H2EError, match="in order"):
            ring.submit(n_runs + 3, d_in[0], status[0])
    finally:
        t.cuda.synchronize()
        if ring is not None:
            ring.close()
        engine.set_option(4, old_depth)


def test_ring_needs_a_forked_launch_and_the_matching_depth(engine):
    from halo2ecc_s_amd import Ring
    from halo2ecc_s_amd.engine import H2EError
    old_depth = engine.get_stat(3)
    try:
        engine.set_option(4, 3)
        with pytest.raises(H2EError, match="no forked launch"):
            Ring(engine, Program.integer_chip_st(0), 2, 3)
        with pytest.raises(H2EError, match="PIPELINE_DEPTH"):
            Ring(engine, Program.msm_bn256_tile(4), 2, 2)
    finally:
        engine.set_option(4, old_depth)


@pytest.mark.parametrize("what", ["int_mul", "integer_chip", "msm", "pairing", "integer_chip_bls_fr", "msm_bls12_381", "pairing_bls12_381"])
def test_columns_straight_out_of_the_expansion(engine, oracle, what):
    """h2e_run_columns: the expansion stores halo2's per-instance advice columns itself (LDS-staged 128-byte runs per instance) - the
    arrays must equal h2e_export(H2E_LAYOUT_COLUMNS) of a plain run of the same batch bit for bit: every assigned cell, zeros everywhere
    else (the column arrays are zeroed once and re-used: a second, different batch into the same arrays must come out right as well),
    every status 0; and one instance against the oracle through the exported rows of the plain run (same batch)."""
    from halo2ecc_s_amd.engine import LAYOUT_COLUMNS, FORM_CANONICAL
    n = 64
    if what == "int_mul":
        prog = Program.int_mul_batch(0, 5)
        make = lambda i: synth.int_mul_batch_inputs(0, 5, seed_index=i)   # noqa: E731
    elif what == "integer_chip":
        prog = Program.integer_chip_st(0)
        make = lambda i: synth.integer_chip_st_inputs(0, seed_index=i)   # noqa: E731
    elif what == "msm":
        prog = Program.msm_bn256_tile(33)
        make = lambda i: synth.msm_bn256_tile_inputs(33, tile=i)[0]   # noqa: E731
    elif what == "integer_chip_bls_fr":   # (3-limb bls12_381 Fr over bn256 Fr: the third field pair's unit)
        prog = Program.integer_chip_st(2)
        make = lambda i: synth.integer_chip_st_inputs(2, seed_index=i)   # noqa: E731
    elif what == "msm_bls12_381":         # (a general-scalar tile: 4-limb bls12_381 Fq segments and 3-limb Fr segments in one program)
        prog = Program.msm_bls12_381_tile(7)
        make = lambda i: synth.msm_bls12_381_tile_inputs(7, tile=i)[0]   # noqa: E731
    elif what == "pairing_bls12_381":
        prog = Program.pairing_check_bls12_381()
        make = lambda i: synth.pairing_check_bls12_381_inputs(instance=i)   # noqa: E731
    else:
        prog = Program.pairing_check_bn256()
        make = lambda i: synth.pairing_check_bn256_inputs(instance=i)   # noqa: E731
    t = engine.torch
    cols = engine.alloc_columns(prog, n)
    for batch in range(2):
        ins = [make(7000 + 100 * batch + j) for j in range(n)]
        d_in = engine.upload_inputs(prog, np.stack(ins))
        plain = engine.alloc(prog, n, fill=0xFF)
        plain[3].zero_()
        engine.run(prog, d_in, *plain)
        t.cuda.synchronize()
        assert (plain[3].cpu().numpy() == 0).all(), plain[3].cpu().numpy()
        want = [engine.export(prog, region, plain[region], layout=LAYOUT_COLUMNS, form=FORM_CANONICAL) for region in range(3)]
        t.cuda.synchronize()
        del plain
        work = engine.alloc(prog, n, fill=0xFF)
        work[3].zero_()
        engine.run_columns(prog, d_in, *work, cols)
        t.cuda.synchronize()
        assert (work[3].cpu().numpy() == 0).all(), work[3].cpu().numpy()
        for region in range(3):
            if not t.equal(cols[region], want[region]):
                bad = (cols[region] != want[region]).any(dim=3).nonzero()
                raise AssertionError(f"batch {batch} region {region}: {bad.shape[0]} cells differ, first (instance, col, row) {bad[:8].tolist()}")
        del work, want


def test_ring_with_a_consumer_that_reads_the_shared_rows(engine, oracle):
    """h2e_ring_release: a consumer that READS every array of a finished run (h2e_digest - the same reads an export makes) while three
    runs are in flight: it works on its own stream (`h2e_wait` + digests + release there), the submissions go out on another, and run
    k + 2 - the next writer of the shared launch's physical rows - waits for the point where run k's reads end.  Ten runs of different
    batches: every run's post-run digests of all three arrays equal the same batch's through h2e_run into plain arrays."""
    from halo2ecc_s_amd import Ring
    prog = Program.msm_bn256_tile(33)
    t = engine.torch
    n_inst, depth, n_runs, n_batches = 64, 3, 10, 4
    ins = [[synth.msm_bn256_tile_inputs(33, tile=9000 + 70 * b + j)[0] for j in range(n_inst)] for b in range(n_batches)]
    d_in = [engine.upload_inputs(prog, np.stack(x)) for x in ins]
    want = []
    ref = engine.alloc(prog, n_inst, fill=0xFF)
    for b in range(n_batches):
        ref[3].zero_()
        engine.run(prog, d_in[b], *ref)
        t.cuda.synchronize()
        assert (ref[3].cpu().numpy() == 0).all()
        want.append([engine.digest(prog, region, ref[region]).cpu().numpy() for region in range(3)])
    del ref
    old_depth = engine.get_stat(3)
    engine.set_option(4, depth)
    ring = None
    try:
        ring = Ring(engine, prog, n_inst, depth)
        for v in range(ring.info["virtual_sets"]):
            for a in ring.arrays(v):
                a.fill_(-1)
        status = [t.zeros((n_inst,), dtype=t.int32, device="cuda") for _ in range(n_runs)]   # (one per run: nothing of the test's own to order)
        submit_s, consume_s = t.cuda.Stream(), t.cuda.Stream()
        t.cuda.synchronize()
        got, jobs = {}, {}

        def consume(k0):
            engine.wait(jobs[k0], stream=consume_s)
            arrs = ring.arrays(k0)
            got[k0] = ([engine.digest(prog, region, arrs[region], stream=consume_s) for region in range(3)], None)
            with t.cuda.stream(consume_s):
                got[k0] = (got[k0][0], status[k0].clone())
            ring.release(k0, stream=consume_s)

        for k in range(n_runs):
            if k >= 2:
                consume(k - 2)          # enqueued on the consumer's stream: the host does not wait, the submission stream knows nothing of it
            jobs[k] = ring.submit(k, d_in[k % n_batches], status[k], stream=submit_s)
        consume(n_runs - 2)
        consume(n_runs - 1)
        t.cuda.synchronize()
        for k in range(n_runs):
            dg, st = got[k]
            assert (st.cpu().numpy() == 0).all(), (k, st.cpu().numpy())
            for region in range(3):
                assert np.array_equal(dg[region].cpu().numpy(), want[k % n_batches][region]), (k, region)
    finally:
        t.cuda.synchronize()
        if ring is not None:
            ring.close()
        engine.set_option(4, old_depth)
