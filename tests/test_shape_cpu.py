"""CPU tests: the recorder's shape artefacts (everything in Records except advice values) must equal
the oracle's, and the oracle must satisfy the reference's constraint system on the same inputs."""
import numpy as np
import pytest

import oracle_lib
from halo2ecc_s_amd import Program, synth
from parity import compare_shape


@pytest.mark.parametrize("fp", [0, 1, 2])
def test_int_mul_batch_shape(oracle, h2e_built, fp):
    n = 5
    inputs = synth.int_mul_batch_inputs(fp, n)
    prog = Program.int_mul_batch(fp, n)
    orun = oracle_lib.run_int_mul_batch(fp, n, inputs)
    ok, msg = orun.check()
    assert ok, msg
    compare_shape(prog, orun)
    # config 1 of BASELINE.json: one bn256 modmul = 17 base + 28 range rows (+ 2 x assign_w)
    if fp == 0:
        assert prog.base_offset == n * (17 + 2) and prog.range_offset == n * (28 + 16)


@pytest.mark.parametrize("fp", [0, 1, 2])
def test_integer_chip_st_shape(oracle, h2e_built, fp):
    inputs = synth.integer_chip_st_inputs(fp)
    prog = Program.integer_chip_st(fp)
    orun = oracle_lib.run_integer_chip_st(fp, inputs)
    ok, msg = orun.check()
    assert ok, msg
    compare_shape(prog, orun)


@pytest.mark.parametrize("n", [1, 4, 7, 11])
def test_msm_tile_shape(oracle, h2e_built, n):
    inputs, _ = synth.msm_bn256_tile_inputs(n)
    prog = Program.msm_bn256_tile(n)
    orun = oracle_lib.run_msm_bn256_tile(n, inputs)
    ok, msg = orun.check()
    assert ok, msg
    compare_shape(prog, orun)


@pytest.mark.parametrize("n", [1, 3, 6])
def test_msm_tile_no_select_shape(oracle, h2e_built, n):
    """the same body on a context without the select chip (msm_batch_on_group_non_zero_without_
    select_chip, ecc_chip.rs:91-221: groups of two, bisec_candidate_non_zero).  The oracle's witness satisfies the
    reference's constraints (no select rows); the recorder's shape artefacts equal the oracle's."""
    inputs, _ = synth.msm_bn256_tile_inputs(n, tile=40 + n)
    prog = Program.msm_bn256_tile(n, with_select=False)
    orun = oracle_lib.run_msm_bn256_tile(n, inputs, with_select=False)
    assert orun.info.status == 0, orun.error
    ok, msg = orun.check()
    assert ok, msg
    assert prog.select_offset == 0
    compare_shape(prog, orun)


@pytest.mark.parametrize("knob", [("H2E_NO_FIELD_CHAIN", "1"), ("H2E_FIELD_NO_SINKS", "1"), ("H2E_FIELD_NO_INLINE", "1"),
                                  ("H2E_FIELD_NO_REBALANCE", "1"), ("H2E_FIELD_NO_LONG", "1"), ("H2E_FIELD_NO_PAIRS", "1"), ("H2E_FIELD_STEP", "54")],
                         ids=["level_parallel_replay", "sinks_in_chain", "no_inlining", "no_rebalancing", "no_long_combinations", "no_product_pairs", "one_pass_rounds"])
def test_pairing_value_chain_variants_compile(h2e_built, knob):
    """every form of the pairing checks' value chain (tests/test_parity_gpu.py::test_pairing_value_chain_variants runs them) compiles on
    the host, for the same rows: the knobs only choose how the hints are computed"""
    import os
    ref = Program.pairing_check_bn256(emit_shape=False)
    name, value = knob
    old = os.environ.get(name)
    os.environ[name] = value
    try:
        prog = Program.pairing_check_bn256(emit_shape=False)
    finally:
        if old is None:
            del os.environ[name]
        else:
            os.environ[name] = old
    assert (prog.base_offset, prog.range_offset, prog.select_offset, prog.n_advice_cells) == (ref.base_offset, ref.range_offset, ref.select_offset, ref.n_advice_cells)


def test_pairing_check_bn256_shape(oracle, h2e_built):
    """config 4 unit (2-pair bn256 check_pairing, G2 as per-instance constants -> fixed patches)"""
    inputs = synth.pairing_check_bn256_inputs()
    prog = Program.pairing_check_bn256()
    orun = oracle_lib.run_pairing_check_bn256(inputs)
    assert orun.info.status == 0, orun.error   # e(a,b) * e(-a,b) == 1 held in-circuit
    compare_shape(prog, orun, patches_inputs=inputs)
    assert prog.n_advice_cells == 6165013


@pytest.mark.parametrize("curve", ["bn256", "bls12_381"])
def test_packed_expansion_order_tables(h2e_built, curve):
    """the order the packed expansion takes a launch's sub-ranges in (batches smaller than half a wave: BASELINE's 8-GPU shares of
    configs[3] / configs[4], 16 bls12_381 checks on one GPU): for every group count every sub-range exactly once, the sub-ranges of
    a wave all have the SAME opcode sequence (so its groups never wait for each other's ops), empty slots only behind a wave's
    sub-ranges, and the pairing programs repeat themselves enough for that to cost few extra waves"""
    prog = (Program.pairing_check_bn256 if curve == "bn256" else Program.pairing_check_bls12_381)(emit_shape=False)
    launches = [i for i, l in enumerate(prog.launches()) if l["n_ops"] > 1000]
    assert len(launches) == 2   # Miller loop | final exponentiation
    for li in launches:
        ops, subs = prog.tape_opcodes(li)
        n_sub = len(subs) - 1
        seq = [ops[subs[k]:subs[k + 1]].tobytes() for k in range(n_sub)]
        assert len(set(seq)) < n_sub // 4
        for groups in (2, 4, 8, 16, 32):
            tab = prog.pack_order(li, groups)
            live = tab[tab != 0xFFFFFFFF]
            assert np.array_equal(np.sort(live), np.arange(n_sub, dtype=np.uint32))
            for w in tab:
                k = int((w != 0xFFFFFFFF).sum())
                assert k >= 1 and (w[k:] == 0xFFFFFFFF).all()
                assert len({seq[s] for s in w[:k]}) == 1
            assert len(tab) <= -(-n_sub // groups) + len(set(seq))   # at most one padded wave per opcode sequence
