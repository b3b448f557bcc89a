"""CPU tests of the oracle itself: golden fixtures, constraint checker self-test, constants."""
import os
import subprocess
import sys

import numpy as np
import pytest

import golden_util
import oracle_lib
from halo2ecc_s_amd import Program, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_run(doc):
    inp = golden_util.inputs_of(doc)
    k, p = doc["kind"], doc["params"]
    if k == "int_mul_batch":
        return oracle_lib.run_int_mul_batch(p["field_pair"], p["n"], inp)
    if k == "integer_chip_st":
        return oracle_lib.run_integer_chip_st(p["field_pair"], inp)
    if k == "msm_bn256_tile":
        return oracle_lib.run_msm_bn256_tile(p["n"], inp, with_select=p.get("with_select", True))
    raise AssertionError(k)


@pytest.mark.parametrize("doc", golden_util.load_all(), ids=lambda d: d["name"])
def test_oracle_reproduces_golden(oracle, doc):
    run = _oracle_run(doc)
    i = run.info
    assert i.status == 0
    assert [i.base_offset, i.range_offset, i.select_offset] == doc["offsets"]
    assert [i.base_height, i.range_height, i.select_height] == doc["heights"]
    assert i.n_advice_cells == doc["n_advice_cells"] and i.n_permutations == doc["n_permutations"]
    for region, name in enumerate(("base", "range", "select")):
        vals, flags = run.adv(region, doc["rows"][region])
        assert golden_util.sha(vals) == doc[name + "_adv_sha256"]
        assert golden_util.sha(flags) == doc[name + "_flags_sha256"]


def test_checker_rejects_corruption(oracle):
    """the MockProver-equivalent must catch a wrong advice cell in every chip"""
    inp, _ = synth.msm_bn256_tile_inputs(2)
    for region, row, col in ((0, 0, 4), (0, 0, 0), (1, 0, 0), (1, 1, 1), (2, 5, 0)):
        run = oracle_lib.run_msm_bn256_tile(2, inp)
        ok, _ = run.check()
        assert ok
        run.corrupt(region, row, col)
        ok, msg = run.check()
        assert not ok, (region, row, col)
        run.close()


def test_oracle_panics_like_reference(oracle):
    """a wrong expected MSM result fails the in-circuit assert (reference: assert_true panics, base_chip.rs:487-490)"""
    inp, _ = synth.msm_bn256_tile_inputs(2, with_expected=False)
    run = oracle_lib.run_msm_bn256_tile(2, inp)
    assert run.info.status == 1 and "assert" in run.error


def test_add_same_point_is_unsafe_error(oracle):
    """two equal points in one group make ecc_add_unsafe hit x1 == x2.  In the reference, try_assert_false's
    assert_constant panics before the bool is formed (base_chip.rs:375-379, :497-500): status 1, not a retry."""
    inp, _ = synth.msm_bn256_tile_inputs(2)
    inp[3:6] = inp[0:3]  # point 1 := point 0
    run = oracle_lib.run_msm_bn256_tile(2, inp)
    assert run.info.status in (1, 2)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference only exists in the authoring container")
def test_constants_equal_reference_tables():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "gen_pairing_constants.py"), "verify",
                                   "/root/reference"]).decode()
    assert "all tables equal" in out


def test_generated_constant_headers_are_current():
    import tempfile
    for ns, path in (("h2o_const", "oracle/pairing_constants.hpp"), ("h2e_const", "halo2ecc_s_amd/csrc/pairing_constants.hpp")):
        with tempfile.NamedTemporaryFile(suffix=".hpp") as t:
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_pairing_constants.py"), "emit", t.name, ns])
            assert open(t.name).read() == open(os.path.join(ROOT, path)).read()


def test_c_abi_exports_every_declared_symbol(h2e_built):
    """libh2e.so loads and exports every function include/h2e.h declares (no compute calls without a GPU)"""
    import ctypes
    import re
    from halo2ecc_s_amd.engine import EXPORTED_SYMBOLS
    L = ctypes.CDLL(h2e_built)
    hdr = open(os.path.join(ROOT, "include", "h2e.h")).read()
    declared = set(re.findall(r"\b(h2e_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(EXPORTED_SYMBOLS), declared ^ set(EXPORTED_SYMBOLS)
    for s in declared:
        getattr(L, s)


def test_engine_fails_loudly_without_gpu(h2e_built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from halo2ecc_s_amd import Engine, H2EError
    with pytest.raises(H2EError):
        Engine(0)
