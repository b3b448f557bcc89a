#!/usr/bin/env python3
"""Generate tests/golden/pyref/*.json with the independent Python restatement (oracle/pyref.py, written from the Rust
sources): seeded inputs + offsets, heights, op counts, permutation-list hash, flag hashes and the 32-byte digest of
every advice / fixed array.  tests/test_pyref_cpu.py then holds the C++ oracle and the engine's recorder against them.

The reference itself holds no golden vectors for this path and cannot be built here (no Rust toolchain, un-vendored git
dependencies), so parity stays "unpinned by the reference"; what these fixtures pin is that two restatements that share
nothing but the reference's text agree cell for cell.  When /root/reference is present this script also checks the
constants oracle/pyref.py derives mathematically (Frobenius coefficients, xi^((q-1)/2)) against the reference's tables
(src/circuit/bn256_constants.rs, src/circuit/bls12_381_pairing_chip.rs:58-107).

Run from the repo root:  python tests/golden/make_pyref_golden.py [--big | --only-next]   (--big adds the 1024-point tile: ~25 min)
"""
import hashlib
import json
import os
import re
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np  # noqa: E402
import pyref  # noqa: E402
from halo2ecc_s_amd import synth  # noqa: E402

OUT = os.path.join(HERE, "pyref")
REF = "/root/reference/src/circuit"


def check_constants_against_reference():
    if not os.path.isdir(REF):
        print("reference not present: constants not re-checked")
        return
    t = open(os.path.join(REF, "bn256_constants.rs")).read()

    def table(name):
        m = re.search(name + r":[^=]*=\s*(\[.*?\]);", t, re.S)
        nums = [int(x) for x in re.findall(r"\d+", m.group(1))]
        assert len(nums) % 32 == 0
        return [int.from_bytes(bytes(nums[k:k + 32]), "little") for k in range(0, len(nums), 32)]

    k = pyref.bn256_frobenius_constants()
    assert table("XI_TO_Q_MINUS_1_OVER_2") == list(k["xi_q12"])
    assert table("FROBENIUS_COEFF_FQ2_C1") == k["fq2_c1"]
    for name, key in (("FROBENIUS_COEFF_FQ6_C1", "fq6_c1"), ("FROBENIUS_COEFF_FQ6_C2", "fq6_c2"), ("FROBENIUS_COEFF_FQ12_C1", "fq12_c1")):
        flat = table(name)
        assert [tuple(flat[2 * i:2 * i + 2]) for i in range(len(flat) // 2)] == k[key], name
    m = re.search(r"SIX_U_PLUS_2_NAF: \[i8; 65\] = \[(.*?)\];", t, re.S)
    assert [int(x) for x in re.findall(r"-?\d+", m.group(1))] == pyref.SIX_U_PLUS_2_NAF
    assert int(re.search(r"BN_X: u64 = (\d+);", t).group(1)) == pyref.BN_X
    # bls12_381: Montgomery-form raw limbs (R = 2^384) -> canonical
    b = open(os.path.join(REF, "bls12_381_pairing_chip.rs")).read()
    raws = re.findall(r"from_raw_unchecked\(\[(.*?)\]\)", b, re.S)
    vals = []
    for r in raws:
        limbs = [int(x.replace("_", ""), 16) for x in re.findall(r"0x[0-9a-f_]+", r)]
        raw = sum(l << (64 * i) for i, l in enumerate(limbs))
        vals.append(raw * pow(1 << 384, -1, pyref.BLS_Q) % pyref.BLS_Q)
    kb = pyref.bls12_381_frobenius_constants()
    assert (0, vals[0]) == kb["fq6_c1"] and (vals[1], 0) == kb["fq6_c2"] and (vals[2], vals[3]) == kb["fq12_c1"]
    assert int(re.search(r"BLS_X: u64 = (0x[0-9a-f_]+);", b).group(1).replace("_", ""), 16) == pyref.BLS_X
    print("constants of oracle/pyref.py == the reference's tables")


def write(name, kind, params, inputs, ctx, secs, inline_inputs=True):
    doc = {"name": name, "kind": kind, "params": params,
           "generator": "oracle/pyref.py (independent Python restatement of the reference), tests/golden/make_pyref_golden.py",
           "pyref_seconds": round(secs, 1),
           "inputs_sha256": hashlib.sha256(np.ascontiguousarray(inputs, dtype=np.uint64).tobytes()).hexdigest(),
           "pyref": pyref.summary(ctx)}
    if inline_inputs:
        doc["inputs_hex"] = [[hex(int(w)) for w in slot] for slot in inputs]
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(doc, f, indent=1)
    s = doc["pyref"]
    print(f"{name}: {secs:.0f} s, offsets {s['offsets']}, heights {s['heights']}, {s['n_advice_cells']} cells, counts {s['counts']}", flush=True)


def base_blocks():
    t = time.time()
    inp = synth.pairing_check_bn256_inputs(instance=1)
    write("pairing_check_bn256_i1", "pairing_check_bn256", {"instance": 1}, inp, pyref.run_pairing_check_bn256(inp), time.time() - t)
    t = time.time()
    inp = synth.pairing_check_bls12_381_inputs(instance=1)
    write("pairing_check_bls12_381_i1", "pairing_check_bls12_381", {"instance": 1}, inp, pyref.run_pairing_check_bls12_381(inp), time.time() - t)
    for n, tile, sel in ((33, 2, True), (12, 61, False)):
        t = time.time()
        inp, _ = synth.msm_bn256_tile_inputs(n, tile=tile)
        write(f"msm_bn256_tile_n{n}" + ("" if sel else "_no_select"), "msm_bn256_tile", {"n": n, "tile": tile, "with_select": sel}, inp,
              pyref.run_msm_bn256_tile(n, inp, with_select=sel), time.time() - t)
    for fp in (0, 1, 2):
        t = time.time()
        inp = synth.integer_chip_st_inputs(fp, seed_index=21)
        write(f"integer_chip_st_fp{fp}", "integer_chip_st", {"field_pair": fp, "seed_index": 21}, inp, pyref.run_integer_chip_st(fp, inp), time.time() - t)


def next_rows_blocks():
    # general-scalar MSM (tests/general_scalar_ecc_chip.rs:14-49 at 7 points: one full group of 4 + a remainder group)
    t = time.time()
    n = 7
    inp, _ = synth.msm_bls12_381_tile_inputs(n, tile=3)
    write(f"msm_bls12_381_tile_n{n}", "msm_bls12_381_tile", {"n": n, "tile": 3}, inp, pyref.run_msm_bls12_381_tile(n, inp), time.time() - t)
    # pairing(terms) == native (tests/native_scalar_pairing_chip.rs:20-65): the expected Fq12 constant is what pyref's
    # own pairing computes (the reference takes it from the curve library); fq12_assert_eq must then hold
    t = time.time()
    inp = synth.pairing_inputs(0, 1, instance=2)
    res = pyref.run_pairing(0, 1, False, inp).s.marks["result"]
    inp = synth.pairing_inputs(0, 1, instance=2, expected=res)
    write("pairing_bn256_1pair_expected", "pairing", {"curve": 0, "n_pairs": 1, "with_expected": True, "instance": 2}, inp,
          pyref.run_pairing(0, 1, True, inp), time.time() - t)


def big_blocks():
    t = time.time()
    n = 1024
    inp, _ = synth.msm_bn256_tile_inputs(n, tile=100, cheap_points=True)
    write("msm_bn256_tile_n1024", "msm_bn256_tile", {"n": n, "tile": 100, "with_select": True, "cheap_points": True}, inp,
          pyref.run_msm_bn256_tile(n, inp), time.time() - t, inline_inputs=False)


def main():
    """no flag: everything but the 1024-point tile; --big: that tile too; --only-next: just the fixtures of the 8(f) rows"""
    check_constants_against_reference()
    if "--only-next" not in sys.argv:
        base_blocks()
    next_rows_blocks()
    if "--big" in sys.argv:
        big_blocks()


if __name__ == "__main__":
    main()
