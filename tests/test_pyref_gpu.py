"""The HIP engine against the INDEPENDENT Python restatement's committed fixtures, with nothing in between.

tests/golden/pyref/*.json were traced by oracle/pyref.py (written from the Rust sources; shares no text with the C++ oracle
or with the recorder; tests/golden/make_pyref_golden.py).  Every other `-m gpu` parity test compares the engine with the C++
oracle, whose L4 text the recorder's tower / pairing ops resemble - here the chain is one link:
engine arrays in HBM -> h2e_digest (include/h2e.h: the position-keyed 32-byte digest of every assigned cell of an array)
== the fixture's `adv_digest`, for both pairing checks, pairing() == expected, the MSM tiles (select chip, no select chip,
bls12_381 with general scalars, BASELINE's 1024-point tile) and the integer chip on the three field pairs.  Offsets, heights,
cell and permutation counts and the permutation list itself are compared with the fixture as well.  Nothing here loads oracle/.

What the digest covers: the cells the RECORDER's own flag arrays mark as assigned (h2e_digest masks with the program's flags).  Those
flags themselves are pinned to the Python restatement on the CPU side - tests/test_pyref_cpu.py compares the recorder's assigned /
flags (SHA-256 per region), offsets, heights, cell count and permutation list with the same fixtures (`assert_recorder_matches`) - so a cell the recorder wrongly left unflagged would fail there (and its
count here: `n_advice_cells` is compared with the fixture's), not silently drop out of this digest."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from halo2ecc_s_amd import Program, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "pyref", "*.json")))

MAKE = {
    "int_mul_batch": lambda p: Program.int_mul_batch(p["field_pair"], p["n"]),
    "integer_chip_st": lambda p: Program.integer_chip_st(p["field_pair"]),
    "msm_bn256_tile": lambda p: Program.msm_bn256_tile(p["n"], with_select=p.get("with_select", True)),
    "msm_bls12_381_tile": lambda p: Program.msm_bls12_381_tile(p["n"]),
    "pairing": lambda p: Program.pairing(p["curve"], p["n_pairs"], p["with_expected"]),
    "pairing_check_bn256": lambda p: Program.pairing_check_bn256(),
    "pairing_check_bls12_381": lambda p: Program.pairing_check_bls12_381(),
}


def _inputs(doc):
    if "inputs_hex" in doc:
        return np.array([[int(w, 16) for w in slot] for slot in doc["inputs_hex"]], dtype=np.uint64)
    p = doc["params"]   # the 1024-point tile: regenerated from its seed (8 MB as hex), pinned by inputs_sha256
    assert doc["kind"] == "msm_bn256_tile"
    return synth.msm_bn256_tile_inputs(p["n"], tile=p["tile"], cheap_points=p.get("cheap_points", False))[0]


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[:-5])
def test_engine_reproduces_pyref_fixture(engine, path):
    with open(path) as f:
        doc = json.load(f)
    fx = doc["pyref"]
    inp = _inputs(doc)
    assert hashlib.sha256(np.ascontiguousarray(inp, dtype=np.uint64).tobytes()).hexdigest() == doc["inputs_sha256"], "synthetic inputs changed"
    prog = MAKE[doc["kind"]](doc["params"])
    assert fx["offsets"] == [prog.base_offset, prog.range_offset, prog.select_offset]
    assert fx["heights"] == [prog.base_height, prog.range_height, prog.select_height]
    assert fx["n_advice_cells"] == prog.n_advice_cells and fx["n_permutations"] == prog.n_permutations
    assert fx["permutations_sha256"] == hashlib.sha256(np.ascontiguousarray(prog.permutations(), dtype="<u4").tobytes()).hexdigest()
    # the fixture's instance as the LAST of three (the others: the same inputs with the first words of slot 0 disturbed would not be
    # valid witnesses; so the same instance three times, into poisoned arrays: lanes 0..2 of a wave must all give the fixture's digest)
    n_inst = 3 if prog.n_advice_cells < 10_000_000 else 1
    d_in = engine.upload_inputs(prog, np.stack([inp] * n_inst))
    arrs = engine.alloc(prog, n_inst, fill=0xFF)
    engine.run(prog, d_in, *arrs)
    dg = [engine.digest(prog, region, arrs[region]) for region in range(3)]
    engine.torch.cuda.synchronize()
    assert (arrs[3].cpu().numpy() == 0).all(), arrs[3].cpu().numpy()
    for region in range(3):
        for k in range(n_inst):
            got = [int(x) for x in dg[region][k].cpu().numpy().view(np.uint64)]
            assert got == fx["adv_digest"][region], f"engine != pyref fixture: region {region}, instance {k}"
    prog.close()
