"""shape artefacts generated on the device in the prover's layout - fixed columns, the range lookup table,
copy constraints - against the oracle's Records (fixed cells, permutation list) and the table definition of the reference
(RangeChip::init_table, src/circuit/range_chip.rs:230-258)."""
import numpy as np
import pytest

import oracle_lib
from halo2ecc_s_amd import Program, synth
from halo2ecc_s_amd import engine as E

pytestmark = pytest.mark.gpu
BN_R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def _ints(a):
    a = a.reshape(-1, 4)
    return [sum(int(a[i, k]) << (64 * k) for k in range(4)) for i in range(a.shape[0])]


def test_fixed_columns_of_an_msm_tile(engine, oracle):
    """every fixed cell of the three regions, column-major and row-major, canonical, == the oracle's fixed arrays; the
    Montgomery form of the first rows == x * 2^256 mod n"""
    n = 6
    inp, _ = synth.msm_bn256_tile_inputs(n, tile=9)
    prog = Program.msm_bn256_tile(n)
    orun = oracle_lib.run_msm_bn256_tile(n, inp)
    for region in range(3):
        rows = (prog.base_rows, prog.range_rows, prog.select_rows)[region]
        ofix, _ = orun.fix(region, rows)                      # [rows][cols][4], None = 0
        cols = engine.export_fixed(prog, region, layout=E.LAYOUT_COLUMNS)
        rws = engine.export_fixed(prog, region, layout=E.LAYOUT_ROWS)
        mont = engine.export_fixed(prog, region, layout=E.LAYOUT_ROWS, form=E.FORM_MONTGOMERY)
        engine.torch.cuda.synchronize()
        assert np.array_equal(rws[0].cpu().numpy().view(np.uint64), ofix)
        assert np.array_equal(cols[0].cpu().numpy().view(np.uint64), np.ascontiguousarray(ofix.transpose(1, 0, 2)))
        k = min(rows, 200)
        assert _ints(mont[0].cpu().numpy().view(np.uint64)[:k]) == [(v << 256) % BN_R for v in _ints(ofix[:k])]


def test_fixed_columns_with_input_dependent_constants(engine, oracle):
    """the G2 constants of a pairing check are fixed cells made from instance inputs (h2e_shape.fixed_patches): with the run's
    inputs the device-side fixed columns of two instances equal the two oracle runs"""
    ins = [synth.pairing_check_bn256_inputs(instance=820 + k) for k in range(2)]
    prog = Program.pairing_check_bn256()
    d_in = engine.upload_inputs(prog, np.stack(ins))
    fixed = engine.export_fixed(prog, 0, n_instances=2, d_inputs=d_in, layout=E.LAYOUT_ROWS)
    engine.torch.cuda.synchronize()
    for k, inp in enumerate(ins):
        ofix, _ = oracle_lib.run_pairing_check_bn256(inp).fix(0, prog.base_rows)
        assert np.array_equal(fixed[k].cpu().numpy().view(np.uint64), ofix), f"instance {k}"
    with pytest.raises(E.H2EError):
        engine.export_fixed(prog, 0)          # input-dependent constants without the inputs


def test_range_table(engine):
    """(tag, value) for tag in 0..=18, value in 0..2^tag (src/circuit/range_chip.rs:233-251): 2^19 - 1 rows"""
    tab = engine.range_table()
    mont = engine.range_table(form=E.FORM_MONTGOMERY)
    engine.torch.cuda.synchronize()
    t = tab.cpu().numpy().view(np.uint64)
    assert (t[:, :, 1:] == 0).all()
    tags, vals = t[0, :, 0], t[1, :, 0]
    want_t = np.concatenate([np.full(1 << k, k, dtype=np.uint64) for k in range(19)])
    want_v = np.concatenate([np.arange(1 << k, dtype=np.uint64) for k in range(19)])
    assert len(want_t) == 524287 and np.array_equal(tags, want_t) and np.array_equal(vals, want_v)
    m = mont.cpu().numpy().view(np.uint64)
    for r in (0, 1, 2, 700, 524286):
        assert _ints(m[0, r:r + 1]) == [(int(tags[r]) << 256) % BN_R] and _ints(m[1, r:r + 1]) == [(int(vals[r]) << 256) % BN_R]


def test_copy_constraints(engine, oracle):
    """the permutation list as (advice column, row) pairs, columns 0-4 base / 5-7 range / 8-9 select, in the reference's order"""
    n = 5
    inp, _ = synth.msm_bn256_tile_inputs(n, tile=10)
    prog = Program.msm_bn256_tile(n)
    cc = engine.export_copy_constraints(prog)
    engine.torch.cuda.synchronize()
    got = cc.cpu().numpy().astype(np.int64)
    perms = oracle_lib.run_msm_bn256_tile(n, inp).permutations().astype(np.int64)
    base_col = np.array([0, 5, 8])

    def dec(c):
        return base_col[c >> 30] + ((c >> 27) & 7), c & 0x3FFFFFF
    ca, ra = dec(perms[:, 0])
    cb, rb = dec(perms[:, 1])
    assert np.array_equal(got, np.stack([ca, ra, cb, rb], axis=1))
