"""Shared comparison helpers: engine program / outputs vs the oracle's Records."""
import numpy as np


def expand_fixed(program, region):
    d = program.fixed_dict()
    idx = (program.base_fix, program.range_fix, program.select_fix)[region]()
    vals = d[idx]  # [rows][cols][4]; id 0 -> zeros
    return vals, (idx != 0).astype(np.uint8)


def compare_shape(program, orun, patches_inputs=None):
    """Everything in Records that is not an advice value: offsets, heights, flags, fixed cells, permutations."""
    i = orun.info
    assert i.status == 0, orun.error
    assert (program.base_offset, program.range_offset, program.select_offset) == (i.base_offset, i.range_offset, i.select_offset)
    assert (program.base_height, program.range_height, program.select_height) == (i.base_height, i.range_height, i.select_height)
    assert program.n_advice_cells == i.n_advice_cells
    rows = (program.base_rows, program.range_rows, program.select_rows)
    flags = (program.base_flags(), program.range_flags(), program.select_flags())
    for region in range(3):
        _, oflags = orun.adv(region, rows[region])
        assert np.array_equal(flags[region], oflags), f"assigned/permute flags differ in region {region}"
        ovals, opresent = orun.fix(region, rows[region])
        vals, present = expand_fixed(program, region)
        if region == 0 and program.n_fixed_patches:
            # constants made from instance inputs are reported as patches, not dictionary ids
            vals = vals.copy()
            present = present.copy()
            for row, col, slot, limb in program.fixed_patches():
                assert present[row, col] == 0
                present[row, col] = 1
                vals[row, col] = patch_value(program, patches_inputs, int(slot), np.int32(limb))
        assert np.array_equal(present, opresent), f"fixed presence differs in region {region}"
        assert np.array_equal(vals, ovals), f"fixed values differ in region {region}"
    assert program.n_permutations == i.n_permutations
    assert np.array_equal(program.permutations(), orun.permutations()), "permutation list differs"


BN_R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def patch_value(program, inputs, slot, limb):
    x = 0
    for k, w in enumerate(inputs[slot]):
        x |= int(w) << (64 * k)
    v = x % BN_R if limb < 0 else (x >> (108 * int(limb))) & ((1 << 108) - 1)
    return np.array([(v >> (64 * k)) & ((1 << 64) - 1) for k in range(4)], dtype=np.uint64)


def compare_advice(program, orun, base, rng, sel, instance=0):
    """GPU advice arrays (torch int64 [inst][rows][cols][4]) vs the oracle, cell for cell."""
    outs = (base, rng, sel)
    rows = (program.base_rows, program.range_rows, program.select_rows)
    for region in range(3):
        ovals, oflags = orun.adv(region, rows[region])
        got = outs[region][instance].cpu().numpy().view(np.uint64)
        if not np.array_equal(got, ovals):
            bad = np.argwhere((got != ovals).any(axis=2))
            r, c = bad[0]
            raise AssertionError(
                f"advice differs in region {region}: {len(bad)} cells, first at row {r} col {c}: "
                f"gpu {[hex(int(x)) for x in got[r, c]]} oracle {[hex(int(x)) for x in ovals[r, c]]}")
