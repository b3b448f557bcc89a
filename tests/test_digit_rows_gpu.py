"""The digit-row primitives of the field-chain kernel (engine.hip DigitRow: one 16-lane DPP row per value, one lane per 32-bit
digit) against Python integers, on the patterns the parity tests never produce: a carry that has to ripple through a run of
0xffffffff digits (probability 2^-32 per digit on random data), quotient
estimates on the boundary (exact multiples of w and their neighbours), operands at the ends of the lazy range [0, 2w).
Through the boundary's test hook h2e_selftest_digit_rows (include/h2e.h; one row per case)."""
import ctypes as C
import random

import numpy as np
import pytest

from halo2ecc_s_amd import Program, synth
from halo2ecc_s_amd.engine import lib

pytestmark = pytest.mark.gpu

W = {0: 21888242871839275222246405745257275088696311157297823662689037894645226208583,
     1: 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab}
DIGITS = {0: 8, 1: 12}
M32 = 0xffffffff


def _digits(v, n=16):
    assert 0 <= v < 1 << (32 * n)
    return [(v >> (32 * k)) & M32 for k in range(n)]


def _value(d):
    return sum(int(x) << (32 * k) for k, x in enumerate(d))


@pytest.fixture(scope="module")
def rows(engine):
    """run(fp, op, cases) -> outputs; cases = list of (in0 digits[16], in1 digits[16])"""
    import torch
    for fp in (0, 1):   # the field constants of a pair reach the device with the first run of a program of that pair
        prog = Program.int_mul_batch(fp, 1)
        ins = np.stack([synth.int_mul_batch_inputs(fp, 1)])
        arrs = engine.alloc(prog, 1)
        engine.run(prog, engine.upload_inputs(prog, ins), *arrs)
    torch.cuda.synchronize()

    def run(fp, op, cases):
        fn = lib().h2e_selftest_digit_rows
        a = np.zeros((len(cases), 2, 16), dtype=np.uint32)
        for k, (x, y) in enumerate(cases):
            a[k, 0], a[k, 1] = x, y
        d_in = torch.from_numpy(a.view(np.int32)).cuda()
        d_out = torch.zeros((len(cases), 16), dtype=torch.int32, device="cuda")
        assert fn(fp, op, len(cases), d_in.data_ptr(), d_out.data_ptr(), None) == 0
        torch.cuda.synchronize()
        return d_out.cpu().numpy().view(np.uint32)
    return run


@pytest.mark.parametrize("fp", [0, 1])
def test_normalize_resolves_carries_through_runs_of_ones(rows, fp):
    D = DIGITS[fp]
    rnd = random.Random(11 + fp)
    cases = []
    for trial in range(400):
        lo, hi = [0] * 16, [0] * 16
        for j in range(D):
            lo[j], hi[j] = rnd.getrandbits(32), rnd.getrandbits(rnd.choice([0, 1, 14, 31]))
        if trial % 2 == 0:
            # digits start .. start + length - 1 become 0xffffffff in the first addition (lo_j + hi_(j-1)), the digit right below
            # them overflows: its carry has to ripple through the whole run
            start = rnd.randrange(2, D)
            length = rnd.randrange(1, D - start + 1)
            hi[start - 2] = max(1, hi[start - 2])
            lo[start - 1] = M32
            for j in range(start, start + length):
                lo[j] = (M32 - hi[j - 1]) & M32
        cases.append((lo, hi))
    out = rows(fp, 0, cases)
    for (lo, hi), got in zip(cases, out):
        want = sum((lo[j] + (hi[j] << 32)) << (32 * j) for j in range(D))
        assert _value(got) == want, (lo, hi)


@pytest.mark.parametrize("fp", [0, 1])
def test_mont_mul_on_the_lazy_range(rows, fp):
    D, w = DIGITS[fp], W[fp]
    R = 1 << (32 * D)
    rinv = pow(R, -1, w)
    rnd = random.Random(31 + fp)
    edge = [0, 1, 2, w - 1, w, w + 1, 2 * w - 1, (1 << 32) - 1, 1 << 32, R // 5]
    cases = [(x, y) for x in edge for y in edge if x < 2 * w and y < 2 * w]
    cases += [(rnd.randrange(2 * w), rnd.randrange(2 * w)) for _ in range(600)]
    cases += [((1 << (32 * k)) - 1, rnd.randrange(2 * w)) for k in range(1, D)] + [(rnd.randrange(2 * w), (1 << (32 * k)) - 1) for k in range(1, D)]
    cases = [(x % (2 * w), y % (2 * w)) for x, y in cases]
    out = rows(fp, 2, [(_digits(x), _digits(y)) for x, y in cases])
    for (x, y), got in zip(cases, out):
        r = _value(got)
        assert r < 2 * w and r % w == x * y * rinv % w, (hex(x), hex(y), hex(r))


@pytest.mark.parametrize("fp", [0, 1])
def test_reduce_columns_quotient_boundaries(rows, fp):
    D, w = DIGITS[fp], W[fp]
    rnd = random.Random(47 + fp)
    values = []
    for k in [0, 1, 2, 3, 1000, 16384, 20000, 32767]:      # the kernel's combinations stay below 2^15 w
        for dv in (-2, -1, 0, 1, 2):
            v = k * w + dv
            if 0 <= v < (1 << 15) * w:
                values.append(v)
    values += [rnd.randrange((1 << 15) * w) for _ in range(500)]
    values += [rnd.randrange(1 << 15) * w + rnd.choice([0, 1, w - 1]) for _ in range(200)]
    values += [(v | (((1 << 64) - 1) << (32 * rnd.randrange(1, D - 2)))) % ((1 << 15) * w) for v in values[:200]]   # runs of 0xffffffff digits
    cases = []
    for v in values:
        col = _digits(v)[:D]                               # columns as Python integers; what exceeds D digits sits in the top one
        col[D - 1] += (v >> (32 * D)) << 32
        if rnd.random() < 0.6:                             # the same value as unnormalised columns: move multiples of 2^32 down
            for j in range(D - 1, 0, -1):
                mv = min(col[j], rnd.getrandbits(13))
                col[j] -= mv
                col[j - 1] += mv << 32
        assert sum(c << (32 * j) for j, c in enumerate(col)) == v and all(c < 1 << 47 for c in col)
        lo = [c & M32 for c in col] + [0] * (16 - D)
        hi = [c >> 32 for c in col] + [0] * (16 - D)
        cases.append((lo, hi))
    out = rows(fp, 3, cases)
    for v, got in zip(values, out):
        r = _value(got)
        assert r < 2 * w and r % w == v % w, (hex(v), hex(r))
