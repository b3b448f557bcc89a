"""ORACLE (second, independent restatement) - TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A pure-Python restatement of the reference's witness path, written directly from the Rust sources (NOT from the C++
oracle and NOT from the engine's recorder), with Python integers as the field elements.  Purpose: pin the C++ oracle
and the recorder by a different route (the reference has no golden vectors and cannot be built here) -
every structural fact (row offsets, heights incl. quirks Q3/Q4, op counts, permutation list, fixed cells) and every
advice value of the workloads below is produced twice, by two texts that share nothing but the reference.

Each function cites the reference file:line it follows (paths relative to /root/reference/src).
Slow by construction (pure-Python loops): small cases run inside the CPU test-suite, the pairing checks and the larger
MSM tiles are run once by tests/golden/make_pyref_golden.py and their digests / counts committed as fixtures.
"""
import hashlib
from collections import Counter

N_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001   # bn256 Fr (the native field N)
BN_Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
BLS_Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
BLS_R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001

BASE, RANGE, SELECT = 0, 1, 2
VAR_COLUMNS, MUL_COLUMNS = 5, 2                      # circuit/base_chip.rs:14-16
COMMON_RANGE_BITS, MAX_CHUNKS = 18, 3                # circuit/range_chip.rs:22-33
RANGE_VALUE_DECOMPOSE = 6
OVERFLOW_BITS = 6                                    # context.rs:38
MSM_PREFIX_OFFSET = 1 << 20                          # circuit/ecc_chip.rs:20


class UnsafeError(Exception):                        # circuit/ecc_chip.rs:23-34
    pass


class AV:
    """AssignedValue (assign.rs:25-29): a cell (region, col, row) + a copy of its value"""
    __slots__ = ("region", "col", "row", "val")

    def __init__(self, region, col, row, val):
        self.region, self.col, self.row, self.val = region, col, row, val

    @property
    def cell(self):
        return (self.region << 30) | (self.col << 27) | self.row


class Shared:
    """RecordsInner (context.rs:241-252): the six arrays, shared by forked contexts (Arc)"""

    def __init__(self):
        self.adv = ({}, {}, {})          # region -> {row * cols + col: value}
        self.fix = ({}, {}, {})
        self.permute = set()             # cells whose permute flag is set
        self.counts = Counter()
        self.marks = {}


ADV_COLS = (5, 3, 2)
FIX_COLS = (9, 2, 2)


class Context:
    """Context + Records (context.rs:40-46, 294-301) with the L0 writers (context.rs:590-997) and the BaseChipOps
    recipes (circuit/base_chip.rs:81-605)"""

    def __init__(self, shared=None):
        self.s = shared if shared is not None else Shared()
        self.permutations = []
        self.base_offset = self.range_offset = self.select_offset = 0
        self.base_height = self.range_height = self.select_height = 0

    # ---- fork / merge (context.rs:145-158, circuit/native_scalar_ecc_chip.rs:50-90) ----
    def clone_with_offset(self, d):
        c = Context(self.s)                                   # clone_without_permutation: heights are copied
        c.base_height, c.range_height, c.select_height = self.base_height, self.range_height, self.select_height
        c.base_offset = self.base_offset + d[0]
        c.range_offset = self.range_offset + d[1]
        c.select_offset = self.select_offset + d[2]
        return c

    def offset(self):
        return (self.base_offset, self.range_offset, self.select_offset)

    def merge(self, other):
        self.permutations.extend(other.permutations)
        self.base_height = max(self.base_height, other.base_height)
        self.range_height = max(self.select_height, other.range_height)     # sic (quirk Q3): native_scalar_ecc_chip.rs:87
        self.select_height = max(self.select_height, other.select_height)

    def apply_offset_diff(self, d):
        self.base_offset += d[0]
        self.range_offset += d[1]
        self.select_offset += d[2]

    # ---- raw cells ----
    def _enable_permute(self, cell):
        self.s.permute.add(cell)

    # context.rs:634-683
    def rec_one_line(self, offset, pairs, constant, mul_coeffs, nxt):
        assert len(pairs) <= VAR_COLUMNS
        if offset >= self.base_height:
            self.base_height = offset + 1
        adv, fix = self.s.adv[BASE], self.s.fix[BASE]
        for i, (base, coeff) in enumerate(pairs):
            if isinstance(base, AV):
                new_cell = (BASE << 30) | (i << 27) | offset
                self._enable_permute(new_cell)
                self._enable_permute(base.cell)
                self.permutations.append((base.cell, new_cell))
                v = base.val
            else:
                v = base
            adv[offset * 5 + i] = v
            fix[offset * 9 + i] = coeff % N_MOD
        for i, m in enumerate(mul_coeffs):
            fix[offset * 9 + VAR_COLUMNS + i] = m % N_MOD
        if nxt is not None:
            fix[offset * 9 + VAR_COLUMNS + MUL_COLUMNS] = nxt % N_MOD
        else:
            assert offset * 9 + VAR_COLUMNS + MUL_COLUMNS not in fix
        if constant is not None:
            fix[offset * 9 + VAR_COLUMNS + MUL_COLUMNS + 1] = constant % N_MOD
        else:
            assert offset * 9 + VAR_COLUMNS + MUL_COLUMNS + 1 not in fix

    # context.rs:685-714
    def rec_one_line_with_last(self, offset, pairs, tail, constant, mul_coeffs, nxt):
        assert len(pairs) <= VAR_COLUMNS - 1
        self.rec_one_line(offset, pairs, constant, mul_coeffs, nxt)
        base, coeff = tail
        i = VAR_COLUMNS - 1
        if isinstance(base, AV):
            new_cell = (BASE << 30) | (i << 27) | offset
            self._enable_permute(new_cell)
            self._enable_permute(base.cell)
            self.permutations.append((base.cell, new_cell))
            v = base.val
        else:
            v = base
        self.s.adv[BASE][offset * 5 + i] = v
        self.s.fix[BASE][offset * 9 + i] = coeff % N_MOD

    def _ensure_range(self, offset):                 # context.rs:716-720 (quirk Q4: called with offset + lines)
        if offset >= self.range_height:
            self.range_height = offset + 1

    # context.rs:835-857
    def assign_one_line_range_value(self, offset, v, v_acc, bits):
        assert bits <= COMMON_RANGE_BITS
        self._ensure_range(offset + 1)
        adv, fix = self.s.adv[RANGE], self.s.fix[RANGE]
        fix[offset * 2 + 0] = 1
        fix[offset * 2 + 1] = bits
        adv[offset * 3 + 1] = v[0]
        adv[offset * 3 + 0] = v_acc
        return AV(RANGE, 0, offset, v_acc)

    # context.rs:859-907
    def assign_two_line_range_value(self, offset, v, v_acc, bits):
        assert 2 * COMMON_RANGE_BITS <= bits <= 4 * COMMON_RANGE_BITS
        self._ensure_range(offset + 2)
        adv, fix = self.s.adv[RANGE], self.s.fix[RANGE]
        fix[offset * 2 + 0] = 2
        adv[offset * 3 + 2] = v[0]
        adv[(offset + 1) * 3 + 2] = v[1]
        fix[offset * 2 + 1] = COMMON_RANGE_BITS if bits >= 3 * COMMON_RANGE_BITS else bits % COMMON_RANGE_BITS
        adv[offset * 3 + 1] = v[2]
        fix[(offset + 1) * 2 + 1] = bits - 3 * COMMON_RANGE_BITS if bits > 3 * COMMON_RANGE_BITS else 0
        adv[(offset + 1) * 3 + 1] = v[3]
        adv[offset * 3 + 0] = v_acc
        return AV(RANGE, 0, offset, v_acc)

    # context.rs:909-972
    def assign_three_line_range_value(self, offset, v, v_acc, bits):
        assert 3 * COMMON_RANGE_BITS <= bits <= 6 * COMMON_RANGE_BITS
        self._ensure_range(offset + 3)
        adv, fix = self.s.adv[RANGE], self.s.fix[RANGE]
        fix[offset * 2 + 0] = 3
        adv[offset * 3 + 2] = v[0]
        adv[(offset + 1) * 3 + 2] = v[1]
        adv[(offset + 2) * 3 + 2] = v[2]
        fix[offset * 2 + 1] = COMMON_RANGE_BITS if bits >= 4 * COMMON_RANGE_BITS else bits % COMMON_RANGE_BITS
        adv[offset * 3 + 1] = v[3]
        if bits >= 5 * COMMON_RANGE_BITS:
            t = COMMON_RANGE_BITS
        elif bits > 4 * COMMON_RANGE_BITS:
            t = bits % COMMON_RANGE_BITS
        else:
            t = 0
        fix[(offset + 1) * 2 + 1] = t
        adv[(offset + 1) * 3 + 1] = v[4]
        fix[(offset + 2) * 2 + 1] = bits - 5 * COMMON_RANGE_BITS if bits > 5 * COMMON_RANGE_BITS else 0
        adv[(offset + 2) * 3 + 1] = v[5]
        adv[offset * 3 + 0] = v_acc
        return AV(RANGE, 0, offset, v_acc)

    # context.rs:974-997
    def assign_range_value(self, offset, v, v_acc, bits):
        if bits <= COMMON_RANGE_BITS:
            return self.assign_one_line_range_value(offset, v, v_acc, bits), 1
        assert bits >= 2 * COMMON_RANGE_BITS
        if bits <= 4 * COMMON_RANGE_BITS:
            return self.assign_two_line_range_value(offset, (list(v) + [0] * 4)[:4], v_acc, bits), 2
        assert bits <= 6 * COMMON_RANGE_BITS
        return self.assign_three_line_range_value(offset, (list(v) + [0] * 6)[:6], v_acc, bits), 3

    # context.rs:749-766
    def rec_assign_cache_value(self, offset, v, encode):
        if offset >= self.select_height:
            self.select_height = offset + 1
        self.s.adv[SELECT][offset * 2 + 0] = v.val
        idx = (SELECT << 30) | (0 << 27) | offset
        self.permutations.append((idx, v.cell))
        self._enable_permute(idx)
        self._enable_permute(v.cell)
        self.s.fix[SELECT][offset * 2 + 0] = encode
        self.s.fix[SELECT][offset * 2 + 1] = 0

    # context.rs:768-801
    def rec_assign_select_value(self, offset, v, encode, selector):
        if offset >= self.select_height:
            self.select_height = offset + 1
        self.s.adv[SELECT][offset * 2 + 0] = v.val
        self.s.adv[SELECT][offset * 2 + 1] = selector.val
        sel_cell = (SELECT << 30) | (1 << 27) | offset
        self.permutations.append((sel_cell, selector.cell))
        self._enable_permute(sel_cell)
        self._enable_permute(selector.cell)
        self.s.fix[SELECT][offset * 2 + 0] = encode
        self.s.fix[SELECT][offset * 2 + 1] = 1
        return AV(SELECT, 0, offset, v.val)

    # ---- BaseChipOps on Context (circuit/base_chip.rs:503-605) ----
    @staticmethod
    def _val(x):
        return x.val if isinstance(x, AV) else x

    def one_line(self, pairs, constant=None, mul=(), nxt=None):          # :516-539
        o = self.base_offset
        res = [AV(BASE, i, o, self._val(b)) for i, (b, _) in enumerate(pairs)]
        self.rec_one_line(o, pairs, constant, mul, nxt)
        self.base_offset += 1
        return res

    def one_line_with_last(self, pairs, last, constant=None, mul=(), nxt=None):   # :541-572
        o = self.base_offset
        res0 = [AV(BASE, i, o, self._val(b)) for i, (b, _) in enumerate(pairs)]
        res1 = AV(BASE, VAR_COLUMNS - 1, o, self._val(last[0]))
        self.rec_one_line_with_last(o, pairs, last, constant, mul, nxt)
        self.base_offset += 1
        return res0, res1

    def sum_with_constant_in_one_line(self, elems, constant):             # :110-132
        assert len(elems) < VAR_COLUMNS
        s = sum(x.val * y for x, y in elems)
        if constant is not None:
            s += constant
        s %= N_MOD
        return self.one_line_with_last([(x, y) for x, y in elems], (s, -1), constant)[1]

    def sum_with_constant(self, elems, constant):                          # :134-153
        if len(elems) < VAR_COLUMNS:
            return self.sum_with_constant_in_one_line(elems, constant)
        curr, tail = elems[:VAR_COLUMNS - 1], elems[VAR_COLUMNS - 1:]
        acc = self.sum_with_constant_in_one_line(list(curr), constant)
        for k in range(0, len(tail), VAR_COLUMNS - 2):
            acc = self.sum_with_constant_in_one_line(list(tail[k:k + VAR_COLUMNS - 2]) + [(acc, 1)], None)
        return acc

    def add(self, a, b):                                                   # :155-160
        return self.sum_with_constant([(a, 1), (b, 1)], None)

    def add_constant(self, a, c):                                          # :162-167
        return self.sum_with_constant([(a, 1)], c % N_MOD)

    def mul(self, a, b):                                                   # :176-193
        return self.one_line_with_last([(a, 0), (b, 0)], (a.val * b.val % N_MOD, -1), None, (1,))[1]

    def mul_add(self, a, b, ab_coeff, c, c_coeff):                         # :219-243
        d = (a.val * b.val * ab_coeff + c.val * c_coeff) % N_MOD
        return self.one_line_with_last([(a, 0), (b, 0), (c, c_coeff)], (d, -1), None, (ab_coeff,))[1]

    def mul_add_with_next_line(self, ls):                                  # :245-281
        assert ls
        if len(ls) == 1:
            a, b, c, cc = ls[0]
            return self.mul_add(a, b, 1, c, cc)
        t = 0
        for i, (a, b, c, cc) in enumerate(ls):
            self.one_line_with_last([(a, 0), (b, 0), (c, cc)], (t, 0 if i == 0 else 1), None, (1,), -1)
            t = (a.val * b.val + c.val * cc + t) % N_MOD
        return self.one_line_with_last([], (t, 0))[1]

    def invert(self, a):                                                   # :298-321
        b = pow(a.val, -1, N_MOD) if a.val % N_MOD else 0
        c = (1 - a.val * b) % N_MOD
        cells = self.one_line([(a, 0), (c, 0)], None, (1,))
        c_cell = cells[1]
        res0, res1 = self.one_line_with_last([(a, 0), (b, 0)], (c_cell, 1), -1, (1,))
        return res1, res0[1]

    def is_zero(self, a):                                                  # :323-325
        return self.invert(a)[0]

    def assign_constant(self, v):                                          # :344-349
        v %= N_MOD
        return self.one_line([(v, -1)], v)[0]

    def assign(self, v):                                                   # :351-355
        return self.one_line([(v % N_MOD, 0)])[0]

    def assign_bit(self, a):                                               # :357-367 (quirk Q2: two copies)
        return self.one_line([(a, 1), (a, 0)], None, (-1,))[0]

    def assert_constant(self, a, b):                                       # :375-379
        if a.val != b % N_MOD:
            raise AssertionError("assert_constant")
        self.one_line([(a, -1)], b % N_MOD)

    def assert_bit(self, a):                                               # :381-390
        self.one_line([(a, 1), (a, 0)], None, (-1,))

    def and_(self, a, b):                                                  # :392-396
        return self.mul(a, b)

    def not_(self, a):                                                     # :398-403
        return self.sum_with_constant([(a, -1)], 1)

    def or_(self, a, b):                                                   # :428-439
        c = (a.val + b.val - a.val * b.val) % N_MOD
        return self.one_line_with_last([(a, 1), (b, 1)], (c, -1), None, (-1,))[1]

    def xnor(self, a, b):                                                  # :455-467
        c = (1 - a.val - b.val + 2 * a.val * b.val) % N_MOD
        return self.one_line_with_last([(a, -1), (b, -1)], (c, -1), 1, (2,))[1]

    def bisec(self, cond, a, b):                                           # :574-604 (VAR_COLUMNS >= 5 branch)
        c = (cond.val * a.val + (1 - cond.val) * b.val) % N_MOD
        return self.one_line_with_last([(cond, 0), (a, 0), (cond, 0), (b, 1)], (c, -1), None, (1, -1))[1]

    def assert_true(self, a):                                              # :487-490
        if a.val != 1:
            raise AssertionError("assert_true")
        self.assert_constant(a, 1)

    def assert_false(self, a):                                             # :492-495
        if a.val != 0:
            raise AssertionError("assert_false")
        self.assert_constant(a, 0)

    def try_assert_false(self, a):                                         # :497-500 (quirk Q8: the row is written either way)
        self.one_line([(a, -1)], 0)
        return a.val == 0


class RangeInfo:
    """range_info.rs:77-184 (only what the witness path reads)"""

    def __init__(self, w_modulus):
        self.w_modulus = w_modulus
        self.n_modulus = N_MOD
        w_max = w_modulus - 1
        self.w_ceil_bits = w_max.bit_length()
        self.n_floor_bits = (N_MOD - 1).bit_length() - 1
        self.limb_bits = COMMON_RANGE_BITS * RANGE_VALUE_DECOMPOSE
        self.limbs = (self.w_ceil_bits + self.limb_bits - 1) // self.limb_bits
        self.overflow_bits = OVERFLOW_BITS
        self.overflow_limit = 1 << OVERFLOW_BITS
        self.d_bits = self.w_ceil_bits + OVERFLOW_BITS * 2 + 1           # :299-314
        self.w_ceil_leading_bits, self.w_ceil_leading_decompose = self._lead(self.w_ceil_bits)
        self.d_leading_bits, self.d_leading_decompose = self._lead(self.d_bits)
        self.limb_mask = (1 << self.limb_bits) - 1
        self.limb_modulus = 1 << self.limb_bits
        self.common_range_mask = (1 << COMMON_RANGE_BITS) - 1
        self.w_ceil = 1 << self.w_ceil_bits
        self.w_native = w_modulus % N_MOD
        self.w_modulus_limbs_le = [(w_modulus >> (i * self.limb_bits)) & self.limb_mask for i in range(self.limbs)]
        self.limb_coeffs = [(1 << (i * self.limb_bits)) % N_MOD for i in range(self.limbs)]
        lb = self.limb_bits
        self.pure_w_check_limbs = (self.w_ceil_bits - self.n_floor_bits + lb - 1) // lb
        self.mul_check_limbs = (max(self.w_ceil_bits * 2 + OVERFLOW_BITS * 2, self.d_bits + self.w_ceil_bits) - self.n_floor_bits + lb - 1) // lb
        self.reduce_check_limbs = (max(self.w_ceil_bits + OVERFLOW_BITS, COMMON_RANGE_BITS + self.w_ceil_bits) - self.n_floor_bits + lb - 1) // lb
        self.w_modulus_of_ceil_times = [None] + [self._ceil_times(t) for t in range(1, self.overflow_limit)]

    @staticmethod
    def _lead(bits):                                                      # :57-75
        common_limb_bits = RANGE_VALUE_DECOMPOSE * COMMON_RANGE_BITS
        leading = common_limb_bits if bits % common_limb_bits == 0 else bits % common_limb_bits
        assert 2 * COMMON_RANGE_BITS <= leading <= common_limb_bits
        chunk = leading % COMMON_RANGE_BITS
        return (COMMON_RANGE_BITS, leading // COMMON_RANGE_BITS) if chunk == 0 else (chunk, leading // COMMON_RANGE_BITS + 1)

    def _ceil_times(self, times):                                         # :334-359
        mx = self.w_ceil * times
        n, rem = divmod(mx, self.w_modulus)
        if rem > 0:
            n += 1
        upper = self.w_modulus * n
        limbs = []
        for _ in range(self.limbs - 1):
            rem = (upper & self.limb_mask) + self.limb_modulus * times
            upper = (upper - rem) >> self.limb_bits
            limbs.append(rem % N_MOD)
        limbs.append(upper % N_MOD)
        return limbs

    def bn_to_limb_le(self, w):
        return [(w >> (i * self.limb_bits)) & self.limb_mask for i in range(self.limbs)]


class AInt:
    """AssignedInteger (assign.rs:31-37)"""
    __slots__ = ("limbs_le", "native", "times")

    def __init__(self, limbs_le, native, times):
        self.limbs_le, self.native, self.times = limbs_le, native, times


_INFO_CACHE = {}


class IntegerContext:
    """IntegerContext<W, N> (context.rs:161-188): RangeChipOps (circuit/range_chip.rs:262-348), SelectChipOps
    (circuit/select_chip.rs:99-162) and IntegerChipOps (circuit/integer_chip.rs:15-686)"""

    def __init__(self, ctx, w_modulus):
        self.ctx = ctx
        if w_modulus not in _INFO_CACHE:
            _INFO_CACHE[w_modulus] = RangeInfo(w_modulus)
        self.info = _INFO_CACHE[w_modulus]

    def fork(self, ctx):
        c = IntegerContext.__new__(IntegerContext)
        c.ctx, c.info = ctx, self.info
        return c

    # ---- range chip ----
    def _decompose(self, bn, n):                                          # range_chip.rs:270-280
        return bn % N_MOD, [(bn >> (i * COMMON_RANGE_BITS)) & self.info.common_range_mask for i in range(n)]

    def assign_common(self, bn):                                          # :287-298
        v = bn % N_MOD
        res = self.ctx.assign_one_line_range_value(self.ctx.range_offset, [v], v, COMMON_RANGE_BITS)
        self.ctx.range_offset += 1
        return res

    def _assign_range(self, bn, n_decompose, bits):
        v_acc, v = self._decompose(bn, n_decompose)
        res, inc = self.ctx.assign_range_value(self.ctx.range_offset, v, v_acc, bits)
        self.ctx.range_offset += inc
        return res

    def assign_nonleading_limb(self, bn):                                 # :300-315
        return self._assign_range(bn, MAX_CHUNKS * 2, self.info.limb_bits)

    def assign_w_ceil_leading_limb(self, bn):                             # :317-333
        return self._assign_range(bn, self.info.w_ceil_leading_decompose, self.info.w_ceil_bits % self.info.limb_bits)

    def assign_d_leading_limb(self, bn):                                  # :335-347
        return self._assign_range(bn, self.info.d_leading_decompose, self.info.d_bits % self.info.limb_bits)

    # ---- select chip ----
    @staticmethod
    def _encode_offset(g, offset, limb_offset):                           # select_chip.rs:118-122
        return ((offset << 128) + (g << 64) + limb_offset) % N_MOD

    def assign_cache_value(self, v, offset, group_index, selector):       # :129-143
        self.ctx.rec_assign_cache_value(self.ctx.select_offset, v, self._encode_offset(group_index, selector, offset))
        self.ctx.select_offset += 1

    def assign_selected_value(self, v, offset, group_index, selector):    # :144-161
        r = self.ctx.rec_assign_select_value(self.ctx.select_offset, v, self._encode_offset(group_index, 0, offset), selector)
        self.ctx.select_offset += 1
        return r

    # ---- integer chip ----
    def get_w_bn(self, a):                                                # integer_chip.rs:217-224
        res = 0
        for i in reversed(range(self.info.limbs)):
            res = (res << self.info.limb_bits) + a.limbs_le[i].val
        return res

    def assign_w(self, w):                                                # :236-258
        info = self.info
        self.ctx.s.counts["assign_w"] += 1
        limbs = [self.assign_nonleading_limb((w >> (i * info.limb_bits)) & info.limb_mask) for i in range(info.limbs - 1)]
        limbs.append(self.assign_w_ceil_leading_limb((w >> ((info.limbs - 1) * info.limb_bits)) & info.limb_mask))
        native = self.ctx.sum_with_constant(list(zip(limbs, info.limb_coeffs)), None)
        return AInt(limbs, native, 1)

    def assign_d(self, d):                                                # :260-281
        info = self.info
        limbs = [self.assign_nonleading_limb((d >> (i * info.limb_bits)) & info.limb_mask) for i in range(info.limbs - 1)]
        limbs.append(self.assign_d_leading_limb((d >> ((info.limbs - 1) * info.limb_bits)) & info.limb_mask))
        native = self.ctx.sum_with_constant(list(zip(limbs, info.limb_coeffs)), None)
        return limbs, native

    def reduce(self, a):                                                  # :283-373
        if a.times == 1:
            return a
        info = self.info
        self.ctx.s.counts["reduce"] += 1
        assert a.times < info.overflow_limit
        a_bn = self.get_w_bn(a)
        d, rem = divmod(a_bn, info.w_modulus)
        assigned_rem = self.assign_w(rem)
        assigned_d = self.assign_common(d)
        self.ctx.one_line_with_last([(assigned_d, info.w_native), (assigned_rem.native, 1)], (a.native, -1))
        last_v = None
        rem_limbs = info.bn_to_limb_le(rem)
        for i in range(info.reduce_check_limbs):
            last_borrow = info.overflow_limit if i != 0 else 0
            carry = last_v.val if last_v is not None else 0
            u = d * info.w_modulus_limbs_le[i] + rem_limbs[i] + info.limb_modulus * info.overflow_limit - a.limbs_le[i].val + carry - last_borrow
            v, v_rem = divmod(u, info.limb_modulus)
            assert v_rem == 0
            v = self.assign_nonleading_limb(v)
            self.ctx.one_line_with_last(
                [(assigned_d, info.w_modulus_limbs_le[i]), (assigned_rem.limbs_le[i], 1), (a.limbs_le[i], -1),
                 (last_v, 1) if last_v is not None else (0, 0)],                      # quirk Q5: pair!(zero, zero)
                (v, -info.limb_modulus), info.limb_modulus * info.overflow_limit - (0 if i == 0 else info.overflow_limit))
            last_v = v
        return assigned_rem

    def conditionally_reduce(self, a):                                    # :375-382
        return self.reduce(a) if a.times > (1 << (self.info.overflow_bits - 2)) else a

    def _native_of(self, limbs):
        return self.ctx.sum_with_constant(list(zip(limbs, self.info.limb_coeffs)), None)

    def int_add(self, a, b):                                              # :384-406
        self.ctx.s.counts["int_add"] += 1
        limbs = [self.ctx.add(a.limbs_le[i], b.limbs_le[i]) for i in range(self.info.limbs)]
        return self.conditionally_reduce(AInt(limbs, self._native_of(limbs), a.times + b.times))

    def int_sub(self, a, b):                                              # :408-437
        self.ctx.s.counts["int_sub"] += 1
        upper = self.info.w_modulus_of_ceil_times[b.times]
        limbs = [self.ctx.sum_with_constant([(a.limbs_le[i], 1), (b.limbs_le[i], -1)], upper[i]) for i in range(self.info.limbs)]
        return self.conditionally_reduce(AInt(limbs, self._native_of(limbs), a.times + b.times + 1))

    def int_neg(self, a):                                                 # :439-464
        self.ctx.s.counts["int_neg"] += 1
        upper = self.info.w_modulus_of_ceil_times[a.times]
        limbs = [self.ctx.sum_with_constant([(a.limbs_le[i], -1)], upper[i]) for i in range(self.info.limbs)]
        return self.conditionally_reduce(AInt(limbs, self._native_of(limbs), a.times + 1))

    def _mul_equation_on_limbs(self, a, b, d, rem):                       # :73-193
        info, ctx = self.info, self.ctx
        assert a.times < info.overflow_limit and b.times < info.overflow_limit and rem.times == 1
        limbs = []
        for pos in range(info.mul_check_limbs):
            r_bound = min(pos + 1, info.limbs)
            l_bound = max(pos - (info.limbs - 1), 0)
            limbs.append(ctx.mul_add_with_next_line(
                [(a.limbs_le[i], b.limbs_le[pos - i], d[i], -info.w_modulus_limbs_le[pos - i]) for i in range(l_bound, r_bound)]))
        lm = info.limb_modulus % N_MOD
        borrow = (info.limbs * lm + 2) % N_MOD
        u = ctx.sum_with_constant([(limbs[0], 1), (rem.limbs_le[0], -1)], lm * borrow % N_MOD)
        v, r = divmod(u.val, info.limb_modulus)
        assert r == 0
        v_h_bn, v_l_bn = divmod(v, info.limb_modulus)
        v_h = self.assign_common(v_h_bn)
        v_l = self.assign_nonleading_limb(v_l_bn)
        ctx.one_line_with_last([(v_h, info.limb_coeffs[2]), (v_l, info.limb_coeffs[1])], (u, -1))
        for i in range(1, info.mul_check_limbs):
            if i < info.limbs:
                elems = [(limbs[i], 1), (rem.limbs_le[i], -1), (v_h, info.limb_coeffs[1]), (v_l, info.limb_coeffs[0])]
            else:                                                                    # only bls12_381 Fq (:161-192)
                elems = [(limbs[i], 1), (v_h, info.limb_coeffs[1]), (v_l, info.limb_coeffs[0])]
            u = ctx.sum_with_constant(elems, (lm * borrow - borrow) % N_MOD)
            v, r = divmod(u.val, info.limb_modulus)
            assert r == 0
            v_h_bn, v_l_bn = divmod(v, info.limb_modulus)
            v_h = self.assign_common(v_h_bn)
            v_l = self.assign_nonleading_limb(v_l_bn)
            ctx.one_line_with_last([(v_h, info.limb_coeffs[2]), (v_l, info.limb_coeffs[1])], (u, -1))

    def _mul_equation_on_native(self, a, b, d_native, rem):               # :195-215
        self.ctx.one_line([(a.native, 0), (b.native, 0), (d_native, self.info.w_native), (rem.native, 1)], None, (-1,))

    def int_mul(self, a, b):                                              # :466-483
        self.ctx.s.counts["int_mul"] += 1
        d, rem = divmod(self.get_w_bn(a) * self.get_w_bn(b), self.info.w_modulus)
        rem = self.assign_w(rem)
        d = self.assign_d(d)
        self._mul_equation_on_limbs(a, b, d[0], rem)
        self._mul_equation_on_native(a, b, d[1], rem)
        return rem

    def int_square(self, a):                                              # :614-616
        return self.int_mul(a, a)

    def int_unsafe_invert(self, x):                                       # :485-491
        one = self.assign_int_constant(1)
        c, v = self.int_div(one, x)
        self.ctx.assert_false(c)
        return v

    def int_div(self, a, b):                                              # :493-538
        info, ctx = self.info, self.ctx
        ctx.s.counts["int_div"] += 1
        b = self.reduce(b)
        is_b_zero = self.is_int_zero(b)
        a_coeff = ctx.not_(is_b_zero)
        a = self.reduce(a)
        limbs_le = [ctx.mul(a.limbs_le[i], a_coeff) for i in range(info.limbs)]
        native = ctx.mul(a.native, a_coeff)
        a = AInt(limbs_le, native, a.times)
        a_bn, b_bn = self.get_w_bn(a), self.get_w_bn(b)
        bw = b_bn % info.w_modulus
        c_bn = (a_bn % info.w_modulus) * pow(bw, -1, info.w_modulus) % info.w_modulus if bw else 0
        d_bn = (b_bn * c_bn - a_bn) // info.w_modulus
        c = self.assign_w(c_bn)
        d = self.assign_d(d_bn)
        self._mul_equation_on_limbs(b, c, d[0], a)
        self._mul_equation_on_native(b, c, d[1], a)
        return is_b_zero, c

    def is_pure_zero(self, a):                                            # :540-548
        s = self.ctx.sum_with_constant([(v, 1) for v in a.limbs_le], None)
        return self.ctx.is_zero(s)

    def is_pure_w_modulus(self, a):                                       # :550-570
        assert a.times == 1
        info, ctx = self.info, self.ctx
        is_eq = ctx.is_zero(ctx.add_constant(a.native, -info.w_native))
        for i in range(info.pure_w_check_limbs):
            is_limb_eq = ctx.is_zero(ctx.add_constant(a.limbs_le[i], -info.w_modulus_limbs_le[i]))
            is_eq = ctx.and_(is_eq, is_limb_eq)
        return is_eq

    def is_int_zero(self, a):                                             # :572-578
        a = self.reduce(a)
        return self.ctx.or_(self.is_pure_zero(a), self.is_pure_w_modulus(a))

    def is_int_equal(self, a, b):                                         # :47-54
        return self.is_int_zero(self.int_sub(a, b))

    def assign_int_constant(self, w):                                     # :580-598 (quirk Q6: no caching)
        self.ctx.s.counts["assign_int_constant"] += 1
        limbs = [self.ctx.assign_constant(v) for v in self.info.bn_to_limb_le(w)]
        return AInt(limbs, self.ctx.assign_constant(w % N_MOD), 1)

    def assert_int_equal(self, a, b):                                     # :600-612
        diff = self.reduce(self.int_sub(a, b))
        s = self.ctx.sum_with_constant([(v, 1) for v in diff.limbs_le], None)
        self.ctx.assert_constant(s, 0)

    def int_mul_small_constant(self, a, b):                               # :618-658
        info = self.info
        assert b < (1 << (info.overflow_bits - 2))
        if a.times * b >= info.overflow_limit:
            a = self.reduce(a)
        limbs = [self.ctx.sum_with_constant([(a.limbs_le[i], b)], None) for i in range(info.limbs)]
        return self.conditionally_reduce(AInt(limbs, self._native_of(limbs), a.times * b))

    def bisec_int(self, cond, a, b):                                      # :660-681
        limbs = [self.ctx.bisec(cond, a.limbs_le[i], b.limbs_le[i]) for i in range(self.info.limbs)]
        return AInt(limbs, self.ctx.bisec(cond, a.native, b.native), max(a.times, b.times))


# ===================================================================================================
# L3: ECC / MSM (circuit/ecc_chip.rs, circuit/native_scalar_ecc_chip.rs)
class Pt:
    __slots__ = ("x", "y", "z")

    def __init__(self, x, y, z=None):
        self.x, self.y, self.z = x, y, z


class NativeScalarEccContext:
    """NativeScalarEccContext<C> (context.rs:190-207): points are (x, y) int pairs or None (identity); `b` = curve b"""

    def __init__(self, ic, curve_b, generator, scalar_bits=254, msm_prefix=0):
        self.ic, self.curve_b, self.generator, self.scalar_bits, self.prefix = ic, curve_b, generator, scalar_bits, msm_prefix

    @property
    def ctx(self):
        return self.ic.ctx

    def has_select_chip(self):
        return self.prefix is not None

    # ParallelClone (native_scalar_ecc_chip.rs:50-90)
    def clone_with_offset(self, d):
        return NativeScalarEccContext(self.ic.fork(self.ctx.clone_with_offset(d)), self.curve_b, self.generator, self.scalar_bits, self.prefix)

    def get_and_increase_msm_prefix(self):                                # :173-178
        r = self.prefix
        self.prefix += MSM_PREFIX_OFFSET
        return r

    # ---- EccChipBaseOps ----
    def assign_point(self, p):                                            # ecc_chip.rs:458-487
        ic, ctx = self.ic, self.ctx
        x, y = p if p is not None else (0, 0)
        z = 1 if p is None else 0
        ax, ay = ic.assign_w(x), ic.assign_w(y)
        az = ctx.assign_bit(z)
        b = ic.assign_int_constant(self.curve_b)
        y2 = ic.int_square(ay)
        x2 = ic.int_square(ax)
        x3 = ic.int_mul(x2, ax)
        right = ic.int_add(x3, b)
        eq = ic.is_int_equal(y2, right)
        ctx.assert_true(ctx.or_(eq, az))
        return Pt(ax, ay, az)

    def assign_non_zero_point(self, p):                                   # :489-512
        ic = self.ic
        assert p is not None
        ax, ay = ic.assign_w(p[0]), ic.assign_w(p[1])
        b = ic.assign_int_constant(self.curve_b)
        y2 = ic.int_square(ay)
        x2 = ic.int_square(ax)
        x3 = ic.int_mul(x2, ax)
        right = ic.int_add(x3, b)
        ic.assert_int_equal(y2, right)
        return Pt(ax, ay)

    def bisec_point(self, cond, a, b):                                    # :531-545
        return Pt(self.ic.bisec_int(cond, a.x, b.x), self.ic.bisec_int(cond, a.y, b.y), self.ctx.bisec(cond, a.z, b.z))

    def bisec_curvature(self, cond, a, b):                                # :547-560
        return (self.ic.bisec_int(cond, a[0], b[0]), self.ctx.bisec(cond, a[1], b[1]))

    def lambda_to_point(self, lam, a, b):                                 # :580-604
        ic = self.ic
        l = lam[0]
        t = ic.int_sub(ic.int_square(l), a.x)
        cx = ic.int_sub(t, b.x)
        t = ic.int_sub(a.x, cx)
        t = ic.int_mul(t, l)
        cy = ic.int_sub(t, a.y)
        return Pt(cx, cy, lam[1])

    def ecc_add(self, a, a_curv, b):                                      # :606-628  (a: point with curvature)
        ic, ctx = self.ic, self.ctx
        diff_x = ic.int_sub(a.x, b.x)
        diff_y = ic.int_sub(a.y, b.y)
        x_eq, tangent = ic.int_div(diff_y, diff_x)
        y_eq = ic.is_int_zero(diff_y)
        eq = ctx.and_(x_eq, y_eq)
        lam = self.bisec_curvature(eq, a_curv, (tangent, x_eq))
        p = self.lambda_to_point(lam, a, b)
        p = self.bisec_point(a.z, b, p)
        return self.bisec_point(b.z, a, p)

    def ecc_assert_equal(self, a, b):                                     # :644-658
        ic, ctx = self.ic, self.ctx
        eq_x = ic.is_int_equal(a.x, b.x)
        eq_y = ic.is_int_equal(a.y, b.y)
        eq_z = ctx.xnor(a.z, b.z)
        eq_xy = ctx.and_(eq_x, eq_y)
        eq_xyz = ctx.and_(eq_xy, eq_z)
        both = ctx.and_(a.z, b.z)
        ctx.assert_true(ctx.or_(eq_xyz, both))

    def assign_constant_point(self, p):                                   # :441-456
        x, y = p if p is not None else (0, 0)
        ax, ay = self.ic.assign_int_constant(x), self.ic.assign_int_constant(y)
        return Pt(ax, ay, self.ctx.assign_constant(1 if p is None else 0))

    def assign_identity(self):                                            # :514-529
        zero = self.ic.assign_int_constant(0)
        one = self.ctx.assign_constant(1)
        return Pt(zero, zero, one), (zero, one)

    def bisec_point_with_curvature(self, cond, a, a_curv, b, b_curv):     # :562-578
        x = self.ic.bisec_int(cond, a.x, b.x)
        y = self.ic.bisec_int(cond, a.y, b.y)
        z = self.ctx.bisec(cond, a.z, b.z)
        return Pt(x, y, z), self.bisec_curvature(cond, a_curv, b_curv)

    def ecc_double(self, a, a_curv):                                      # :630-642
        p = self.lambda_to_point(a_curv, a, a)
        p.z = self.ctx.bisec(a.z, a.z, p.z)
        return p

    def ecc_neg(self, a):                                                 # :660-666
        return Pt(a.x, self.ic.int_neg(a.y), a.z)

    def ecc_reduce(self, a):                                              # :668-675
        x, y = self.ic.reduce(a.x), self.ic.reduce(a.y)
        identity, _ = self.assign_identity()
        return self.bisec_point(a.z, identity, Pt(x, y, a.z))

    def ecc_reduce_with_curvature(self, a):                               # :677-693
        a = self.ecc_reduce(a)
        x_square = self.ic.int_square(a.x)
        num = self.ic.int_mul_small_constant(x_square, 3)
        den = self.ic.int_mul_small_constant(a.y, 2)
        z, v = self.ic.int_div(num, den)
        return a, (self.ic.reduce(v), z)

    def ecc_encode(self, p):                                              # :710-732
        p = self.ecc_reduce(p)
        shift = (1 << self.ic.info.limb_bits) % N_MOD
        c = self.ctx
        return [c.sum_with_constant([(p.x.limbs_le[0], 1), (p.x.limbs_le[1], shift)], None),
                c.sum_with_constant([(p.x.limbs_le[2], 1), (p.y.limbs_le[0], shift)], None),
                c.sum_with_constant([(p.y.limbs_le[1], 1), (p.y.limbs_le[2], shift)], None)]

    def assign_cache_point(self, p, curv, g, sc):                         # :779-788
        i = self.assign_cache_integer(p.x, sc, g, 0)
        i = self.assign_cache_integer(p.y, sc, g, i)
        self.ic.assign_cache_value(p.z, i, g, sc)
        i = self.assign_cache_integer(curv[0], sc, g, i + 1)
        self.ic.assign_cache_value(curv[1], i, g, sc)

    def assign_selected_point(self, p, curv, sc, g):                      # :790-812
        x, i = self.assign_selected_integer(p.x, sc, g, 0)
        y, i = self.assign_selected_integer(p.y, sc, g, i)
        z = self.ic.assign_selected_value(p.z, i, g, sc)
        cv, i = self.assign_selected_integer(curv[0], sc, g, i + 1)
        cz = self.ic.assign_selected_value(curv[1], i, g, sc)
        return Pt(x, y, z), (cv, cz)

    def to_point_with_curvature(self, a):                                 # :695-708
        ic = self.ic
        x_square = ic.int_square(a.x)
        num = ic.int_mul_small_constant(x_square, 3)
        den = ic.int_mul_small_constant(a.y, 2)
        z, v = ic.int_div(num, den)
        return a, (v, z)

    def assign_cache_integer(self, p, sc, g, offset):                     # :734-751
        assert p.times == 1
        for j in range(self.ic.info.limbs):
            self.ic.assign_cache_value(p.limbs_le[j], offset, g, sc)
            offset += 1
        self.ic.assign_cache_value(p.native, offset, g, sc)
        return offset + 1

    def assign_selected_integer(self, p, sc, g, offset):                  # :753-777
        limbs = []
        for j in range(self.ic.info.limbs):
            limbs.append(self.ic.assign_selected_value(p.limbs_le[j], offset, g, sc))
            offset += 1
        native = self.ic.assign_selected_value(p.native, offset, g, sc)
        return AInt(limbs, native, 1), offset + 1

    def lambda_to_point_non_zero(self, l, a, b):                          # :814-838
        ic = self.ic
        t = ic.int_sub(ic.int_square(l), a.x)
        cx = ic.int_sub(t, b.x)
        t = ic.int_sub(a.x, cx)
        t = ic.int_mul(t, l)
        cy = ic.int_sub(t, a.y)
        return Pt(cx, cy)

    def ecc_add_unsafe(self, a, b):                                       # :840-858
        ic = self.ic
        self.ctx.s.counts["ecc_add_unsafe"] += 1
        diff_x = ic.int_sub(a.x, b.x)
        diff_y = ic.int_sub(a.y, b.y)
        x_eq, tangent = ic.int_div(diff_y, diff_x)
        ok = self.ctx.try_assert_false(x_eq)
        res = self.lambda_to_point_non_zero(tangent, a, b)
        if not ok:
            raise UnsafeError("AddSameOrNegPoint")
        return res

    def ecc_double_unsafe(self, a):                                       # :860-882
        ic = self.ic
        self.ctx.s.counts["ecc_double_unsafe"] += 1
        x_square = ic.int_square(a.x)
        num = ic.int_mul_small_constant(x_square, 3)
        den = ic.int_mul_small_constant(a.y, 2)
        z, v = ic.int_div(num, den)
        ok = self.ctx.try_assert_false(z)
        res = self.lambda_to_point_non_zero(v, a, a)
        if not ok:
            raise UnsafeError("AddIdentity")
        return res

    def ecc_neg_non_zero(self, a):                                        # :884-889
        return Pt(a.x, self.ic.int_neg(a.y))

    def ecc_reduce_non_zero(self, a):                                     # :891-899
        return Pt(self.ic.reduce(a.x), self.ic.reduce(a.y))

    def ecc_bisec_non_zero_point(self, cond, a, b):                       # :901-911
        return Pt(self.ic.bisec_int(cond, a.x, b.x), self.ic.bisec_int(cond, a.y, b.y))

    def bisec_candidate_non_zero(self, candidates, group_bits):           # :913-933
        curr = list(candidates)
        for bit in group_bits:
            curr = [self.ecc_bisec_non_zero_point(bit, curr[k + 1], curr[k]) for k in range(0, len(curr), 2)]
        assert len(curr) == 1
        return curr[0]

    def pick_candidate_non_zero(self, candidates, group_bits):            # :935-953
        index = self.ctx.sum_with_constant([(x, 1 << i) for i, x in enumerate(group_bits)], None)
        return index, candidates[index.val & 0xFF]

    def assign_selected_point_non_zero(self, p, sc, g):                   # :955-967
        x, i = self.assign_selected_integer(p.x, sc, g, 0)
        y, i = self.assign_selected_integer(p.y, sc, g, i)
        return Pt(x, y)

    def assign_cache_point_non_zero(self, p, g, sc):                      # :969-973
        i = self.assign_cache_integer(p.x, sc, g, 0)
        self.assign_cache_integer(p.y, sc, g, i)

    def ecc_non_zero_point_downgrade(self, a):                            # :984-997
        return Pt(a.x, a.y, self.ctx.assign_constant(0))

    def ecc_bisec_to_non_zero_point(self, a, b):                          # :999-1008
        return Pt(self.ic.bisec_int(a.z, b.x, a.x), self.ic.bisec_int(a.z, b.y, a.y))

    # ---- EccChipScalarOps (native scalars) ----
    def decompose_scalar(self, s):                                        # native_scalar_ecc_chip.rs:97-171, WINDOW_SIZE = 1
        ctx = self.ctx
        bits = []
        s_bn = s.val
        v = s
        for i in range(self.scalar_bits // 2):
            b0 = ctx.assign_bit((s_bn >> (2 * i)) & 1)
            b1 = ctx.assign_bit((s_bn >> (2 * i + 1)) & 1)
            v_next = (s_bn >> (2 * i + 2)) % N_MOD
            cells = ctx.one_line_with_last([(v_next, 4), (b1, 2), (b0, 1)], (v, -1))
            v = cells[0][0]
            bits += [b0, b1]
        if self.scalar_bits & 1:
            ctx.assert_bit(v)
            bits.append(v)
        else:
            ctx.assert_constant(v, 0)
        return [[b] for b in reversed(bits)]

    def _msm_batch(self, points, scalars, r1, r2, with_select):           # ecc_chip.rs:223-371 / :91-221
        points = [self.ecc_reduce_non_zero(p) for p in points]
        rand_acc_point = self.assign_non_zero_point(r1)
        rand_line_point = self.assign_non_zero_point(r2)
        rand_acc_point_neg = self.ecc_reduce_non_zero(self.ecc_neg_non_zero(rand_acc_point))
        rand_line_point_neg = self.ecc_reduce_non_zero(self.ecc_neg_non_zero(rand_line_point))
        best_group_size = 5 if with_select else 2
        n_group = (len(points) + best_group_size - 1) // best_group_size
        group_size = (len(points) + n_group - 1) // n_group
        candidates = []
        group_prefix = self.get_and_increase_msm_prefix() if with_select else 0
        chunks = [points[k:k + group_size] for k in range(0, len(points), group_size)]
        for group_index, chunk in enumerate(chunks):
            init = rand_line_point if group_index % 2 == 0 else rand_line_point_neg
            cl = [init]
            candidates.append(cl)
            if with_select:
                self.assign_cache_point_non_zero(init, group_prefix + group_index, 0)
            for i in range(1, 1 << len(chunk)):
                pos = (i & -i).bit_length() - 1                         # the last 1-bit position
                other = i - (1 << pos)
                p = self.ecc_reduce_non_zero(self.ecc_add_unsafe(cl[other], chunk[pos]))
                if with_select:
                    self.assign_cache_point_non_zero(p, group_prefix + group_index, i)
                cl.append(p)
        bits = [self.decompose_scalar(s) for s in scalars]
        groups = [bits[k:k + group_size] for k in range(0, len(bits), group_size)]
        windows = len(bits[0])

        def window(op, wi):
            acc = rand_acc_point_neg
            for group_index in range(len(groups)):
                group_bits = [b[wi][0] for b in groups[group_index]]
                if with_select:
                    index_cell, ci = op.pick_candidate_non_zero(candidates[group_index], group_bits)
                    ci = op.assign_selected_point_non_zero(ci, index_cell, group_index + group_prefix)
                else:
                    ci = op.bisec_candidate_non_zero(candidates[group_index], group_bits)
                acc = op.ecc_add_unsafe(ci, acc)
            return acc

        predict_ops = self.clone_with_offset((0, 0, 0))
        before = predict_ops.ctx.offset()
        line_acc_arr = [window(predict_ops, 0)]
        after = predict_ops.ctx.offset()
        diff = tuple(a - b for a, b in zip(after, before))
        self.ctx.merge(predict_ops.ctx)
        cloned = [(i, self.clone_with_offset(tuple(d * i for d in diff))) for i in range(1, windows)]
        for wi, op in cloned:
            b0 = op.ctx.offset()
            line_acc_arr.append(window(op, wi))
            assert tuple(a - b for a, b in zip(op.ctx.offset(), b0)) == diff        # :339 (quirk Q10)
        for _, op in cloned:
            self.ctx.merge(op.ctx)
        self.ctx.apply_offset_diff(tuple(d * windows for d in diff))
        acc = rand_acc_point
        for wi in range(windows):
            acc = self.ecc_double_unsafe(acc)
            acc = self.ecc_add_unsafe(line_acc_arr[wi], acc)
            if len(groups) % 2 == 1:
                acc = self.ecc_add_unsafe(acc, rand_line_point_neg)
        acc = self.ecc_non_zero_point_downgrade(acc)
        acc, curv = self.to_point_with_curvature(acc)
        carry = self.ecc_non_zero_point_downgrade(rand_acc_point_neg)
        return self.ecc_add(acc, curv, carry)

    def ecc_bisec_scalar(self, cond, a, b):                               # native_scalar_ecc_chip.rs:180-187
        return self.ctx.bisec(cond, a, b)

    def ecc_assign_constant_zero_scalar(self):                            # :188-192
        return self.ctx.assign_constant(0)

    def msm_unsafe(self, points, scalars, r1, r2):                        # ecc_chip.rs:373-408 (quirk Q1: r1, r2 are inputs here)
        non_zero_p = self.assign_non_zero_point(self.generator)
        s_zero = self.ecc_assign_constant_zero_scalar()
        nz_points, nz_scalars = [], []
        for p, s in zip(points, scalars):                                 # quirk Q9: every input goes through both bisecs
            s2 = self.ecc_bisec_scalar(p.z, s_zero, s)
            p2 = self.ecc_bisec_to_non_zero_point(p, non_zero_p)
            nz_points.append(p2)
            nz_scalars.append(s2)
        return self._msm_batch(nz_points, nz_scalars, r1, r2, self.has_select_chip())


class GeneralScalarEccContext(NativeScalarEccContext):
    """GeneralScalarEccContext<C, N> (context.rs:215-239, circuit/general_scalar_ecc_chip.rs): a second integer context over
    the curve's scalar field on the same Context; scalars are AssignedIntegers of that context"""

    def __init__(self, ic, sc, curve_b, generator, msm_prefix=0):
        super().__init__(ic, curve_b, generator, 0, msm_prefix)
        self.sc = sc

    def clone_with_offset(self, d):                                       # general_scalar_ecc_chip.rs:50-72
        c = self.ctx.clone_with_offset(d)
        return GeneralScalarEccContext(self.ic.fork(c), self.sc.fork(c), self.curve_b, self.generator, self.prefix)

    def decompose_scalar(self, s):                                        # :96-147, WINDOW_SIZE = 1
        ctx = self.ctx
        two_inv = pow(2, -1, N_MOD)
        s = self.sc.reduce(s)
        bits = []
        for l in s.limbs_le:
            v = l.val
            rest = l
            for j in range(self.sc.info.limb_bits):
                b = ctx.assign_bit((v >> j) & 1)
                nv = (rest.val - b.val) * two_inv % N_MOD
                rest = ctx.one_line_with_last([(rest, -1), (b, 1)], (nv, 2))[1]
                bits.append(b)
            ctx.assert_constant(rest, 0)
        return [[b] for b in reversed(bits)]

    def ecc_bisec_scalar(self, cond, a, b):                               # :156-163
        return self.sc.bisec_int(cond, a, b)

    def ecc_assign_constant_zero_scalar(self):                            # :165-168
        return self.sc.assign_int_constant(0)


# ===================================================================================================
# L4: Fq2 / Fq6 / Fq12 tower and pairing (circuit/fq12.rs, pairing_chip.rs, bn256_pairing_chip.rs, bls12_381_pairing_chip.rs)
# 6u+2 of bn256 as the signed-digit table the reference walks (circuit/bn256_constants.rs:8-12; it is NOT the canonical
# NAF, so it is data of the algorithm; checked below against BN_X)
BN_X = 4965661367192848881                                               # bn256_constants.rs:6
SIX_U_PLUS_2_NAF = [0, 0, 0, 1, 0, 1, 0, -1, 0, 0, 1, -1, 0, 0, 1, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, -1, 0, 0, 0, 0,
                    1, 1, 1, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0, -1, 0, 0, 1, 1, 0, 0, -1, 0, 0, 0, 1, 1, 0, -1, 0,
                    0, 1, 0, 1, 1]
assert sum(d << i for i, d in enumerate(SIX_U_PLUS_2_NAF)) == 6 * BN_X + 2
BLS_X = 0xD201000000010000                                               # bls12_381_pairing_chip.rs:162


def _fp2_mul(a, b, q):
    return ((a[0] * b[0] - a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)


def _fp2_pow(a, e, q):
    r = (1, 0)
    while e:
        if e & 1:
            r = _fp2_mul(r, a, q)
        a = _fp2_mul(a, a, q)
        e >>= 1
    return r


def bn256_frobenius_constants():
    """The constants of circuit/bn256_constants.rs:14-283 from their definitions (xi = 9 + u):
    FQ2_C1[i] = (-1)^i, FQ6_C1[i] = xi^((q^i - 1) / 3), FQ6_C2[i] = xi^((2 q^i - 2) / 3), FQ12_C1[i] = xi^((q^i - 1) / 6),
    XI_TO_Q_MINUS_1_OVER_2 = xi^((q - 1) / 2).  tests/golden/make_pyref_golden.py checks them against the reference's tables."""
    q, xi = BN_Q, (9, 1)
    return {
        "fq2_c1": [1, q - 1],
        "fq6_c1": [_fp2_pow(xi, (q ** i - 1) // 3, q) for i in range(6)],
        "fq6_c2": [_fp2_pow(xi, (2 * q ** i - 2) // 3, q) for i in range(6)],
        "fq12_c1": [_fp2_pow(xi, (q ** i - 1) // 6, q) for i in range(12)],
        "xi_q12": _fp2_pow(xi, (q - 1) // 2, q),
    }


def bls12_381_frobenius_constants():
    """bls12_381_pairing_chip.rs:58-107 gives Montgomery-form raw limbs; their canonical values are
    (1 + u)^((q - 1) / 3) = (0, c1), (1 + u)^(2 (q - 1) / 3) = (c2, 0) and (1 + u)^((q - 1) / 6)"""
    q, xi = BLS_Q, (1, 1)
    return {"fq6_c1": _fp2_pow(xi, (q - 1) // 3, q), "fq6_c2": _fp2_pow(xi, (2 * q - 2) // 3, q), "fq12_c1": _fp2_pow(xi, (q - 1) // 6, q)}


class G2A:
    __slots__ = ("x", "y", "z")

    def __init__(self, x, y, z):
        self.x, self.y, self.z = x, y, z


class PairingOps:
    """Fq2ChipOps / Fq6ChipOps / Fq12ChipOps (circuit/fq12.rs:24-459) + PairingChipOps (circuit/pairing_chip.rs:13-176)"""

    def __init__(self, ecc):
        self.ecc, self.ic, self.ctx = ecc, ecc.ic, ecc.ic.ctx

    # ---- Fq2 (fq12.rs:24-104) ----
    def fq2_reduce(self, x):
        return (self.ic.reduce(x[0]), self.ic.reduce(x[1]))

    def fq2_assert_equal(self, x, y):
        self.ic.assert_int_equal(x[0], y[0])
        self.ic.assert_int_equal(x[1], y[1])

    def fq2_assign_zero(self):
        z = self.ic.assign_int_constant(0)
        return (z, z)

    def fq2_assign_one(self):
        return (self.ic.assign_int_constant(1), self.ic.assign_int_constant(0))

    def fq2_assign_constant(self, c):
        return (self.ic.assign_int_constant(c[0]), self.ic.assign_int_constant(c[1]))

    def fq2_add(self, a, b):
        return (self.ic.int_add(a[0], b[0]), self.ic.int_add(a[1], b[1]))

    def fq2_mul(self, a, b):                                              # :57-69
        ic = self.ic
        self.ctx.s.counts["fq2_mul"] += 1
        ab00 = ic.int_mul(a[0], b[0])
        ab11 = ic.int_mul(a[1], b[1])
        c0 = ic.int_sub(ab00, ab11)
        a01 = ic.int_add(a[0], a[1])
        b01 = ic.int_add(b[0], b[1])
        c1 = ic.int_mul(a01, b01)
        c1 = ic.int_sub(c1, ab00)
        c1 = ic.int_sub(c1, ab11)
        return (c0, c1)

    def fq2_sub(self, a, b):
        return (self.ic.int_sub(a[0], b[0]), self.ic.int_sub(a[1], b[1]))

    def fq2_double(self, a):
        return (self.ic.int_add(a[0], a[0]), self.ic.int_add(a[1], a[1]))

    def fq2_square(self, a):
        return self.fq2_mul(a, a)

    def fq2_neg(self, a):
        return (self.ic.int_neg(a[0]), self.ic.int_neg(a[1]))

    def fq2_conjugate(self, a):
        return (a[0], self.ic.int_neg(a[1]))

    def fq2_unsafe_invert(self, x):                                       # :93-103
        ic = self.ic
        t0 = ic.int_square(x[0])
        t1 = ic.int_square(x[1])
        t0 = ic.int_add(t0, t1)
        t = ic.int_unsafe_invert(t0)
        c0 = ic.int_mul(x[0], t)
        c1 = ic.int_mul(x[1], t)
        return (c0, ic.int_neg(c1))

    # ---- Fq6 (fq12.rs:106-287) ----
    def fq6_reduce(self, x):
        return (self.fq2_reduce(x[0]), self.fq2_reduce(x[1]), self.fq2_reduce(x[2]))

    def fq6_assert_equal(self, x, y):
        for k in range(3):
            self.fq2_assert_equal(x[k], y[k])

    def fq6_assign_zero(self):
        z = self.fq2_assign_zero()
        return (z, z, z)

    def fq6_assign_one(self):
        one = self.fq2_assign_one()
        z = self.fq2_assign_zero()
        return (one, z, z)

    def fq6_add(self, a, b):
        return (self.fq2_add(a[0], b[0]), self.fq2_add(a[1], b[1]), self.fq2_add(a[2], b[2]))

    def fq6_sub(self, a, b):
        return (self.fq2_sub(a[0], b[0]), self.fq2_sub(a[1], b[1]), self.fq2_sub(a[2], b[2]))

    def fq6_neg(self, a):
        return (self.fq2_neg(a[0]), self.fq2_neg(a[1]), self.fq2_neg(a[2]))

    def fq6_mul(self, a, b):                                              # :135-170
        ab00 = self.fq2_mul(a[0], b[0])
        ab11 = self.fq2_mul(a[1], b[1])
        ab22 = self.fq2_mul(a[2], b[2])
        b12 = self.fq2_add(b[1], b[2])
        a12 = self.fq2_add(a[1], a[2])
        t = self.fq2_mul(a12, b12)
        t = self.fq2_sub(t, ab11)
        t = self.fq2_sub(t, ab22)
        t = self.fq2_mul_by_nonresidue(t)
        c0 = self.fq2_add(t, ab00)
        b01 = self.fq2_add(b[0], b[1])
        a01 = self.fq2_add(a[0], a[1])
        t = self.fq2_mul(a01, b01)
        t = self.fq2_sub(t, ab00)
        t = self.fq2_sub(t, ab11)
        ab22n = self.fq2_mul_by_nonresidue(ab22)
        c1 = self.fq2_add(t, ab22n)
        b02 = self.fq2_add(b[0], b[2])
        a02 = self.fq2_add(a[0], a[2])
        t = self.fq2_mul(a02, b02)
        t = self.fq2_sub(t, ab00)
        t = self.fq2_add(t, ab11)
        c2 = self.fq2_sub(t, ab22)
        return (c0, c1, c2)

    def fq6_square(self, a):
        return self.fq6_mul(a, a)

    def fq6_mul_by_1(self, a, b1):                                        # :190-212
        ab11 = self.fq2_mul(a[1], b1)
        a12 = self.fq2_add(a[1], a[2])
        t = self.fq2_mul(a12, b1)
        t = self.fq2_sub(t, ab11)
        c0 = self.fq2_mul_by_nonresidue(t)
        a01 = self.fq2_add(a[0], a[1])
        t = self.fq2_mul(a01, b1)
        c1 = self.fq2_sub(t, ab11)
        return (c0, c1, ab11)

    def fq6_mul_by_01(self, a, b0, b1):                                   # :213-249
        ab00 = self.fq2_mul(a[0], b0)
        ab11 = self.fq2_mul(a[1], b1)
        a12 = self.fq2_add(a[1], a[2])
        t = self.fq2_mul(a12, b1)
        t = self.fq2_sub(t, ab11)
        t = self.fq2_mul_by_nonresidue(t)
        c0 = self.fq2_add(t, ab00)
        b01 = self.fq2_add(b0, b1)
        a01 = self.fq2_add(a[0], a[1])
        t = self.fq2_mul(a01, b01)
        t = self.fq2_sub(t, ab00)
        c1 = self.fq2_sub(t, ab11)
        a02 = self.fq2_add(a[0], a[2])
        t = self.fq2_mul(a02, b0)
        t = self.fq2_sub(t, ab00)
        c2 = self.fq2_add(t, ab11)
        return (c0, c1, c2)

    def fq6_unsafe_invert(self, x):                                       # :250-279
        c0 = self.fq2_mul_by_nonresidue(x[2])
        c0 = self.fq2_mul(c0, x[1])
        c0 = self.fq2_neg(c0)
        x0s = self.fq2_square(x[0])
        c0 = self.fq2_add(c0, x0s)
        c1 = self.fq2_square(x[2])
        c1 = self.fq2_mul_by_nonresidue(c1)
        x01 = self.fq2_mul(x[0], x[1])
        c1 = self.fq2_sub(c1, x01)
        c2 = self.fq2_square(x[1])
        x02 = self.fq2_mul(x[0], x[2])
        c2 = self.fq2_sub(c2, x02)
        c0x0 = self.fq2_mul(c0, x[0])
        c1x2 = self.fq2_mul(c1, x[2])
        c2x1 = self.fq2_mul(c2, x[1])
        t = self.fq2_add(c1x2, c2x1)
        t = self.fq2_mul_by_nonresidue(t)
        t = self.fq2_add(t, c0x0)
        t = self.fq2_unsafe_invert(t)
        return (self.fq2_mul(t, c0), self.fq2_mul(t, c1), self.fq2_mul(t, c2))

    def fq6_mul_by_nonresidue(self, a):                                   # bn256_pairing_chip.rs:59-61 / bls12_381_pairing_chip.rs:47-49
        return (self.fq2_mul_by_nonresidue(a[2]), a[0], a[1])

    # ---- Fq12 (fq12.rs:289-459) ----
    def fq12_assert_eq(self, x, y):
        self.fq6_assert_equal(x[0], y[0])
        self.fq6_assert_equal(x[1], y[1])

    def fq12_assign_one(self):
        one = self.fq6_assign_one()
        z = self.fq6_assign_zero()
        return (one, z)

    def fq12_assert_one(self, x):                                         # :296-299
        self.fq12_assert_eq(x, self.fq12_assign_one())

    def fq12_assign_constant(self, c):
        return (tuple(self.fq2_assign_constant(v) for v in c[0]), tuple(self.fq2_assign_constant(v) for v in c[1]))

    def fq12_mul(self, a, b):                                             # :316-330
        self.ctx.s.counts["fq12_mul"] += 1
        ab00 = self.fq6_mul(a[0], b[0])
        ab11 = self.fq6_mul(a[1], b[1])
        a01 = self.fq6_add(a[0], a[1])
        b01 = self.fq6_add(b[0], b[1])
        c1 = self.fq6_mul(a01, b01)
        c1 = self.fq6_sub(c1, ab00)
        c1 = self.fq6_sub(c1, ab11)
        ab11 = self.fq6_mul_by_nonresidue(ab11)
        c0 = self.fq6_add(ab00, ab11)
        return (c0, c1)

    def fq12_square(self, a):
        return self.fq12_mul(a, a)

    def fq12_conjugate(self, x):
        return (x[0], self.fq6_neg(x[1]))

    def fq12_mul_by_014(self, x, c0, c1, c4):                             # :346-366
        t0 = self.fq6_mul_by_01(x[0], c0, c1)
        t1 = self.fq6_mul_by_1(x[1], c4)
        o = self.fq2_add(c1, c4)
        x0 = self.fq6_mul_by_nonresidue(t1)
        x0 = self.fq6_add(x0, t0)
        x1 = self.fq6_add(x[0], x[1])
        x1 = self.fq6_mul_by_01(x1, c0, o)
        x1 = self.fq6_sub(x1, t0)
        x1 = self.fq6_sub(x1, t1)
        return (x0, x1)

    def fq12_mul_by_034(self, x, c0, c3, c4):                             # :367-388
        t0 = (self.fq2_mul(x[0][0], c0), self.fq2_mul(x[0][1], c0), self.fq2_mul(x[0][2], c0))
        t1 = self.fq6_mul_by_01(x[1], c3, c4)
        t2 = self.fq6_add(x[0], x[1])
        o = self.fq2_add(c0, c3)
        t2 = self.fq6_mul_by_01(t2, o, c4)
        t2 = self.fq6_sub(t2, t0)
        x1 = self.fq6_sub(t2, t1)
        t1 = self.fq6_mul_by_nonresidue(t1)
        x0 = self.fq6_add(t0, t1)
        return (x0, x1)

    def fp4_square(self, a0, a1):                                         # :389-404 (returns (c0, c1))
        t0 = self.fq2_square(a0)
        t1 = self.fq2_square(a1)
        t2 = self.fq2_mul_by_nonresidue(t1)
        c0 = self.fq2_add(t2, t0)
        t2 = self.fq2_add(a0, a1)
        t2 = self.fq2_square(t2)
        t2 = self.fq2_sub(t2, t0)
        c1 = self.fq2_sub(t2, t1)
        return c0, c1

    def fq12_cyclotomic_square(self, x):                                  # :405-440
        self.ctx.s.counts["cyclotomic_square"] += 1
        self.fq2_assign_zero()                                            # quirk Q6: assigned and overwritten
        t3, t4 = self.fp4_square(x[0][0], x[1][1])
        t2 = self.fq2_sub(t3, x[0][0])
        t2 = self.fq2_double(t2)
        c00 = self.fq2_add(t2, t3)
        t2 = self.fq2_add(t4, x[1][1])
        t2 = self.fq2_double(t2)
        c11 = self.fq2_add(t2, t4)
        t3, t4 = self.fp4_square(x[1][0], x[0][2])
        t5, t6 = self.fp4_square(x[0][1], x[1][2])
        t2 = self.fq2_sub(t3, x[0][1])
        t2 = self.fq2_double(t2)
        c01 = self.fq2_add(t2, t3)
        t2 = self.fq2_add(t4, x[1][2])
        t2 = self.fq2_double(t2)
        c12 = self.fq2_add(t2, t4)
        t3 = self.fq2_mul_by_nonresidue(t6)
        t2 = self.fq2_add(t3, x[1][0])
        t2 = self.fq2_double(t2)
        c10 = self.fq2_add(t2, t3)
        t2 = self.fq2_sub(t5, x[0][2])
        t2 = self.fq2_double(t2)
        c02 = self.fq2_add(t2, t5)
        return ((c00, c01, c02), (c10, c11, c12))

    def fq12_unsafe_invert(self, x):                                      # :441-452
        x0s = self.fq6_square(x[0])
        x1s = self.fq6_square(x[1])
        t = self.fq6_mul_by_nonresidue(x1s)
        t = self.fq6_sub(x0s, t)
        t = self.fq6_unsafe_invert(t)
        c0 = self.fq6_mul(t, x[0])
        c1 = self.fq6_mul(t, x[1])
        return (c0, self.fq6_neg(c1))

    # ---- pairing_chip.rs ----
    def doubling_step(self, pt):                                          # :13-76 (pt = [x, y, z], updated in place)
        f = self
        x2 = f.fq2_square(pt[0])
        y2 = f.fq2_square(pt[1])
        _2y2 = f.fq2_double(y2)
        _4y2 = f.fq2_double(_2y2)
        _4y4 = f.fq2_square(_2y2)
        _8y4 = f.fq2_double(_4y4)
        z2 = f.fq2_square(pt[2])
        t = f.fq2_mul(y2, pt[0])
        t = f.fq2_double(t)
        _4xy2 = f.fq2_double(t)
        t = f.fq2_double(x2)
        _3x2 = f.fq2_add(t, x2)
        _6x2 = f.fq2_double(_3x2)
        _9x4 = f.fq2_square(_3x2)
        f.fq2_add(_3x2, pt[0])                                            # quirk Q7: _3x2_x is computed and dropped
        t = f.fq2_sub(_9x4, _4xy2)
        rx = f.fq2_sub(t, _4xy2)
        t = f.fq2_sub(_4xy2, rx)
        t = f.fq2_mul(t, _3x2)
        ry = f.fq2_sub(t, _8y4)
        yz = f.fq2_mul(pt[1], pt[2])
        rz = f.fq2_double(yz)
        t = f.fq2_mul(z2, rz)
        c0 = f.fq2_double(t)
        _6x2z2 = f.fq2_mul(z2, _6x2)
        c1 = f.fq2_neg(_6x2z2)
        _6x3 = f.fq2_mul(_6x2, pt[0])
        c2 = f.fq2_sub(_6x3, _4y2)
        pt[0], pt[1], pt[2] = rx, ry, rz
        return [c0, c1, c2]

    def addition_step(self, pt, pq):                                      # :78-133
        f = self
        zt2 = f.fq2_square(pt[2])
        yqzt = f.fq2_mul(pq.y, pt[2])
        yqzt3 = f.fq2_mul(yqzt, zt2)
        yqzt3_yt = f.fq2_sub(yqzt3, pt[1])
        _2yqzt3_2yt = f.fq2_double(yqzt3_yt)
        xqzt2 = f.fq2_mul(pq.x, zt2)
        xqzt2_xt = f.fq2_sub(xqzt2, pt[0])
        _2_xqzt2_xt = f.fq2_double(xqzt2_xt)
        _4_xqzt2_xt_2 = f.fq2_square(_2_xqzt2_xt)
        t0 = f.fq2_mul(_4_xqzt2_xt_2, xqzt2_xt)
        t1 = f.fq2_double(_4_xqzt2_xt_2)
        t2 = f.fq2_mul(t1, pt[0])
        t = f.fq2_square(_2yqzt3_2yt)
        t = f.fq2_sub(t, t0)
        rx = f.fq2_sub(t, t2)
        t0 = f.fq2_mul(_4_xqzt2_xt_2, pt[0])
        t0 = f.fq2_sub(t0, rx)
        t0 = f.fq2_mul(_2yqzt3_2yt, t0)
        t1 = f.fq2_mul(_2_xqzt2_xt, _4_xqzt2_xt_2)
        t1 = f.fq2_mul(t1, pt[1])
        ry = f.fq2_sub(t0, t1)
        rz = f.fq2_mul(pt[2], _2_xqzt2_xt)
        c0 = f.fq2_double(rz)
        t = f.fq2_double(_2yqzt3_2yt)
        c1 = f.fq2_neg(t)
        t0 = f.fq2_double(_2yqzt3_2yt)
        t0 = f.fq2_mul(t0, pq.x)
        t1 = f.fq2_mul(pq.y, rz)
        t1 = f.fq2_double(t1)
        c2 = f.fq2_sub(t0, t1)
        pt[0], pt[1], pt[2] = rx, ry, rz
        return [c0, c1, c2]

    def g2affine_to_g2(self, g2):                                         # :135-141
        self.ctx.assert_false(g2.z)
        return [g2.x, g2.y, self.fq2_assign_one()]

    def g2_neg(self, g2):                                                 # :143-146
        return G2A(g2.x, self.fq2_neg(g2.y), g2.z)

    def pairing(self, terms):                                             # :157-171
        prepared = [(p, self.prepare_g2(q)) for p, q in terms]
        return self.final_exponentiation(self.multi_miller_loop(prepared))

    def check_pairing(self, terms):                                       # :173-176
        self.fq12_assert_one(self.pairing(terms))


class Bn256Pairing(PairingOps):
    """circuit/bn256_pairing_chip.rs:29-323"""

    def __init__(self, ecc):
        super().__init__(ecc)
        self.k = bn256_frobenius_constants()

    def fq2_mul_by_nonresidue(self, a):                                   # :32-44  (xi = 9 + u by doublings)
        ic = self.ic
        a2 = self.fq2_double(a)
        a4 = self.fq2_double(a2)
        a8 = self.fq2_double(a4)
        t = ic.int_add(a8[0], a[0])
        c0 = ic.int_sub(t, a[1])
        t = ic.int_add(a8[1], a[0])
        c1 = ic.int_add(t, a[1])
        return (c0, c1)

    def fq2_frobenius_map(self, x, power):                                # :46-53 (quirk Q7: multiplies by the constant 1 for even powers)
        v = self.ic.assign_int_constant(self.k["fq2_c1"][power % 2])
        return (x[0], self.ic.int_mul(x[1], v))

    def fq6_frobenius_map(self, x, power):                                # :63-79
        c0 = self.fq2_frobenius_map(x[0], power)
        c1 = self.fq2_frobenius_map(x[1], power)
        c2 = self.fq2_frobenius_map(x[2], power)
        coeff_c1 = self.fq2_assign_constant(self.k["fq6_c1"][power % 6])
        c1 = self.fq2_mul(c1, coeff_c1)
        coeff_c2 = self.fq2_assign_constant(self.k["fq6_c2"][power % 6])
        c2 = self.fq2_mul(c2, coeff_c2)
        return (c0, c1, c2)

    def fq12_frobenius_map(self, x, power):                               # :85-96
        c0 = self.fq6_frobenius_map(x[0], power)
        c1 = self.fq6_frobenius_map(x[1], power)
        coeff = self.fq2_assign_constant(self.k["fq12_c1"][power % 12])
        return (c0, (self.fq2_mul(c1[0], coeff), self.fq2_mul(c1[1], coeff), self.fq2_mul(c1[2], coeff)))

    def prepare_g2(self, g2):                                             # :104-155
        neg_g2 = self.g2_neg(g2)
        coeffs = []
        r = self.g2affine_to_g2(g2)
        for i in reversed(range(1, len(SIX_U_PLUS_2_NAF))):
            coeffs.append(self.doubling_step(r))
            x = SIX_U_PLUS_2_NAF[i - 1]
            if x == 1:
                coeffs.append(self.addition_step(r, g2))
            elif x == -1:
                coeffs.append(self.addition_step(r, neg_g2))
        c11 = self.fq2_assign_constant(self.k["fq6_c1"][1])
        c12 = self.fq2_assign_constant(self.k["fq6_c1"][2])
        xi = self.fq2_assign_constant(self.k["xi_q12"])
        q1x = (g2.x[0], self.ic.int_neg(g2.x[1]))
        q1x = self.fq2_mul(q1x, c11)
        q1y = (g2.y[0], self.ic.int_neg(g2.y[1]))
        q1y = self.fq2_mul(q1y, xi)
        coeffs.append(self.addition_step(r, G2A(q1x, q1y, g2.z)))
        m2x = self.fq2_mul(g2.x, c12)
        coeffs.append(self.addition_step(r, G2A(m2x, g2.y, g2.z)))
        return coeffs

    def ell(self, f, coeffs, p):                                          # :157-174
        ic = self.ic
        c00 = ic.int_mul(coeffs[0][0], p.y)
        c01 = ic.int_mul(coeffs[0][1], p.y)
        c10 = ic.int_mul(coeffs[1][0], p.x)
        c11 = ic.int_mul(coeffs[1][1], p.x)
        return self.fq12_mul_by_034(f, (c00, c01), (c10, c11), coeffs[2])

    def multi_miller_loop(self, terms):                                   # :176-228
        pairs = []
        for p, q in terms:
            self.ctx.assert_false(p.z)
            pairs.append((p, iter(q)))
        f = self.fq12_assign_one()
        n = len(SIX_U_PLUS_2_NAF)
        for i in reversed(range(1, n)):
            if i != n - 1:
                f = self.fq12_square(f)
            for p, it in pairs:
                f = self.ell(f, next(it), p)
            if SIX_U_PLUS_2_NAF[i - 1] != 0:
                for p, it in pairs:
                    f = self.ell(f, next(it), p)
        for _ in range(2):
            for p, it in pairs:
                f = self.ell(f, next(it), p)
        for _, it in pairs:
            assert next(it, None) is None
        return f

    def exp_by_x(self, f):                                                # :230-240
        res = self.fq12_assign_one()
        for i in reversed(range(64)):
            res = self.fq12_cyclotomic_square(res)
            if (BN_X >> i) & 1:
                res = self.fq12_mul(res, f)
        return res

    def final_exponentiation(self, f):                                    # :242-323
        s = self
        f1 = s.fq12_conjugate(f)
        f2 = s.fq12_unsafe_invert(f)
        r = s.fq12_mul(f1, f2)
        f2 = r
        r = s.fq12_frobenius_map(r, 2)
        r = s.fq12_mul(r, f2)
        fp = s.fq12_frobenius_map(r, 1)
        fp2 = s.fq12_frobenius_map(r, 2)
        fp3 = s.fq12_frobenius_map(fp2, 1)
        fu = s.exp_by_x(r)
        fu2 = s.exp_by_x(fu)
        fu3 = s.exp_by_x(fu2)
        y3 = s.fq12_frobenius_map(fu, 1)
        fu2p = s.fq12_frobenius_map(fu2, 1)
        fu3p = s.fq12_frobenius_map(fu3, 1)
        y2 = s.fq12_frobenius_map(fu2, 2)
        y0 = s.fq12_mul(fp, fp2)
        y0 = s.fq12_mul(y0, fp3)
        y1 = s.fq12_conjugate(r)
        y5 = s.fq12_conjugate(fu2)
        y3 = s.fq12_conjugate(y3)
        y4 = s.fq12_mul(fu, fu2p)
        y4 = s.fq12_conjugate(y4)
        y6 = s.fq12_mul(fu3, fu3p)
        y6 = s.fq12_conjugate(y6)
        y6 = s.fq12_cyclotomic_square(y6)
        y6 = s.fq12_mul(y6, y4)
        y6 = s.fq12_mul(y6, y5)
        t1 = s.fq12_mul(y3, y5)
        t1 = s.fq12_mul(t1, y6)
        y6 = s.fq12_mul(y6, y2)
        t1 = s.fq12_cyclotomic_square(t1)
        t1 = s.fq12_mul(t1, y6)
        t1 = s.fq12_cyclotomic_square(t1)
        t0 = s.fq12_mul(t1, y1)
        t1 = s.fq12_mul(t1, y0)
        t0 = s.fq12_cyclotomic_square(t0)
        return s.fq12_mul(t0, t1)


class Bls12381Pairing(PairingOps):
    """circuit/bls12_381_pairing_chip.rs:29-286"""

    def __init__(self, ecc):
        super().__init__(ecc)
        self.k = bls12_381_frobenius_constants()

    def fq2_mul_by_nonresidue(self, a):                                   # :32-37 (xi = 1 + u)
        return (self.ic.int_sub(a[0], a[1]), self.ic.int_add(a[0], a[1]))

    def fq2_frobenius_map(self, x, _power):                               # :39-41
        return self.fq2_conjugate(x)

    def fq6_frobenius_map(self, x, power):                                # :51-82 (quirk Q7: `power` is ignored)
        c0 = self.fq2_frobenius_map(x[0], power)
        c1 = self.fq2_frobenius_map(x[1], power)
        c2 = self.fq2_frobenius_map(x[2], power)
        coeff_c1 = self.fq2_assign_constant(self.k["fq6_c1"])
        c1 = self.fq2_mul(c1, coeff_c1)
        coeff_c2 = self.fq2_assign_constant(self.k["fq6_c2"])
        c2 = self.fq2_mul(c2, coeff_c2)
        return (c0, c1, c2)

    def fq12_frobenius_map(self, x, power):                               # :88-115
        c0 = self.fq6_frobenius_map(x[0], power)
        c1 = self.fq6_frobenius_map(x[1], power)
        coeff = self.fq2_assign_constant(self.k["fq12_c1"])
        return (c0, (self.fq2_mul(c1[0], coeff), self.fq2_mul(c1[1], coeff), self.fq2_mul(c1[2], coeff)))

    def ell(self, f, coeffs, p):                                          # :123-140
        ic = self.ic
        c00 = ic.int_mul(coeffs[0][0], p.y)
        c01 = ic.int_mul(coeffs[0][1], p.y)
        c10 = ic.int_mul(coeffs[1][0], p.x)
        c11 = ic.int_mul(coeffs[1][1], p.x)
        return self.fq12_mul_by_014(f, coeffs[2], (c10, c11), (c00, c01))

    def cyclotomic_exp(self, f):                                          # :142-159
        tmp = self.fq12_assign_one()
        found_one = False
        for b in reversed(range(64)):
            i = (BLS_X >> b) & 1 == 1
            if found_one:
                tmp = self.fq12_cyclotomic_square(tmp)
            else:
                found_one = i
            if i:
                tmp = self.fq12_mul(tmp, f)
        return self.fq12_conjugate(tmp)

    def prepare_g2(self, g2):                                             # :165-189
        f = self.g2affine_to_g2(g2)
        coeffs = []
        found_one = False
        for b in reversed(range(64)):
            i = ((BLS_X >> 1) >> b) & 1 == 1
            if not found_one:
                found_one = i
                continue
            coeffs.append(self.doubling_step(f))
            if i:
                coeffs.append(self.addition_step(f, g2))
        coeffs.append(self.doubling_step(f))
        return coeffs

    def multi_miller_loop(self, terms):                                   # :191-234
        pairs = []
        for p, q in terms:
            self.ctx.assert_false(p.z)
            pairs.append((p, iter(q)))
        f = self.fq12_assign_one()
        found_one = False
        for b in reversed(range(64)):
            i = ((BLS_X >> 1) >> b) & 1 == 1
            if not found_one:
                found_one = i
                continue
            for p, it in pairs:
                f = self.ell(f, next(it), p)
            if i:
                for p, it in pairs:
                    f = self.ell(f, next(it), p)
            f = self.fq12_square(f)
        for p, it in pairs:
            f = self.ell(f, next(it), p)
        return self.fq12_conjugate(f)

    def final_exponentiation(self, f):                                    # :236-286
        s = self
        t0 = s.fq12_frobenius_map(f, 1)
        for _ in range(5):
            t0 = s.fq12_frobenius_map(t0, 1)
        t1 = s.fq12_unsafe_invert(f)
        t2 = s.fq12_mul(t0, t1)
        t1 = t2
        t2 = s.fq12_frobenius_map(t2, 1)
        t2 = s.fq12_frobenius_map(t2, 1)
        t2 = s.fq12_mul(t2, t1)
        t1 = s.fq12_cyclotomic_square(t2)
        t1 = s.fq12_conjugate(t1)
        t3 = s.cyclotomic_exp(t2)
        t4 = s.fq12_cyclotomic_square(t3)
        t5 = s.fq12_mul(t1, t3)
        t1 = s.cyclotomic_exp(t5)
        t0 = s.cyclotomic_exp(t1)
        t6 = s.cyclotomic_exp(t0)
        t6 = s.fq12_mul(t6, t4)
        t4 = s.cyclotomic_exp(t6)
        t5 = s.fq12_conjugate(t5)
        t = s.fq12_mul(t5, t2)
        t4 = s.fq12_mul(t4, t)
        t5 = s.fq12_conjugate(t2)
        t1 = s.fq12_mul(t1, t2)
        for _ in range(3):
            t1 = s.fq12_frobenius_map(t1, 1)
        t6 = s.fq12_mul(t6, t5)
        t6 = s.fq12_frobenius_map(t6, 1)
        t3 = s.fq12_mul(t3, t0)
        for _ in range(2):
            t3 = s.fq12_frobenius_map(t3, 1)
        t3 = s.fq12_mul(t3, t1)
        t3 = s.fq12_mul(t3, t6)
        return s.fq12_mul(t3, t4)


# ===================================================================================================
# workloads: the same bodies and input vectors as include/h2e.h's programs / oracle_capi.cpp's runs
def _w(inputs, slot):
    return sum(int(w) << (64 * k) for k, w in enumerate(inputs[slot]))


W_MOD = {0: BN_Q, 1: BLS_Q, 2: BLS_R}


def run_int_mul_batch(fp, n, inputs):
    ctx = Context()
    ic = IntegerContext(ctx, W_MOD[fp])
    for k in range(n):
        a = ic.assign_w(_w(inputs, 2 * k))
        b = ic.assign_w(_w(inputs, 2 * k + 1))
        ic.int_mul(a, b)
    return ctx


def run_integer_chip_st(fp, inputs):                                      # tests/integer_chip.rs:11-55
    ctx = Context()
    ic = IntegerContext(ctx, W_MOD[fp])
    a, b = ic.assign_w(_w(inputs, 0)), ic.assign_w(_w(inputs, 1))
    c1 = ic.assign_w(_w(inputs, 2))
    ic.assert_int_equal(c1, ic.int_add(a, b))
    d1 = ic.assign_w(_w(inputs, 3))
    ic.assert_int_equal(d1, ic.int_sub(a, b))
    e1 = ic.assign_w(_w(inputs, 4))
    ic.assert_int_equal(e1, ic.int_mul(a, b))
    f1 = ic.assign_w(_w(inputs, 5))
    ic.assert_int_equal(f1, ic.int_div(a, b)[1])
    zero = ic.int_sub(a, a)
    g = ic.int_div(a, zero)
    ctx.assert_true(g[0])
    return ctx


def _pt(inputs, xs, ys, zs):
    return None if _w(inputs, zs) != 0 else (_w(inputs, xs), _w(inputs, ys))


def run_msm_bn256_tile(n, inputs, with_select=True):                      # tests/native_scalar_ecc_chip.rs:34-47
    ctx = Context()
    ic = IntegerContext(ctx, BN_Q)
    gen = (_w(inputs, 4 * n), _w(inputs, 4 * n + 1))
    ecc = NativeScalarEccContext(ic, 3, gen, 254, 0 if with_select else None)
    points = [ecc.assign_point(_pt(inputs, 3 * k, 3 * k + 1, 3 * k + 2)) for k in range(n)]
    scalars = [ctx.assign(_w(inputs, 3 * n + k)) for k in range(n)]
    r1 = (_w(inputs, 4 * n + 2), _w(inputs, 4 * n + 3))
    r2 = (_w(inputs, 4 * n + 4), _w(inputs, 4 * n + 5))
    before = (ctx.offset(), Counter(ctx.s.counts))
    res = ecc.msm_unsafe(points, scalars, r1, r2)
    ctx.s.marks = {"msm_unsafe_rows": [a - b for a, b in zip(ctx.offset(), before[0])],
                   "msm_unsafe_counts": dict(sorted((ctx.s.counts - before[1]).items()))}
    res_expect = ecc.assign_point(_pt(inputs, 4 * n + 6, 4 * n + 7, 4 * n + 8))
    ecc.ecc_assert_equal(res, res_expect)
    return ctx


def run_ops_msm_twice(n, inputs):
    """operator-API scenario (tests/test_ops_gpu.py): ops one after the other on one context - assign_point x n, assign x n,
    int_mul / int_add / int_sub / reduce / int_div, msm_unsafe twice (the second with msm prefix 2^20 and swapped blinding
    points), ecc_assert_equal(res1, res2)"""
    ctx = Context()
    ic = IntegerContext(ctx, BN_Q)
    gen = (_w(inputs, 4 * n), _w(inputs, 4 * n + 1))
    ecc = NativeScalarEccContext(ic, 3, gen, 254, 0)
    points = [ecc.assign_point(_pt(inputs, 3 * k, 3 * k + 1, 3 * k + 2)) for k in range(n)]
    scalars = [ctx.assign(_w(inputs, 3 * n + k)) for k in range(n)]
    m = ic.int_mul(points[0].x, points[0].y)
    s = ic.int_add(m, m)
    s2 = ic.int_sub(s, points[0].x)
    rd = ic.reduce(s2)
    ic.int_div(rd, points[0].y)
    r1 = (_w(inputs, 4 * n + 2), _w(inputs, 4 * n + 3))
    r2 = (_w(inputs, 4 * n + 4), _w(inputs, 4 * n + 5))
    res1 = ecc.msm_unsafe(points, scalars, r1, r2)
    res2 = ecc.msm_unsafe(points, scalars, r2, r1)
    ecc.ecc_assert_equal(res1, res2)
    return ctx


def run_ops_ecc_surface(inputs):
    """operator-API scenario over the complete-addition / curvature surface (tests/test_ops_gpu.py): inputs = P, Q (x, y, z each),
    scalar, index (= 1), generator (x, y), r1 (x, y), r2 (x, y)"""
    ctx = Context()
    ic = IntegerContext(ctx, BN_Q)
    gen = (_w(inputs, 8), _w(inputs, 9))
    ecc = NativeScalarEccContext(ic, 3, gen, 254, 0)
    P = ecc.assign_point(_pt(inputs, 0, 1, 2))
    Q = ecc.assign_point(_pt(inputs, 3, 4, 5))
    s = ctx.assign(_w(inputs, 6))
    idx = ctx.assign(_w(inputs, 7))
    Pr, Pc = ecc.ecc_reduce_with_curvature(P)
    D = ecc.ecc_double(Pr, Pc)
    Qp, Qc = ecc.to_point_with_curvature(Q)
    S = ecc.ecc_add(Qp, Qc, D)
    N = ecc.ecc_neg(S)
    ecc.ecc_encode(N)
    r1 = (_w(inputs, 10), _w(inputs, 11))
    r2 = (_w(inputs, 12), _w(inputs, 13))
    ecc.msm_unsafe([P], [s], r1, r2)                                      # ecc_mul (ecc_chip.rs:418-420)
    C = ecc.assign_constant_point(gen)
    Cp, Cc = ecc.to_point_with_curvature(C)
    ecc.bisec_point_with_curvature(P.z, Pr, Pc, Cp, Cc)
    ecc.assign_cache_point(Pr, Pc, 7, 0)
    ecc.assign_cache_point(Cp, Cc, 7, 1)
    cands = [(Pr, Pc), (Cp, Cc)]
    pick = cands[idx.val & 0xFF]
    Sel, _ = ecc.assign_selected_point(pick[0], pick[1], idx, 7)
    ecc.ecc_assert_equal(Sel, C)
    return ctx


def run_msm_bls12_381_tile(n, inputs):                                    # tests/general_scalar_ecc_chip.rs:14-49
    ctx = Context()
    ic = IntegerContext(ctx, BLS_Q)
    sc = IntegerContext(ctx, BLS_R)
    gen = (_w(inputs, 4 * n), _w(inputs, 4 * n + 1))
    ecc = GeneralScalarEccContext(ic, sc, 4, gen, 0)
    points = [ecc.assign_point(_pt(inputs, 3 * k, 3 * k + 1, 3 * k + 2)) for k in range(n)]
    scalars = [sc.assign_w(_w(inputs, 3 * n + k)) for k in range(n)]
    r1 = (_w(inputs, 4 * n + 2), _w(inputs, 4 * n + 3))
    r2 = (_w(inputs, 4 * n + 4), _w(inputs, 4 * n + 5))
    before = (ctx.offset(), Counter(ctx.s.counts))
    res = ecc.msm_unsafe(points, scalars, r1, r2)
    ctx.s.marks = {"msm_unsafe_rows": [a - b for a, b in zip(ctx.offset(), before[0])],
                   "msm_unsafe_counts": dict(sorted((ctx.s.counts - before[1]).items()))}
    res_expect = ecc.assign_point(_pt(inputs, 4 * n + 6, 4 * n + 7, 4 * n + 8))
    ecc.ecc_assert_equal(res, res_expect)
    return ctx


def run_pairing_check_bn256(inputs):                                      # tests/native_scalar_pairing_chip.rs:67-97
    ctx = Context()
    ic = IntegerContext(ctx, BN_Q)
    ecc = NativeScalarEccContext(ic, 3, None, 254, 0)
    po = Bn256Pairing(ecc)
    bx = po.fq2_assign_constant((_w(inputs, 0), _w(inputs, 1)))
    by = po.fq2_assign_constant((_w(inputs, 2), _w(inputs, 3)))
    b = G2A(bx, by, ctx.assign_constant(0))
    neg_a = ecc.assign_point(_pt(inputs, 4, 5, 6))
    a = ecc.assign_point(_pt(inputs, 7, 8, 9))
    po.check_pairing([(a, b), (neg_a, b)])
    return ctx


def run_pairing_check_bls12_381(inputs):                                  # tests/general_scalar_pairing_chip.rs:74-105
    ctx = Context()
    ic = IntegerContext(ctx, BLS_Q)
    ecc = NativeScalarEccContext(ic, 4, None, 255, 0)                    # EccChipBaseOps of GeneralScalarEccContext: base_integer_ctx
    po = Bls12381Pairing(ecc)
    bx = po.fq2_assign_constant((_w(inputs, 0), _w(inputs, 1)))
    by = po.fq2_assign_constant((_w(inputs, 2), _w(inputs, 3)))
    b = G2A(bx, by, ctx.assign_constant(0))
    bcx = po.fq2_assign_constant((_w(inputs, 4), _w(inputs, 5)))
    bcy = po.fq2_assign_constant((_w(inputs, 6), _w(inputs, 7)))
    bc = G2A(bcx, bcy, ctx.assign_constant(0))
    neg_a = ecc.assign_point(_pt(inputs, 8, 9, 10))
    ac = ecc.assign_point(_pt(inputs, 11, 12, 13))
    po.check_pairing([(ac, b), (neg_a, bc)])
    return ctx


def run_pairing(curve, n_pairs, with_expected, inputs):
    """pairing(terms) [== expected]: tests/native_scalar_pairing_chip.rs:20-65 (bn256, 1 pair),
    tests/general_scalar_pairing_chip.rs:20-72 (bls12_381, 2 pairs); inputs as h2e_program_pairing"""
    ctx = Context()
    ic = IntegerContext(ctx, BN_Q if curve == 0 else BLS_Q)
    ecc = NativeScalarEccContext(ic, 3 if curve == 0 else 4, None, 254 if curve == 0 else 255, 0)
    po = Bn256Pairing(ecc) if curve == 0 else Bls12381Pairing(ecc)
    g2 = []
    for k in range(n_pairs):
        x = po.fq2_assign_constant((_w(inputs, 4 * k), _w(inputs, 4 * k + 1)))
        y = po.fq2_assign_constant((_w(inputs, 4 * k + 2), _w(inputs, 4 * k + 3)))
        g2.append(G2A(x, y, ctx.assign_constant(0)))
    e0 = 4 * n_pairs
    if with_expected:
        v = [(_w(inputs, e0 + 2 * i), _w(inputs, e0 + 2 * i + 1)) for i in range(6)]
        expected = po.fq12_assign_constant(((v[0], v[1], v[2]), (v[3], v[4], v[5])))
    p0 = e0 + (12 if with_expected else 0)
    g1 = [ecc.assign_point(_pt(inputs, p0 + 3 * k, p0 + 3 * k + 1, p0 + 3 * k + 2)) for k in range(n_pairs)]
    res = po.pairing(list(zip(g1, g2)))
    if with_expected:
        po.fq12_assert_eq(expected, res)
    ctx.s.marks["result"] = [ic.get_w_bn(a) % ic.info.w_modulus for f6 in res for f2 in f6 for a in f2]
    return ctx


# ===================================================================================================
# summaries: what the fixtures hold and what the tests compare
M64 = (1 << 64) - 1


def _sm64(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def digest(cells):
    """the streaming-job digest of include/h2e.h (h2e_digest) over a {row * cols + col: value} map"""
    d = [0, 0, 0, 0]
    for cell, v in cells.items():
        v %= N_MOD
        t = _sm64(cell)
        for j in range(4):
            d[j] = (d[j] + _sm64(((v >> (64 * j)) & M64) ^ t ^ ((j * 0xA24BAED4963EE407) & M64))) & M64
    return d


def summary(ctx):
    """offsets, heights, counts and digests of a finished context (everything a fixture pins)"""
    s = ctx.s
    perm = hashlib.sha256()
    for a, b in ctx.permutations:
        perm.update(a.to_bytes(4, "little") + b.to_bytes(4, "little"))
    flags = []
    for region in range(3):
        h = hashlib.sha256()
        for cell in sorted(s.adv[region]):
            row, col = divmod(cell, ADV_COLS[region])
            pc = (region << 30) | (col << 27) | row
            h.update(cell.to_bytes(8, "little") + bytes([1 | (2 if pc in s.permute else 0)]))
        flags.append(h.hexdigest())
    return {
        "offsets": [ctx.base_offset, ctx.range_offset, ctx.select_offset],
        "heights": [ctx.base_height, ctx.range_height, ctx.select_height],
        "n_advice_cells": sum(len(a) for a in s.adv),
        "n_fixed_cells": [len(f) for f in s.fix],
        "n_permutations": len(ctx.permutations),
        "permutations_sha256": perm.hexdigest(),
        "adv_digest": [digest(s.adv[r]) for r in range(3)],
        "fix_digest": [digest(s.fix[r]) for r in range(3)],
        "assigned_flags_sha256": flags,
        "counts": dict(sorted(s.counts.items())),
        "marks": s.marks,
    }
