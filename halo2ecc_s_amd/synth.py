"""Seeded synthetic inputs for the hot-path workloads (pure Python integers; no engine, no oracle).

PRNG = SplitMix64 seeded 0x68326563632d73 + config index; field elements by rejection sampling; points
are multiples of the generator.  The reference's tests draw the same kinds of inputs
from time-seeded / OS randomness (src/tests/mod.rs:34-42, src/tests/native_scalar_pairing_chip.rs:26-27).
"""
import numpy as np

SEED0 = 0x68326563632D73
BN_Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
BN_R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
BLS_Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
BLS_R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
W_MODULUS = {0: BN_Q, 1: BLS_Q, 2: BLS_R}
SLOT_WORDS = {0: 4, 1: 6, 2: 4}
M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & M64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)

    def below(self, m):
        bits = m.bit_length()
        words = (bits + 63) // 64
        while True:
            x = 0
            for i in range(words):
                x |= self.next() << (64 * i)
            x &= (1 << bits) - 1
            if x < m:
                return x


def words(x, n):
    return [(x >> (64 * i)) & M64 for i in range(n)]


def pack(values, slot_words):
    a = np.zeros((len(values), slot_words), dtype=np.uint64)
    for i, v in enumerate(values):
        a[i, :] = words(int(v), slot_words)
    return a


# ---- generic short Weierstrass arithmetic over Fp or Fp2 (affine, None = identity) ----------------
class Fp2:
    __slots__ = ("a", "b", "p")

    def __init__(self, a, b, p):
        self.a, self.b, self.p = a % p, b % p, p

    def __add__(self, o):
        return Fp2(self.a + o.a, self.b + o.b, self.p)

    def __sub__(self, o):
        return Fp2(self.a - o.a, self.b - o.b, self.p)

    def __neg__(self):
        return Fp2(-self.a, -self.b, self.p)

    def __mul__(self, o):
        if isinstance(o, int):
            return Fp2(self.a * o, self.b * o, self.p)
        return Fp2(self.a * o.a - self.b * o.b, self.a * o.b + self.b * o.a, self.p)

    def __eq__(self, o):
        return self.a == o.a and self.b == o.b

    def inv(self):
        t = pow(self.a * self.a + self.b * self.b, -1, self.p)
        return Fp2(self.a * t, -self.b * t, self.p)

    def is_zero(self):
        return self.a == 0 and self.b == 0


class Fp1:
    __slots__ = ("a", "p")

    def __init__(self, a, p):
        self.a, self.p = a % p, p

    def __add__(self, o):
        return Fp1(self.a + o.a, self.p)

    def __sub__(self, o):
        return Fp1(self.a - o.a, self.p)

    def __neg__(self):
        return Fp1(-self.a, self.p)

    def __mul__(self, o):
        if isinstance(o, int):
            return Fp1(self.a * o, self.p)
        return Fp1(self.a * o.a, self.p)

    def __eq__(self, o):
        return self.a == o.a

    def inv(self):
        return Fp1(pow(self.a, -1, self.p), self.p)

    def is_zero(self):
        return self.a == 0


def ec_add(P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2).is_zero():
            return None
        lam = (x1 * x1 * 3) * (y1 * 2).inv()
    else:
        lam = (y2 - y1) * (x2 - x1).inv()
    x3 = lam * lam - x1 - x2
    y3 = lam * (x1 - x3) - y1
    return (x3, y3)


def ec_neg(P):
    return None if P is None else (P[0], -P[1])


def ec_mul(P, k):
    R = None
    while k:
        if k & 1:
            R = ec_add(R, P)
        P = ec_add(P, P)
        k >>= 1
    return R


def bn_g1_gen():
    return (Fp1(1, BN_Q), Fp1(2, BN_Q))


def bn_g2_gen():
    return (Fp2(10857046999023057135944570762232829481370756359578518086990519993285655852781,
                11559732032986387107991004021392285783925812861821192530917403151452391805634, BN_Q),
            Fp2(8495653923123431417604973247489272438418190587263600148770280649306958101930,
                4082367875863433681332203403145435568316851327593401208105741076214120093531, BN_Q))


def bls_g1_gen():
    return (Fp1(0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB, BLS_Q),
            Fp1(0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1, BLS_Q))


def bls_g2_gen():
    return (Fp2(0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
                0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E, BLS_Q),
            Fp2(0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
                0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE, BLS_Q))


# ---- workloads ---------------------------------------------------------------------------------------
def int_mul_batch_inputs(field_pair, n, seed_index=1):
    rng = SplitMix64(SEED0 + seed_index)
    w = W_MODULUS[field_pair]
    vals = [rng.below(w) for _ in range(2 * n)]
    return pack(vals, SLOT_WORDS[field_pair])


def integer_chip_st_inputs(field_pair, seed_index=1, b_zero=False):
    """a, b, a+b, a-b, a*b, a/b  (src/tests/integer_chip.rs:16-31)"""
    rng = SplitMix64(SEED0 + seed_index)
    w = W_MODULUS[field_pair]
    a, b = rng.below(w), rng.below(w)
    while b == 0:
        b = rng.below(w)
    vals = [a, b, (a + b) % w, (a - b) % w, (a * b) % w, (a * pow(b, -1, w)) % w]
    return pack(vals, SLOT_WORDS[field_pair])


def msm_bn256_tile_inputs(n, seed_index=2, tile=0, cheap_points=False, with_expected=True, identity_at=()):
    """Input vector of h2e_program_msm_bn256_tile: (x, y, z) x n, n scalars, G, r1, r2, expected (x, y, z).

    cheap_points: P_i = P_0 + i*D instead of n independent scalar multiplications (bench-sized tiles).
    identity_at: indices whose point is the identity (z = 1, x = y = 0).
    Returns (inputs[4n+9][4], expected_point or None)."""
    return _msm_tile_inputs(bn_g1_gen(), BN_R, 4, n, seed_index, tile, cheap_points, with_expected, identity_at)


def msm_bls12_381_tile_inputs(n, seed_index=8, tile=0, cheap_points=False, with_expected=True, identity_at=()):
    """the same for h2e_program_msm_bls12_381_tile (general-scalar MSM: bls12_381 G1 points, scalars < bls12_381 r; 6-word slots)"""
    return _msm_tile_inputs(bls_g1_gen(), BLS_R, 6, n, seed_index, tile, cheap_points, with_expected, identity_at)


def _msm_tile_inputs(G, R, slot_words, n, seed_index, tile, cheap_points, with_expected, identity_at):
    rng = SplitMix64(SEED0 + seed_index + 1000003 * tile)
    pts = []
    if cheap_points:
        P = ec_mul(G, rng.below(R))
        D = ec_mul(G, rng.below(R))
        for _ in range(n):
            pts.append(P)
            P = ec_add(P, D)
    else:
        for _ in range(n):
            pts.append(ec_mul(G, rng.below(R)))
    for i in identity_at:
        pts[i] = None
    scalars = [rng.below(R) for _ in range(n)]
    r1 = ec_mul(G, rng.below(R))
    r2 = ec_mul(G, rng.below(R))
    vals = []
    for P in pts:
        vals += [0, 0, 1] if P is None else [P[0].a, P[1].a, 0]
    vals += scalars
    vals += [G[0].a, G[1].a, r1[0].a, r1[1].a, r2[0].a, r2[1].a]
    expected = None
    if with_expected:
        acc = None
        for P, s in zip(pts, scalars):
            if P is not None:
                acc = ec_add(acc, ec_mul(P, s))
        expected = acc
        vals += [0, 0, 1] if acc is None else [acc[0].a, acc[1].a, 0]
    else:
        vals += [G[0].a, G[1].a, 0]  # placeholder: the final ecc_assert_equal will flag ASSERT_FAILED
    return pack(vals, slot_words), expected


def pairing_check_bn256_inputs(seed_index=4, instance=0):
    """b (G2, constants), -a, a  for check_pairing([(a, b), (-a, b)])"""
    rng = SplitMix64(SEED0 + seed_index + 1000003 * instance)
    a = ec_mul(bn_g1_gen(), rng.below(BN_R))
    b = ec_mul(bn_g2_gen(), rng.below(BN_R))
    na = ec_neg(a)
    vals = [b[0].a, b[0].b, b[1].a, b[1].b, na[0].a, na[1].a, 0, a[0].a, a[1].a, 0]
    return pack(vals, 4)


def pairing_check_bls12_381_inputs(seed_index=5, instance=0):
    """b, bc (G2, constants), -a, ac  for check_pairing([(ac, b), (-a, bc)])"""
    rng = SplitMix64(SEED0 + seed_index + 1000003 * instance)
    a = ec_mul(bls_g1_gen(), rng.below(BLS_R))
    b = ec_mul(bls_g2_gen(), rng.below(BLS_R))
    c = rng.below(BLS_R)
    ac = ec_mul(a, c)
    bc = ec_mul(b, c)
    na = ec_neg(a)
    vals = [b[0].a, b[0].b, b[1].a, b[1].b, bc[0].a, bc[0].b, bc[1].a, bc[1].b, na[0].a, na[1].a, 0, ac[0].a, ac[1].a, 0]
    return pack(vals, 6)


def pairing_inputs(curve, n_pairs, seed_index=6, instance=0, expected=None):
    """inputs of h2e_program_pairing: per pair the G2 point (constants), [the expected Fq12 value: 12 W values], per pair the
    G1 point.  curve 0 = bn256, 1 = bls12_381.  `expected` = list of 12 ints (the native pairing value, in the
    reference's tests computed by the curve library: src/tests/native_scalar_pairing_chip.rs:29, general_...:30-33)"""
    rng = SplitMix64(SEED0 + seed_index + 1000003 * instance + 7919 * curve)
    g1, g2, r, sw = (bn_g1_gen(), bn_g2_gen(), BN_R, 4) if curve == 0 else (bls_g1_gen(), bls_g2_gen(), BLS_R, 6)
    a = [ec_mul(g1, rng.below(r)) for _ in range(n_pairs)]
    b = [ec_mul(g2, rng.below(r)) for _ in range(n_pairs)]
    vals = []
    for q in b:
        vals += [q[0].a, q[0].b, q[1].a, q[1].b]
    if expected is not None:
        assert len(expected) == 12
        vals += list(expected)
    for p in a:
        vals += [p[0].a, p[1].a, 0]
    return pack(vals, sw)


def ops_ecc_surface_inputs(seed_index=9, instance=0):
    """inputs of the operator-API scenario over the complete-addition surface (tests/test_ops_gpu.py, oracle_run_ops_ecc_surface):
    P (x, y, z), Q (x, y, z), scalar, index (= 1), generator (x, y), r1 (x, y), r2 (x, y)"""
    rng = SplitMix64(SEED0 + seed_index + 1000003 * instance)
    G = bn_g1_gen()
    P, Q = ec_mul(G, rng.below(BN_R)), ec_mul(G, rng.below(BN_R))
    s = rng.below(BN_R)
    r1, r2 = ec_mul(G, rng.below(BN_R)), ec_mul(G, rng.below(BN_R))
    return pack([P[0].a, P[1].a, 0, Q[0].a, Q[1].a, 0, s, 1, G[0].a, G[1].a, r1[0].a, r1[1].a, r2[0].a, r2[1].a], 4)
