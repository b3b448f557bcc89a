"""The engine's modular inversion (division steps in 62-bit batches, csrc/modinv62.h) compiled for the host and held
against Python's pow(x, -1, p) over the four prime fields the engine inverts in, edge values included."""
import ctypes
import os
import random
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
BN_FQ = 21888242871839275222246405745257275088696311157297823662689037894645226208583
BN_FR = 21888242871839275222246405745257275088548364400416034343698204186575808495617
BLS_FQ = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
BLS_FR = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(HERE, "_build", "modinv_host.so")
    src = os.path.join(HERE, "modinv_host.cpp")
    hdr = os.path.join(HERE, "..", "halo2ecc_s_amd", "csrc", "modinv62.h")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", src, "-o", so])
    return ctypes.CDLL(so)


def _words(x, n):
    return [(x >> (64 * i)) & (2**64 - 1) for i in range(n)]


@pytest.mark.parametrize("form", ["registers", "memory"])   # memory: the state streamed through caller-provided memory (the digit chain's loader wave: LDS)
@pytest.mark.parametrize("p,n", [(BN_FQ, 4), (BN_FR, 4), (BLS_FR, 4), (BLS_FQ, 6)])
def test_modinv(lib, p, n, form):
    rnd = random.Random(p & 0xffff)
    vals = [0, 1, 2, 3, p - 1, p - 2, (p + 1) // 2, 2**62, 2**62 - 1, 2**124, 2**(64 * n - 3) % p, (1 << (p.bit_length() - 1))]
    vals += [rnd.randrange(p) for _ in range(3000)]
    vals += [rnd.randrange(1 << k) % p for k in (1, 8, 61, 62, 63, 64, 65, 127, 190) for _ in range(20)]
    a = np.array([_words(v, n) for v in vals], dtype=np.uint64)
    out = np.zeros_like(a)
    pw = np.array(_words(p, n), dtype=np.uint64)
    fn = getattr(lib, f"modinv_{n}" if form == "registers" else f"modinv_mem_{n}")
    fn(a.ctypes.data_as(ctypes.c_void_p), pw.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), len(vals))
    for v, o in zip(vals, out):
        got = sum(int(w) << (64 * i) for i, w in enumerate(o))
        want = pow(v, -1, p) if v else 0
        assert got == want, (hex(v), hex(got), hex(want))
