"""ctypes wrapper of oracle/liboracle.so — the CPU restatement used as the parity checker.
TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")


class Info(C.Structure):
    _fields_ = [("base_offset", C.c_uint64), ("range_offset", C.c_uint64), ("select_offset", C.c_uint64),
                ("base_height", C.c_uint64), ("range_height", C.c_uint64), ("select_height", C.c_uint64),
                ("n_permutations", C.c_uint64), ("n_advice_cells", C.c_uint64), ("status", C.c_int32),
                ("seconds", C.c_double)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".hpp", ".cpp"))]
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(LIB)
    vp = C.c_void_p
    for name in ("oracle_run_int_mul_batch", "oracle_run_integer_chip_st", "oracle_run_msm_bn256_tile",
                 "oracle_run_msm_bn256_tile_no_select", "oracle_run_pairing_check_bn256", "oracle_run_pairing_check_bls12_381", "oracle_run_pairing", "oracle_run_msm_bls12_381_tile", "oracle_run_ops_msm_twice", "oracle_run_ops_ecc_surface", "oracle_run_ops_int_tower"):
        getattr(L, name).restype = vp
    L.oracle_run_int_mul_batch.argtypes = [C.c_int, C.c_uint32, vp]
    L.oracle_run_integer_chip_st.argtypes = [C.c_int, vp]
    L.oracle_run_msm_bn256_tile.argtypes = [C.c_uint32, vp, C.c_int]
    L.oracle_run_msm_bn256_tile_no_select.argtypes = [C.c_uint32, vp, C.c_int]
    L.oracle_run_msm_bls12_381_tile.argtypes = [C.c_uint32, vp]
    L.oracle_run_ops_msm_twice.argtypes = [C.c_uint32, vp]
    L.oracle_run_ops_ecc_surface.argtypes = [vp]
    L.oracle_run_ops_int_tower.argtypes = [C.c_int, vp]
    L.oracle_run_pairing.argtypes = [C.c_int, C.c_uint32, C.c_int, vp]
    L.oracle_run_pairing_check_bn256.argtypes = [vp]
    L.oracle_run_pairing_check_bls12_381.argtypes = [vp]
    L.oracle_info.argtypes = [vp, C.POINTER(Info)]
    L.oracle_error.argtypes = [vp]
    L.oracle_error.restype = C.c_char_p
    L.oracle_export_adv.argtypes = [vp, C.c_int, vp, vp, C.c_uint64]
    L.oracle_digest.argtypes = [vp, C.c_int, vp]
    L.oracle_digest.restype = None
    L.oracle_stream_digest.argtypes = [vp, C.c_int, vp]
    L.oracle_stream_digest.restype = None
    L.oracle_export_fix.argtypes = [vp, C.c_int, vp, vp, C.c_uint64]
    L.oracle_export_permutations.argtypes = [vp, vp]
    L.oracle_check.argtypes = [vp, C.c_char_p, C.c_int]
    L.oracle_check_counts.argtypes = [vp, vp]
    L.oracle_corrupt_adv.argtypes = [vp, C.c_int, C.c_uint64, C.c_int]
    L.oracle_free.argtypes = [vp]
    L.oracle_free.restype = None
    _lib = L
    return L


class Run:
    """One oracle run: Records of the CPU restatement for a given input vector."""

    def __init__(self, handle):
        self.L = load()
        self.h = handle
        self.info = Info()
        self.L.oracle_info(self.h, C.byref(self.info))

    @property
    def error(self):
        return self.L.oracle_error(self.h).decode()

    def adv(self, region, rows):
        cols = (5, 3, 2)[region]
        out = np.zeros((rows, cols, 4), dtype=np.uint64)
        flags = np.zeros((rows, cols), dtype=np.uint8)
        self.L.oracle_export_adv(self.h, region, out.ctypes.data, flags.ctypes.data, rows)
        return out, flags

    def fix(self, region, rows):
        cols = (9, 2, 2)[region]
        out = np.zeros((rows, cols, 4), dtype=np.uint64)
        present = np.zeros((rows, cols), dtype=np.uint8)
        self.L.oracle_export_fix(self.h, region, out.ctypes.data, present.ctypes.data, rows)
        return out, present

    def digest(self, region):
        """32-byte streaming-job digest of one advice array (definition: include/h2e.h, h2e_digest)"""
        out = np.zeros(4, dtype=np.uint64)
        self.L.oracle_digest(self.h, region, out.ctypes.data)
        return out

    def stream_digest(self, region):
        """32-byte stream digest of one advice array (definition: include/h2e.h, h2e_run_digest: what the expansion accumulates)"""
        out = np.zeros(4, dtype=np.uint64)
        self.L.oracle_stream_digest(self.h, region, out.ctypes.data)
        return out

    def permutations(self):
        out = np.zeros((self.info.n_permutations, 2), dtype=np.uint32)
        if self.info.n_permutations:
            self.L.oracle_export_permutations(self.h, out.ctypes.data)
        return out

    def check(self):
        buf = C.create_string_buffer(512)
        rc = self.L.oracle_check(self.h, buf, 512)
        return rc == 0, buf.value.decode()

    def check_counts(self):
        """failing rows per class: base gate, range gates, range lookups, select lookup, copy constraints"""
        out = np.zeros(5, dtype=np.uint64)
        self.L.oracle_check_counts(self.h, out.ctypes.data)
        return out

    def corrupt(self, region, row, col):
        self.L.oracle_corrupt_adv(self.h, region, row, col)

    def close(self):
        if self.h:
            self.L.oracle_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _ptr(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data


def run_int_mul_batch(fp, n, inputs):
    a, p = _ptr(inputs)
    return Run(load().oracle_run_int_mul_batch(fp, n, p))


def run_integer_chip_st(fp, inputs):
    a, p = _ptr(inputs)
    return Run(load().oracle_run_integer_chip_st(fp, p))


def run_msm_bn256_tile(n, inputs, threads=1, with_select=True):
    a, p = _ptr(inputs)
    f = load().oracle_run_msm_bn256_tile if with_select else load().oracle_run_msm_bn256_tile_no_select
    return Run(f(n, p, threads))


def run_pairing_check_bn256(inputs):
    a, p = _ptr(inputs)
    return Run(load().oracle_run_pairing_check_bn256(p))


def run_pairing_check_bls12_381(inputs):
    a, p = _ptr(inputs)
    return Run(load().oracle_run_pairing_check_bls12_381(p))


def run_pairing(curve, n_pairs, with_expected, inputs):
    a, p = _ptr(inputs)
    return Run(load().oracle_run_pairing(curve, n_pairs, int(with_expected), p))


def run_msm_bls12_381_tile(n, inputs):
    a, p = _ptr(inputs)
    return Run(load().oracle_run_msm_bls12_381_tile(n, p))


def run_ops_msm_twice(n, inputs):
    a, p = _ptr(inputs)
    return Run(load().oracle_run_ops_msm_twice(n, p))


def run_ops_ecc_surface(inputs):
    a, p = _ptr(inputs)
    return Run(load().oracle_run_ops_ecc_surface(p))


def run_ops_int_tower(curve, inputs):
    a, p = _ptr(inputs)
    return Run(load().oracle_run_ops_int_tower(curve, p))
