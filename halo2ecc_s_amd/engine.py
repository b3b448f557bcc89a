"""ctypes binding of include/h2e.h + thin torch plumbing.

Mirrors the reference-side usage: build a context (`Context::new`, src/context.rs:136-143), describe
a workload with the chip API (here: one of the recorded programs), then read `Records`
(src/context.rs:294-301).  Advice values live in torch CUDA tensors; everything shape-only comes back
as numpy views of host arrays owned by the program.
"""
import ctypes as C
import os

import numpy as np

FIELD_BN256_FQ, FIELD_BLS12_381_FQ, FIELD_BLS12_381_FR = 0, 1, 2
ST_OK, ST_ASSERT_FAILED, ST_RETRY_ADD_SAME_OR_NEG_POINT, ST_RETRY_ADD_IDENTITY, ST_ARITH = 0, 1, 2, 4, 8

EXPORTED_SYMBOLS = [
    "h2e_last_error", "h2e_version", "h2e_ctx_create", "h2e_ctx_destroy", "h2e_program_int_mul_batch",
    "h2e_program_integer_chip_st", "h2e_program_msm_bn256_tile", "h2e_program_pairing_check_bn256",
    "h2e_program_pairing_check_bls12_381", "h2e_program_destroy", "h2e_program_shape", "h2e_run",
    "h2e_int_mul_batch", "h2e_msm_bn256_tile", "h2e_pairing_check_bn256", "h2e_pairing_check_bls12_381",
    "h2e_last_run_launch_ms", "h2e_set_profiling", "h2e_program_outputs", "h2e_program_launches", "h2e_export_columns",
    "h2e_program_msm_bn256_tile_no_select", "h2e_last_run_expansion_launches",
]


class H2EError(RuntimeError):
    pass


def lib_path():
    # H2E_LIB: another build of the same sources (A/B timing experiments only)
    return os.environ.get("H2E_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libh2e.so")


class _Shape(C.Structure):
    _fields_ = [
        ("field_pair", C.c_int), ("slot_words", C.c_uint32), ("n_input_slots", C.c_uint32),
        ("base_offset", C.c_uint64), ("range_offset", C.c_uint64), ("select_offset", C.c_uint64),
        ("base_height", C.c_uint64), ("range_height", C.c_uint64), ("select_height", C.c_uint64),
        ("base_rows", C.c_uint64), ("range_rows", C.c_uint64), ("select_rows", C.c_uint64),
        ("n_advice_cells", C.c_uint64), ("n_permutations", C.c_uint64), ("n_dict", C.c_uint64),
        ("n_fixed_patches", C.c_uint64), ("n_segments", C.c_uint32), ("n_ops", C.c_uint64),
        ("dict", C.c_void_p), ("base_fix", C.c_void_p), ("range_fix", C.c_void_p), ("select_fix", C.c_void_p),
        ("base_flags", C.c_void_p), ("range_flags", C.c_void_p), ("select_flags", C.c_void_p),
        ("permutations", C.c_void_p), ("fixed_patches", C.c_void_p),
    ]


_lib = None


def lib():
    """Load libh2e.so (built in-tree by halo2ecc_s_amd.build).  Fails loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise H2EError(f"{path} not found: build it with `python -m halo2ecc_s_amd.build` "
                       "(hipcc, gfx950); the witness engine has no CPU fallback")
    L = C.CDLL(path)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
    L.h2e_last_error.restype = C.c_char_p
    L.h2e_version.restype = C.c_char_p
    L.h2e_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.h2e_ctx_destroy.argtypes = [vp]
    L.h2e_ctx_destroy.restype = None
    L.h2e_program_int_mul_batch.argtypes = [i32, u32, i32, C.POINTER(vp)]
    L.h2e_program_integer_chip_st.argtypes = [i32, i32, C.POINTER(vp)]
    L.h2e_program_msm_bn256_tile.argtypes = [u32, i32, C.POINTER(vp)]
    L.h2e_program_msm_bn256_tile_no_select.argtypes = [u32, i32, C.POINTER(vp)]
    L.h2e_program_pairing_check_bn256.argtypes = [i32, C.POINTER(vp)]
    L.h2e_program_pairing_check_bls12_381.argtypes = [i32, C.POINTER(vp)]
    L.h2e_program_destroy.argtypes = [vp]
    L.h2e_program_destroy.restype = None
    L.h2e_program_shape.argtypes = [vp, C.POINTER(_Shape)]
    L.h2e_run.argtypes = [vp, vp, u32, vp, vp, vp, vp, vp, vp]
    L.h2e_int_mul_batch.argtypes = [vp, i32, u32, u32, vp, vp, vp, vp, vp, vp]
    L.h2e_msm_bn256_tile.argtypes = [vp, u32, u32, vp, vp, vp, vp, vp, vp]
    L.h2e_pairing_check_bn256.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp]
    L.h2e_pairing_check_bls12_381.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp]
    L.h2e_last_run_launch_ms.argtypes = [vp, C.POINTER(C.c_float), u32]
    L.h2e_set_profiling.argtypes = [vp, i32]
    L.h2e_last_run_expansion_launches.argtypes = [vp, C.POINTER(u32), u32]
    L.h2e_program_outputs.argtypes = [vp, C.POINTER(u32), u32]
    L.h2e_program_launches.argtypes = [vp, C.POINTER(C.c_uint64), u32]
    L.h2e_export_columns.argtypes = [vp, u32, C.c_uint64, u32, vp, vp, vp]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise H2EError(f"h2e error {rc}: {lib().h2e_last_error().decode()}")


def _view(ptr, count, dtype):
    if not ptr or count == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype)


class Program:
    """Recorded shape of one workload (host only; usable without a GPU)."""

    def __init__(self, handle):
        self._h = handle
        s = _Shape()
        _check(lib().h2e_program_shape(self._h, C.byref(s)))
        self.shape = s
        for name, _ in _Shape._fields_[:18]:
            setattr(self, name, int(getattr(s, name)))

    @staticmethod
    def _make(fn, *args):
        h = C.c_void_p()
        _check(fn(*args, C.byref(h)))
        return Program(h)

    @classmethod
    def int_mul_batch(cls, field_pair, n, emit_shape=True):
        return cls._make(lib().h2e_program_int_mul_batch, field_pair, n, int(emit_shape))

    @classmethod
    def integer_chip_st(cls, field_pair, emit_shape=True):
        return cls._make(lib().h2e_program_integer_chip_st, field_pair, int(emit_shape))

    @classmethod
    def msm_bn256_tile(cls, n_points, emit_shape=True, with_select=True):
        f = lib().h2e_program_msm_bn256_tile if with_select else lib().h2e_program_msm_bn256_tile_no_select
        return cls._make(f, n_points, int(emit_shape))

    @classmethod
    def pairing_check_bn256(cls, emit_shape=True):
        return cls._make(lib().h2e_program_pairing_check_bn256, int(emit_shape))

    @classmethod
    def pairing_check_bls12_381(cls, emit_shape=True):
        return cls._make(lib().h2e_program_pairing_check_bls12_381, int(emit_shape))

    # ---- shape artefacts (numpy views; valid while the program is alive) ----
    def fixed_dict(self):
        return _view(self.shape.dict, self.n_dict * 4, np.uint64).reshape(-1, 4)

    def base_fix(self):
        return _view(self.shape.base_fix, self.base_rows * 9, np.uint32).reshape(-1, 9)

    def range_fix(self):
        return _view(self.shape.range_fix, self.range_rows * 2, np.uint32).reshape(-1, 2)

    def select_fix(self):
        return _view(self.shape.select_fix, self.select_rows * 2, np.uint32).reshape(-1, 2)

    def base_flags(self):
        return _view(self.shape.base_flags, self.base_rows * 5, np.uint8).reshape(-1, 5)

    def range_flags(self):
        return _view(self.shape.range_flags, self.range_rows * 3, np.uint8).reshape(-1, 3)

    def select_flags(self):
        return _view(self.shape.select_flags, self.select_rows * 2, np.uint8).reshape(-1, 2)

    def permutations(self):
        return _view(self.shape.permutations, self.n_permutations * 2, np.uint32).reshape(-1, 2)

    def fixed_patches(self):
        return _view(self.shape.fixed_patches, self.n_fixed_patches * 4, np.uint32).reshape(-1, 4)

    def outputs(self):
        buf = (C.c_uint32 * 64)()
        n = lib().h2e_program_outputs(self._h, buf, 64)
        return [int(buf[i]) for i in range(n)]

    def launches(self):
        """per engine launch: dict(n_strands, n_ops, cells, dbase, drange, dselect, n_params, base0)"""
        cap = 256
        buf = (C.c_uint64 * (8 * cap))()
        n = lib().h2e_program_launches(self._h, buf, cap)
        keys = ("n_strands", "n_ops", "cells", "dbase", "drange", "dselect", "n_params", "base0")
        return [dict(zip(keys, [int(buf[8 * i + j]) for j in range(8)])) for i in range(n)]

    def close(self):
        if self._h:
            lib().h2e_program_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One per GPU (h2e_ctx).  Advice arrays are torch CUDA tensors of uint64 words."""

    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise H2EError("no GPU visible: the witness engine has no CPU fallback")
        self.torch = torch
        self.device = device
        h = C.c_void_p()
        _check(lib().h2e_ctx_create(device, C.byref(h)))
        self._h = h

    def alloc(self, program, n_instances):
        t = self.torch
        dev = f"cuda:{self.device}"
        kw = dict(dtype=t.int64, device=dev)
        base = t.zeros((n_instances, program.base_rows, 5, 4), **kw)
        rng = t.zeros((n_instances, program.range_rows, 3, 4), **kw)
        sel = t.zeros((n_instances, program.select_rows, 2, 4), **kw)
        status = t.zeros((n_instances,), dtype=t.int32, device=dev)
        return base, rng, sel, status

    def upload_inputs(self, program, inputs):
        """inputs: numpy uint64 [n_instances][n_input_slots][slot_words]"""
        t = self.torch
        a = np.ascontiguousarray(inputs, dtype=np.uint64)
        assert a.shape[1:] == (program.n_input_slots, program.slot_words), (a.shape, program.n_input_slots)
        return t.from_numpy(a.view(np.int64)).to(f"cuda:{self.device}")

    def run(self, program, d_inputs, base, rng, sel, status, stream=None):
        t = self.torch
        n = d_inputs.shape[0]
        s = stream if stream is not None else t.cuda.current_stream(self.device)
        _check(lib().h2e_run(self._h, program._h, n, d_inputs.data_ptr(), base.data_ptr(), rng.data_ptr(),
                             sel.data_ptr(), status.data_ptr(), s.cuda_stream))

    def export_columns(self, rows_major, stream=None):
        """row-major advice tensor [instances][rows][cols][4] -> column-major [instances][cols][rows][4] on the device
        (h2e_export_columns: the halo2 side keeps one array per advice column)"""
        t = self.torch
        n, rows, cols, w = rows_major.shape
        assert w == 4 and rows_major.is_contiguous()
        out = t.empty((n, cols, rows, 4), dtype=rows_major.dtype, device=rows_major.device)
        s = stream if stream is not None else t.cuda.current_stream(self.device)
        _check(lib().h2e_export_columns(self._h, n, rows, cols, rows_major.data_ptr(), out.data_ptr(), s.cuda_stream))
        return out

    def set_profiling(self, on):
        _check(lib().h2e_set_profiling(self._h, int(on)))

    def last_run_launch_ms(self, cap=128):
        """per launched segment: (value-chain ms, expansion ms), from HIP events on the launching streams"""
        buf = (C.c_float * cap)()
        n = lib().h2e_last_run_launch_ms(self._h, buf, cap)
        if n < 0:
            _check(n)
        return [(buf[2 * i], buf[2 * i + 1]) for i in range(min(n, cap // 2))]

    def last_run_expansion_launches(self, cap=64):
        """per launched segment: kernel launches its full expansion went out as (2 = split, see h2e.h)"""
        buf = (C.c_uint32 * cap)()
        n = lib().h2e_last_run_expansion_launches(self._h, buf, cap)
        if n < 0:
            _check(n)
        return [int(buf[i]) for i in range(min(n, cap))]

    def close(self):
        if self._h:
            lib().h2e_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
