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
ST_TEST_HOOK = 0x80
LAYOUT_ROWS, LAYOUT_COLUMNS = 0, 1
FORM_CANONICAL, FORM_MONTGOMERY = 0, 1
OPT_X_SPLIT_PCT, OPT_X_SPLIT_MIN_LANES, OPT_TEST_SKIP_EXPANSION, OPT_PIPELINE_DEPTH, OPT_TEST_SCAN_FALLBACK = 1, 2, 3, 4, 5
STAT_LAST_SPLIT_SEGMENTS, STAT_RUNS, STAT_PIPELINE_DEPTH, STAT_MAX_PIPELINE_DEPTH, STAT_SCAN_FALLBACKS = 1, 2, 3, 4, 5
OPT_PREFAULT_HBM = 6
OPT_OP_CACHE_CAP = 7
OPT_OFF = -(1 << 63)
COLS = (5, 3, 2)
CHECK_BASE_GATE, CHECK_RANGE_GATE, CHECK_RANGE_LOOKUP, CHECK_SELECT_LOOKUP, CHECK_COPY = range(5)
CHECK_CLASSES = 5

EXPORTED_SYMBOLS = [
    "h2e_last_error", "h2e_version", "h2e_ctx_create", "h2e_ctx_destroy", "h2e_program_int_mul_batch",
    "h2e_program_integer_chip_st", "h2e_program_msm_bn256_tile", "h2e_program_pairing_check_bn256",
    "h2e_program_pairing_check_bls12_381", "h2e_program_destroy", "h2e_program_shape", "h2e_run",
    "h2e_int_mul_batch", "h2e_msm_bn256_tile", "h2e_pairing_check_bn256", "h2e_pairing_check_bls12_381",
    "h2e_last_run_launch_ms", "h2e_set_profiling", "h2e_program_outputs", "h2e_program_launches", "h2e_export",
    "h2e_submit", "h2e_wait", "h2e_job_launch_ms", "h2e_digest", "h2e_program_pairing", "h2e_program_msm_bls12_381_tile",
    "h2e_records_create", "h2e_records_destroy", "h2e_records_arrays", "h2e_records_shape", "h2e_op_assign_w", "h2e_op_assign",
    "h2e_op_int", "h2e_op_assign_points", "h2e_op_assign_scalars", "h2e_op_msm_unsafe", "h2e_op_ecc_assert_equal",
    "h2e_op_assign_g2_constant", "h2e_op_check_pairing", "h2e_op_to_point_with_curvature", "h2e_op_ecc_reduce_with_curvature",
    "h2e_op_ecc_double", "h2e_op_ecc_add", "h2e_op_ecc_neg", "h2e_op_ecc_encode", "h2e_op_ecc_mul", "h2e_op_assign_constant_point",
    "h2e_op_bisec_point_with_curvature", "h2e_op_assign_cache_point", "h2e_op_assign_selected_point", "h2e_export_fixed", "h2e_range_table", "h2e_export_copy_constraints", "h2e_ctx_set_option", "h2e_ctx_get_stat",
    "h2e_program_msm_bn256_tile_no_select", "h2e_last_run_expansion_launches",
    "h2e_run_digest", "h2e_submit_digest", "h2e_records_attach", "h2e_op_int_mul_small_constant", "h2e_op_assign_int_constant", "h2e_op_bisec_int", "h2e_op_fq", "h2e_op_pairing",
    "h2e_check", "h2e_program_tape_opcodes", "h2e_program_value_chain_kind", "h2e_program_pack_order",
    "h2e_program_launch_rows", "h2e_unit_records", "h2e_unit_record_words", "h2e_selftest_digit_rows", "h2e_last_warning", "h2e_run_batches", "h2e_submit_batches",
    "h2e_ring_create", "h2e_ring_destroy", "h2e_ring_arrays", "h2e_ring_info", "h2e_ring_submit", "h2e_ring_submit_digest", "h2e_run_columns", "h2e_ring_release",
]


class H2EError(RuntimeError):
    pass


class _DevArray:
    """a device allocation torch did not make (h2e_ring's mapped array sets), as torch.as_tensor takes it"""

    def __init__(self, ptr, shape, keep):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<i8", "data": (int(ptr), False), "version": 3, "strides": None}
        self._keep = keep


class Ring:
    """h2e_ring (include/h2e.h): output arrays for `depth` runs in flight in less than `depth` array sets - the rows of the program's
    biggest launch are shared by runs k and k + 2.  Runs are submitted in order: submit(0, ...), submit(1, ...), ..."""

    def __init__(self, engine, program, n_instances, depth):
        self.engine, self.program, self.n, self.depth = engine, program, n_instances, depth
        h = C.c_void_p()
        _check(lib().h2e_ring_create(engine._h, program._h, n_instances, depth, C.byref(h)))
        self._h = h
        buf = (C.c_uint64 * 10)()
        _check(min(0, lib().h2e_ring_info(self._h, buf, 10)))
        v = [int(x) for x in buf]
        self.info = {"set_bytes": v[0:3], "shared_bytes": v[3:6], "physical_bytes": v[6], "shared_launch": v[7], "virtual_sets": v[8], "depth": v[9]}
        self._arrays = {}

    def arrays(self, k):
        """run k's (base, range, select) as tensors [rows][cols][half][instance][2 words]"""
        v = k % self.info["virtual_sets"]
        if v not in self._arrays:
            ptrs = [C.c_void_p() for _ in range(3)]
            _check(lib().h2e_ring_arrays(self._h, k, *[C.byref(p) for p in ptrs]))
            t = self.engine.torch
            out = []
            for ptr, rows, cols in zip(ptrs, (self.program.base_rows, self.program.range_rows, self.program.select_rows), COLS):
                out.append(t.as_tensor(_DevArray(ptr.value, (max(1, rows), cols, 2, self.n, 2), self), device=f"cuda:{self.engine.device}")[:rows])
            self._arrays[v] = tuple(out)
        return self._arrays[v]

    def submit(self, k, d_inputs, status, digests=None, stream=None):
        assert d_inputs.shape[0] == self.n and status.shape[0] == self.n
        job = C.c_int(-1)
        st = self.engine._stream(stream).cuda_stream
        if digests is None:
            _check(lib().h2e_ring_submit(self._h, k, d_inputs.data_ptr(), status.data_ptr(), st, C.byref(job)))
        else:
            assert tuple(digests.shape) == (3, self.n, 4) and digests.is_contiguous()
            _check(lib().h2e_ring_submit_digest(self._h, k, d_inputs.data_ptr(), status.data_ptr(), digests.data_ptr(), st, C.byref(job)))
        return job.value

    def release(self, k, stream=None):
        """h2e_ring_release: the consumer's reads of run k end at this point of `stream` (run k + 2 waits for it)"""
        _check(lib().h2e_ring_release(self._h, k, self.engine._stream(stream).cuda_stream))

    def close(self):
        if self._h:
            self._arrays = {}
            lib().h2e_ring_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass


def lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libh2e.so")


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


class HInt(C.Structure):      # h2e_int: AssignedInteger (src/assign.rs:31-37) as cell references + times
    _fields_ = [("limbs", C.c_uint32 * 4), ("native", C.c_uint32), ("times", C.c_uint32)]


class HPoint(C.Structure):    # h2e_point: AssignedPoint
    _fields_ = [("x", HInt), ("y", HInt), ("z", C.c_uint32)]


class HPointC(C.Structure):   # h2e_point_c: AssignedPointWithCurvature
    _fields_ = [("p", HPoint), ("cv", HInt), ("cz", C.c_uint32)]


class HG2(C.Structure):       # h2e_g2: AssignedG2Affine
    _fields_ = [("x0", HInt), ("x1", HInt), ("y0", HInt), ("y1", HInt), ("z", C.c_uint32)]


INT_ADD, INT_SUB, INT_MUL, INT_DIV, INT_REDUCE = 0, 1, 2, 3, 4
INT_NEG, INT_SQUARE, INT_UNSAFE_INVERT, INT_IS_ZERO, INT_IS_EQUAL, INT_ASSERT_EQUAL = 5, 6, 7, 8, 9, 10
(FQ_ADD, FQ_SUB, FQ_MUL, FQ_SQUARE, FQ_NEG, FQ_DOUBLE, FQ_CONJUGATE, FQ_UNSAFE_INVERT, FQ_MUL_BY_NONRESIDUE, FQ_FROBENIUS_MAP,
 FQ_CYCLOTOMIC_SQUARE, FQ_REDUCE, FQ_ASSERT_EQUAL) = range(13)
STAT_OP_CACHE_HITS, STAT_OP_CACHE_MISSES, STAT_OP_CACHE_EVICTIONS, STAT_OP_CACHE_SIZE = 6, 7, 8, 9
STAT_HW_QUEUES, STAT_HW_QUEUES_WANTED = 10, 11

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
    # torch first: it brings its own HIP runtime, and that has to be the one the process binds - with libh2e.so (linked against the
    # system's libamdhip64) loaded before it, h2e_ctx_create found no device (round 6: smoke() touched the library before Engine())
    import torch  # noqa: F401
    L = C.CDLL(path)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
    L.h2e_last_error.restype = C.c_char_p
    L.h2e_last_warning.restype = C.c_char_p
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
    L.h2e_program_msm_bls12_381_tile.argtypes = [u32, i32, C.POINTER(vp)]
    L.h2e_export_fixed.argtypes = [vp, vp, i32, i32, i32, u32, vp, vp, vp]
    L.h2e_range_table.argtypes = [vp, i32, vp, vp]
    L.h2e_export_copy_constraints.argtypes = [vp, vp, vp, vp]
    L.h2e_records_create.argtypes = [vp, i32, i32, u32, C.c_uint64, C.c_uint64, C.c_uint64, i32, C.POINTER(vp)]
    L.h2e_records_attach.argtypes = [vp, i32, i32, u32, vp, vp, vp, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint64, i32, C.POINTER(vp)]
    L.h2e_op_int_mul_small_constant.argtypes = [vp, C.POINTER(HInt), C.c_uint64, C.POINTER(HInt), vp]
    L.h2e_op_assign_int_constant.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(HInt), vp]
    L.h2e_op_bisec_int.argtypes = [vp, u32, C.POINTER(HInt), C.POINTER(HInt), C.POINTER(HInt), vp]
    L.h2e_op_fq.argtypes = [vp, i32, i32, C.POINTER(HInt), C.POINTER(HInt), C.c_uint64, C.POINTER(HInt), vp]
    L.h2e_op_pairing.argtypes = [vp, u32, C.POINTER(HPoint), C.POINTER(HG2), C.POINTER(HInt), vp]
    L.h2e_records_destroy.argtypes = [vp]
    L.h2e_records_destroy.restype = None
    L.h2e_records_arrays.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.h2e_records_shape.argtypes = [vp, C.POINTER(_Shape)]
    L.h2e_op_assign_w.argtypes = [vp, vp, C.POINTER(HInt), vp]
    L.h2e_op_assign.argtypes = [vp, vp, C.POINTER(u32), vp]
    L.h2e_op_int.argtypes = [vp, i32, C.POINTER(HInt), C.POINTER(HInt), C.POINTER(HInt), C.POINTER(u32), vp]
    L.h2e_op_assign_points.argtypes = [vp, u32, vp, C.POINTER(HPoint), vp]
    L.h2e_op_assign_scalars.argtypes = [vp, u32, vp, C.POINTER(HInt), vp]
    L.h2e_op_msm_unsafe.argtypes = [vp, u32, C.POINTER(HPoint), C.POINTER(HInt), vp, C.POINTER(HPoint), vp]
    L.h2e_op_ecc_assert_equal.argtypes = [vp, C.POINTER(HPoint), C.POINTER(HPoint), vp]
    L.h2e_op_assign_g2_constant.argtypes = [vp, vp, C.POINTER(HG2), vp]
    L.h2e_op_check_pairing.argtypes = [vp, u32, C.POINTER(HPoint), C.POINTER(HG2), vp]
    u64 = C.c_uint64
    L.h2e_op_to_point_with_curvature.argtypes = [vp, C.POINTER(HPoint), C.POINTER(HPointC), vp]
    L.h2e_op_ecc_reduce_with_curvature.argtypes = [vp, C.POINTER(HPoint), C.POINTER(HPointC), vp]
    L.h2e_op_ecc_double.argtypes = [vp, C.POINTER(HPointC), C.POINTER(HPoint), vp]
    L.h2e_op_ecc_add.argtypes = [vp, C.POINTER(HPointC), C.POINTER(HPoint), C.POINTER(HPoint), vp]
    L.h2e_op_ecc_neg.argtypes = [vp, C.POINTER(HPoint), C.POINTER(HPoint), vp]
    L.h2e_op_ecc_encode.argtypes = [vp, C.POINTER(HPoint), C.POINTER(u32), vp]
    L.h2e_op_ecc_mul.argtypes = [vp, C.POINTER(HPoint), C.POINTER(HInt), vp, C.POINTER(HPoint), vp]
    L.h2e_op_assign_constant_point.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), i32, C.POINTER(HPoint), vp]
    L.h2e_op_bisec_point_with_curvature.argtypes = [vp, u32, C.POINTER(HPointC), C.POINTER(HPointC), C.POINTER(HPointC), vp]
    L.h2e_op_assign_cache_point.argtypes = [vp, C.POINTER(HPointC), u64, u64, vp]
    L.h2e_op_assign_selected_point.argtypes = [vp, u32, C.POINTER(HPointC), u32, u64, C.POINTER(HPointC), vp]
    L.h2e_program_pairing.argtypes = [i32, u32, i32, i32, C.POINTER(vp)]
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
    L.h2e_job_launch_ms.argtypes = [vp, i32, C.POINTER(C.c_float), u32]
    L.h2e_last_run_expansion_launches.argtypes = [vp, C.POINTER(u32), u32]
    L.h2e_program_outputs.argtypes = [vp, C.POINTER(u32), u32]
    L.h2e_program_launches.argtypes = [vp, C.POINTER(C.c_uint64), u32]
    L.h2e_program_launch_rows.argtypes = [vp, u32, C.POINTER(C.c_uint64)]
    L.h2e_export.argtypes = [vp, vp, u32, i32, i32, i32, vp, vp, vp]
    L.h2e_digest.argtypes = [vp, vp, u32, i32, vp, vp, vp]
    L.h2e_submit.argtypes = [vp, vp, u32, vp, vp, vp, vp, vp, vp, C.POINTER(i32)]
    L.h2e_wait.argtypes = [vp, i32, vp]
    pvp = C.POINTER(vp)
    L.h2e_run_batches.argtypes = [vp, vp, u32, u32, pvp, pvp, pvp, pvp, pvp, vp]
    L.h2e_submit_batches.argtypes = [vp, vp, u32, u32, pvp, pvp, pvp, pvp, pvp, vp, C.POINTER(i32)]
    L.h2e_selftest_digit_rows.argtypes = [i32, u32, u32, vp, vp, vp]
    L.h2e_run_columns.argtypes = [vp, vp, u32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]
    L.h2e_ring_create.argtypes = [vp, vp, u32, u32, C.POINTER(vp)]
    L.h2e_ring_destroy.argtypes = [vp]
    L.h2e_ring_destroy.restype = None
    L.h2e_ring_arrays.argtypes = [vp, C.c_uint64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.h2e_ring_info.argtypes = [vp, C.POINTER(C.c_uint64), u32]
    L.h2e_ring_release.argtypes = [vp, C.c_uint64, vp]
    L.h2e_ring_submit.argtypes = [vp, C.c_uint64, vp, vp, vp, C.POINTER(i32)]
    L.h2e_ring_submit_digest.argtypes = [vp, C.c_uint64, vp, vp, vp, vp, C.POINTER(i32)]
    L.h2e_run_digest.argtypes = [vp, vp, u32, vp, vp, vp, vp, vp, vp, vp]
    L.h2e_submit_digest.argtypes = [vp, vp, u32, vp, vp, vp, vp, vp, vp, vp, C.POINTER(i32)]
    L.h2e_unit_records.argtypes = [vp, vp, u32, vp, vp, vp, vp, u32, vp]
    L.h2e_unit_record_words.argtypes = [vp]
    L.h2e_check.argtypes = [vp, vp, u32, vp, vp, vp, vp, u32, vp, vp]
    L.h2e_ctx_set_option.argtypes = [vp, i32, C.c_int64]
    L.h2e_ctx_get_stat.argtypes = [vp, i32]
    L.h2e_ctx_get_stat.restype = C.c_int64
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

    @classmethod
    def msm_bls12_381_tile(cls, n_points, emit_shape=True):
        """general-scalar MSM tile: bls12_381 G1 points, bls12_381 Fr scalars as 3-limb integers"""
        return cls._make(lib().h2e_program_msm_bls12_381_tile, n_points, int(emit_shape))

    @classmethod
    def pairing(cls, curve, n_pairs, with_expected, emit_shape=True):
        """pairing(terms) [+ fq12_assert_eq(expected, result)]; curve 0 = bn256, 1 = bls12_381"""
        return cls._make(lib().h2e_program_pairing, curve, n_pairs, int(with_expected), int(emit_shape))

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
        buf = (C.c_uint32 * 256)()
        n = lib().h2e_program_outputs(self._h, buf, 256)
        return [int(buf[i]) for i in range(n)]

    def launches(self):
        """per engine launch: dict(n_strands, n_ops, cells, dbase, drange, dselect, n_params, base0)"""
        cap = 256
        buf = (C.c_uint64 * (8 * cap))()
        n = lib().h2e_program_launches(self._h, buf, cap)
        keys = ("n_strands", "n_ops", "cells", "dbase", "drange", "dselect", "n_params", "base0")
        return [dict(zip(keys, [int(buf[8 * i + j]) for j in range(8)])) for i in range(n)]

    def launch_rows(self, launch):
        """first (base, range, select) row the k-th launch writes"""
        buf = (C.c_uint64 * 3)()
        _check(lib().h2e_program_launch_rows(self._h, launch, buf))
        return int(buf[0]), int(buf[1]), int(buf[2])

    def tape_opcodes(self, launch):
        """diagnostics: (opcodes of the launch's tape, op indices its sub-ranges start at + the op count) as numpy arrays"""
        L = lib()
        L.h2e_program_tape_opcodes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        n = L.h2e_program_tape_opcodes(self._h, launch, None, 0, None, 0, None)
        _check(min(n, 0))
        ops = np.zeros(n, dtype=np.uint16)
        subs = np.zeros(n + 2, dtype=np.uint32)
        ns = C.c_uint32(0)
        L.h2e_program_tape_opcodes(self._h, launch, ops.ctypes.data, n, subs.ctypes.data, n + 2, C.byref(ns))
        return ops, subs[:ns.value]

    def pack_order(self, launch, groups):
        """diagnostics: the packed expansion's order table of a launch for `groups` (2..32) sub-ranges per wave, [waves][groups]
        (0xffffffff = empty slot)"""
        k = {2: 0, 4: 1, 8: 2, 16: 3, 32: 4}[groups]
        L = lib()
        L.h2e_program_pack_order.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        n = L.h2e_program_pack_order(self._h, launch, k, None, 0)
        _check(min(n, 0))
        out = np.zeros(max(n, 1), dtype=np.uint32)
        L.h2e_program_pack_order(self._h, launch, k, out.ctypes.data, n)
        return out[:n].reshape(-1, groups)

    def close(self):
        if self._h:
            lib().h2e_program_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Records:
    """Operator API (include/h2e.h): a device-resident Context for a batch of instances.  Mirrors the reference's usage:
    create a context, call chip ops on it with handles, read `Records` at the end."""

    def __init__(self, engine, field_pair, n_instances, rows, scalar_field=-1, emit_shape=True, select_chip=True):
        """select_chip=False: NativeScalarEccContext::new_without_select_chip (src/context.rs:201-205) - msm_unsafe takes the bisection form"""
        self.engine, self.n, self.field_pair = engine, n_instances, field_pair
        self.rows = tuple(rows)
        self.slot_words = 6 if field_pair == FIELD_BLS12_381_FQ else 4
        h = C.c_void_p()
        flags = (1 if emit_shape else 0) | (0 if select_chip else 2)
        _check(lib().h2e_records_create(engine._h, field_pair, scalar_field, n_instances, rows[0], rows[1], rows[2], flags, C.byref(h)))
        self._h = h
        self._keep = []   # input tensors stay alive while kernels may read them

    @classmethod
    def attach(cls, engine, field_pair, arrays, offsets, msm_prefix=0, scalar_field=-1, emit_shape=True):
        """h2e_records_attach: a forked context over arrays the caller owns - `arrays` = (base, range, select, status) torch
        tensors (batch-interleaved [rows][cols][2][n][2]), ops start at `offsets` = (base, range, select) and `msm_prefix`"""
        base, rng, sel, status = arrays
        self = cls.__new__(cls)
        self.engine, self.n, self.field_pair = engine, base.shape[3], field_pair
        self.rows = (base.shape[0], rng.shape[0], sel.shape[0])
        self.slot_words = 6 if field_pair == FIELD_BLS12_381_FQ else 4
        self._keep = [base, rng, sel, status]
        cap = (C.c_uint64 * 3)(*self.rows)
        off = (C.c_uint64 * 3)(*offsets)
        h = C.c_void_p()
        _check(lib().h2e_records_attach(engine._h, field_pair, scalar_field, self.n, base.data_ptr(), rng.data_ptr(), sel.data_ptr(),
                                        status.data_ptr(), cap, off, msm_prefix, int(emit_shape), C.byref(h)))
        self._h = h
        return self

    def _in(self, values):
        """numpy uint64 [n_instances][slots][slot_words] -> device pointer"""
        t = self.engine.torch
        a = np.ascontiguousarray(values, dtype=np.uint64)
        assert a.shape[0] == self.n and a.shape[2] == self.slot_words, a.shape
        d = t.from_numpy(a.view(np.int64)).to(f"cuda:{self.engine.device}")
        self._keep.append(d)
        return d.data_ptr()

    def _s(self):
        return self.engine._stream(None).cuda_stream

    def assign_w(self, values):
        out = HInt()
        _check(lib().h2e_op_assign_w(self._h, self._in(values), C.byref(out), self._s()))
        return out

    def assign(self, values):
        out = C.c_uint32()
        _check(lib().h2e_op_assign(self._h, self._in(values), C.byref(out), self._s()))
        return out.value

    def int_op(self, which, a, b=None):
        out, cond = HInt(), C.c_uint32()
        _check(lib().h2e_op_int(self._h, which, C.byref(a), C.byref(b) if b is not None else None, C.byref(out), C.byref(cond), self._s()))
        return (out, cond.value) if which == INT_DIV else out

    def int_unary(self, which, a, b=None):
        """int_neg / int_square / int_unsafe_invert -> HInt; is_int_zero / is_int_equal -> condition cell; assert_int_equal -> None"""
        out, cond = HInt(), C.c_uint32()
        _check(lib().h2e_op_int(self._h, which, C.byref(a), C.byref(b) if b is not None else None, C.byref(out), C.byref(cond), self._s()))
        return cond.value if which in (INT_IS_ZERO, INT_IS_EQUAL) else None if which == INT_ASSERT_EQUAL else out

    def int_mul_small_constant(self, a, k):
        out = HInt()
        _check(lib().h2e_op_int_mul_small_constant(self._h, C.byref(a), k, C.byref(out), self._s()))
        return out

    def assign_int_constant(self, value):
        n = self.slot_words
        ws = (C.c_uint64 * n)(*[(value >> (64 * k)) & (2**64 - 1) for k in range(n)])
        out = HInt()
        _check(lib().h2e_op_assign_int_constant(self._h, ws, C.byref(out), self._s()))
        return out

    def bisec_int(self, cond_cell, a, b):
        out = HInt()
        _check(lib().h2e_op_bisec_int(self._h, cond_cell, C.byref(a), C.byref(b), C.byref(out), self._s()))
        return out

    def fq(self, degree, which, a, b=None, imm=0):
        """Fq2 / Fq6 / Fq12 op on assigned elements (lists of `degree` HInt); returns a list of HInt (None for assert_equal)"""
        A = (HInt * degree)(*a)
        B = (HInt * degree)(*b) if b is not None else None
        out = (HInt * degree)()
        _check(lib().h2e_op_fq(self._h, degree, which, A, B, imm, out if which != FQ_ASSERT_EQUAL else None, self._s()))
        return None if which == FQ_ASSERT_EQUAL else list(out)

    def pairing(self, g1, g2):
        a = (HPoint * len(g1))(*g1)
        b = (HG2 * len(g2))(*g2)
        out = (HInt * 12)()
        _check(lib().h2e_op_pairing(self._h, len(g1), a, b, out, self._s()))
        return list(out)

    def assign_points(self, n, values):
        out = (HPoint * n)()
        _check(lib().h2e_op_assign_points(self._h, n, self._in(values), out, self._s()))
        return out

    def assign_scalars(self, n, values):
        out = (HInt * n)()
        _check(lib().h2e_op_assign_scalars(self._h, n, self._in(values), out, self._s()))
        return out

    def msm_unsafe(self, points, scalars, values):
        out = HPoint()
        _check(lib().h2e_op_msm_unsafe(self._h, len(points), points, scalars, self._in(values), C.byref(out), self._s()))
        return out

    def ecc_assert_equal(self, a, b):
        _check(lib().h2e_op_ecc_assert_equal(self._h, C.byref(a), C.byref(b), self._s()))

    # ---- the complete-addition / curvature surface of EccChipBaseOps ----
    def to_point_with_curvature(self, a):
        out = HPointC()
        _check(lib().h2e_op_to_point_with_curvature(self._h, C.byref(a), C.byref(out), self._s()))
        return out

    def ecc_reduce_with_curvature(self, a):
        out = HPointC()
        _check(lib().h2e_op_ecc_reduce_with_curvature(self._h, C.byref(a), C.byref(out), self._s()))
        return out

    def ecc_double(self, a):
        out = HPoint()
        _check(lib().h2e_op_ecc_double(self._h, C.byref(a), C.byref(out), self._s()))
        return out

    def ecc_add(self, a, b):
        out = HPoint()
        _check(lib().h2e_op_ecc_add(self._h, C.byref(a), C.byref(b), C.byref(out), self._s()))
        return out

    def ecc_neg(self, a):
        out = HPoint()
        _check(lib().h2e_op_ecc_neg(self._h, C.byref(a), C.byref(out), self._s()))
        return out

    def ecc_encode(self, a):
        out = (C.c_uint32 * 3)()
        _check(lib().h2e_op_ecc_encode(self._h, C.byref(a), out, self._s()))
        return list(out)

    def ecc_mul(self, a, scalar, values):
        out = HPoint()
        _check(lib().h2e_op_ecc_mul(self._h, C.byref(a), C.byref(scalar), self._in(values), C.byref(out), self._s()))
        return out

    def assign_constant_point(self, x, y, is_identity=False):
        n = self.slot_words
        xs = (C.c_uint64 * n)(*[(x >> (64 * k)) & (2**64 - 1) for k in range(n)])
        ys = (C.c_uint64 * n)(*[(y >> (64 * k)) & (2**64 - 1) for k in range(n)])
        out = HPoint()
        _check(lib().h2e_op_assign_constant_point(self._h, xs, ys, int(is_identity), C.byref(out), self._s()))
        return out

    def bisec_point_with_curvature(self, cond_cell, a, b):
        out = HPointC()
        _check(lib().h2e_op_bisec_point_with_curvature(self._h, cond_cell, C.byref(a), C.byref(b), C.byref(out), self._s()))
        return out

    def assign_cache_point(self, p, group, selector):
        _check(lib().h2e_op_assign_cache_point(self._h, C.byref(p), group, selector, self._s()))

    def assign_selected_point(self, candidates, index_cell, group):
        arr = (HPointC * len(candidates))(*candidates)
        out = HPointC()
        _check(lib().h2e_op_assign_selected_point(self._h, len(candidates), arr, index_cell, group, C.byref(out), self._s()))
        return out

    def assign_g2_constant(self, values):
        out = HG2()
        _check(lib().h2e_op_assign_g2_constant(self._h, self._in(values), C.byref(out), self._s()))
        return out

    def check_pairing(self, g1, g2):
        a = (HPoint * len(g1))(*g1)
        b = (HG2 * len(g2))(*g2)
        _check(lib().h2e_op_check_pairing(self._h, len(g1), a, b, self._s()))

    def shape(self):
        s = _Shape()
        _check(lib().h2e_records_shape(self._h, C.byref(s)))
        return s

    def arrays(self):
        """the batch-interleaved advice arrays as torch tensors [rows][cols][2][n][2] + status words (views of the
        records' device memory: valid while the records are alive)"""
        t = self.engine.torch
        p = [C.c_void_p() for _ in range(4)]
        _check(lib().h2e_records_arrays(self._h, *(C.byref(x) for x in p)))
        out = []
        for k, (rows, cols) in enumerate(zip(self.rows, COLS)):
            out.append(_device_view(t, p[k].value, (rows, cols, 2, self.n, 2), t.int64, self.engine.device))
        out.append(_device_view(t, p[3].value, (self.n,), t.int32, self.engine.device))
        return out

    def close(self):
        if self._h:
            self.engine.torch.cuda.synchronize()
            lib().h2e_records_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _device_view(t, ptr, shape, dtype, device):
    """torch tensor over foreign device memory (through __cuda_array_interface__)"""
    n = int(np.prod(shape))
    itemsize = 8 if dtype == t.int64 else 4

    class _Mem:
        __cuda_array_interface__ = {"shape": (n,), "typestr": "<i8" if itemsize == 8 else "<i4", "data": (ptr, False), "version": 2}
    return t.as_tensor(_Mem(), device=f"cuda:{device}").view(shape)


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

    def alloc(self, program, n_instances, fill=0):
        """Batch-interleaved advice arrays [rows][cols][half][instance][2 words] (include/h2e.h) + status words.
        `fill`: byte the arrays are initialised with (the engine only writes assigned cells; 0xFF poisons the rest)."""
        t = self.torch
        dev = f"cuda:{self.device}"
        kw = dict(dtype=t.int64, device=dev)
        arrs = []
        for rows, cols in zip((program.base_rows, program.range_rows, program.select_rows), COLS):
            shape = (rows, cols, 2, n_instances, 2)
            assert fill in (0, 0xFF)
            arrs.append(t.zeros(shape, **kw) if fill == 0 else t.full(shape, -1, **kw))
        status = t.zeros((n_instances,), dtype=t.int32, device=dev)
        return arrs[0], arrs[1], arrs[2], status

    def upload_inputs(self, program, inputs):
        """inputs: numpy uint64 [n_instances][n_input_slots][slot_words]"""
        t = self.torch
        a = np.ascontiguousarray(inputs, dtype=np.uint64)
        assert a.shape[1:] == (program.n_input_slots, program.slot_words), (a.shape, program.n_input_slots)
        return t.from_numpy(a.view(np.int64)).to(f"cuda:{self.device}")

    def _stream(self, stream):
        return stream if stream is not None else self.torch.cuda.current_stream(self.device)

    def run(self, program, d_inputs, base, rng, sel, status, stream=None):
        n = d_inputs.shape[0]
        assert base.shape[3] == n and rng.shape[3] == n and sel.shape[3] == n, "advice arrays are batch-interleaved for n instances"
        _check(lib().h2e_run(self._h, program._h, n, d_inputs.data_ptr(), base.data_ptr(), rng.data_ptr(),
                             sel.data_ptr(), status.data_ptr(), self._stream(stream).cuda_stream))

    def submit(self, program, d_inputs, base, rng, sel, status, stream=None):
        """pipelined submission (h2e_submit): returns the job id to pass to wait()"""
        n = d_inputs.shape[0]
        assert base.shape[3] == n and rng.shape[3] == n and sel.shape[3] == n
        job = C.c_int(-1)
        _check(lib().h2e_submit(self._h, program._h, n, d_inputs.data_ptr(), base.data_ptr(), rng.data_ptr(),
                                sel.data_ptr(), status.data_ptr(), self._stream(stream).cuda_stream, C.byref(job)))
        return job.value

    def alloc_columns(self, program, n_instances):
        """zeroed per-instance column-major arrays [instances][cols][rows][4] for run_columns (zero them once: a run writes assigned cells only)"""
        t = self.torch
        return tuple(t.zeros((n_instances, cols, rows, 4), dtype=t.int64, device=f"cuda:{self.device}")
                     for rows, cols in zip((program.base_rows, program.range_rows, program.select_rows), COLS))

    def run_columns(self, program, d_inputs, base, rng, sel, status, columns, form=FORM_CANONICAL, stream=None):
        """h2e_run_columns: the run's expansions store halo2's advice columns themselves (no export pass)"""
        n = d_inputs.shape[0]
        for a, (rows, cols) in zip(columns, zip((program.base_rows, program.range_rows, program.select_rows), COLS)):
            assert tuple(a.shape) == (n, cols, rows, 4) and a.is_contiguous()
        _check(lib().h2e_run_columns(self._h, program._h, n, d_inputs.data_ptr(), base.data_ptr(), rng.data_ptr(), sel.data_ptr(),
                                     columns[0].data_ptr(), columns[1].data_ptr(), columns[2].data_ptr(), form, status.data_ptr(),
                                     self._stream(stream).cuda_stream))

    def _batch_tables(self, batches):
        """batches: [(d_inputs, base, rng, sel, status), ...] of equal instance counts -> (k, n_each, five ctypes pointer tables)"""
        k = len(batches)
        n = batches[0][0].shape[0]
        for d_in, base, rng, sel, status in batches:
            assert d_in.shape[0] == n and base.shape[3] == n and rng.shape[3] == n and sel.shape[3] == n and status.shape[0] == n
        tabs = [(C.c_void_p * k)(*[b[j].data_ptr() for b in batches]) for j in range(5)]
        return k, n, tabs

    def run_batches(self, program, batches, stream=None):
        """h2e_run_batches: several caller batches (own inputs, arrays, status words each) as ONE run"""
        k, n, t = self._batch_tables(batches)
        _check(lib().h2e_run_batches(self._h, program._h, k, n, t[0], t[1], t[2], t[3], t[4], self._stream(stream).cuda_stream))

    def submit_batches(self, program, batches, stream=None):
        """h2e_submit_batches: the pipelined form; returns the job id to pass to wait()"""
        k, n, t = self._batch_tables(batches)
        job = C.c_int(-1)
        _check(lib().h2e_submit_batches(self._h, program._h, k, n, t[0], t[1], t[2], t[3], t[4], self._stream(stream).cuda_stream, C.byref(job)))
        return job.value

    def run_digest(self, program, d_inputs, base, rng, sel, status, digests=None, stream=None):
        """h2e_run_digest: h2e_run + the stream digest of the run's three arrays, int64 [3][instances][4] (accumulated by the
        expansion while it stores: no second pass over the cells)"""
        n = d_inputs.shape[0]
        if digests is None:
            digests = self.torch.empty((3, n, 4), dtype=self.torch.int64, device=base.device)
        assert tuple(digests.shape) == (3, n, 4) and digests.is_contiguous()
        _check(lib().h2e_run_digest(self._h, program._h, n, d_inputs.data_ptr(), base.data_ptr(), rng.data_ptr(), sel.data_ptr(),
                                    status.data_ptr(), digests.data_ptr(), self._stream(stream).cuda_stream))
        return digests

    def submit_digest(self, program, d_inputs, base, rng, sel, status, digests, stream=None):
        n = d_inputs.shape[0]
        assert tuple(digests.shape) == (3, n, 4) and digests.is_contiguous()
        job = C.c_int(-1)
        _check(lib().h2e_submit_digest(self._h, program._h, n, d_inputs.data_ptr(), base.data_ptr(), rng.data_ptr(), sel.data_ptr(),
                                       status.data_ptr(), digests.data_ptr(), self._stream(stream).cuda_stream, C.byref(job)))
        return job.value

    def wait(self, job, stream=None):
        _check(lib().h2e_wait(self._h, job, self._stream(stream).cuda_stream))

    def export(self, program, region, batch, layout=LAYOUT_ROWS, form=FORM_CANONICAL, stream=None, out=None):
        """h2e_export: batch-interleaved array of one region -> per-instance arrays on the device,
        [instances][rows][cols][4] (LAYOUT_ROWS, the reference's Records layout) or [instances][cols][rows][4]
        (LAYOUT_COLUMNS, halo2's advice columns); unassigned cells zero; canonical or Montgomery-form cells."""
        t = self.torch
        rows, cols, two, n, w = batch.shape
        assert two == 2 and w == 2 and cols == COLS[region] and batch.is_contiguous()
        assert rows == (program.base_rows, program.range_rows, program.select_rows)[region]
        shape = (n, rows, cols, 4) if layout == LAYOUT_ROWS else (n, cols, rows, 4)
        if out is None:
            out = t.empty(shape, dtype=batch.dtype, device=batch.device)
        assert tuple(out.shape) == shape and out.is_contiguous()
        _check(lib().h2e_export(self._h, program._h, n, region, layout, form, batch.data_ptr(), out.data_ptr(),
                                self._stream(stream).cuda_stream))
        return out

    def digest(self, program, region, batch, stream=None, out=None):
        """h2e_digest: 32-byte digest per instance of one region's batch-interleaved array -> int64 tensor [instances][4]"""
        t = self.torch
        rows, cols, two, n, w = batch.shape
        assert two == 2 and w == 2 and cols == COLS[region] and batch.is_contiguous()
        if out is None:
            out = t.empty((n, 4), dtype=t.int64, device=batch.device)
        _check(lib().h2e_digest(self._h, program._h, n, region, batch.data_ptr(), out.data_ptr(), self._stream(stream).cuda_stream))
        return out

    def unit_records(self, program, base, status, digests=None, out=None, col0=0, stream=None):
        """h2e_unit_records: the per-unit records {status, Offset, result point cells, digests} of a finished run, one kernel.
        out: int64 [units][>= col0 + record words] (a row block of the job's table; `col0` leading columns are the caller's, e.g.
        the global unit index of the gather) or None."""
        t = self.torch
        units = status.shape[0]
        R = lib().h2e_unit_record_words(program._h)
        if out is None:
            out = t.zeros((units, col0 + R), dtype=t.int64, device=status.device)
        assert out.dim() == 2 and out.shape[0] == units and out.shape[1] >= col0 + R and out.stride(1) == 1
        _check(lib().h2e_unit_records(self._h, program._h, units, base.data_ptr(), status.data_ptr(),
                                      None if digests is None else digests.data_ptr(), out.data_ptr() + 8 * col0, out.stride(0), self._stream(stream).cuda_stream))
        return out

    def check(self, program, d_inputs, base, rng, sel, classes=0, stream=None, out=None):
        """h2e_check: the reference's MockProver criterion (base gate, range gates + lookups, select lookup, copy constraints) over
        the batch-interleaved arrays of every instance -> int64 [instances][10]: failing rows per class, then the lowest failing
        row of each class (-1 = none).  An instance passes iff out[i, :5] is all zero."""
        t = self.torch
        n = base.shape[3]
        assert base.is_contiguous() and rng.is_contiguous() and sel.is_contiguous()
        if out is None:
            out = t.empty((n, 2 * CHECK_CLASSES), dtype=t.int64, device=base.device)
        _check(lib().h2e_check(self._h, program._h, n, d_inputs.data_ptr() if d_inputs is not None else None, base.data_ptr(), rng.data_ptr(),
                               sel.data_ptr(), classes, out.data_ptr(), self._stream(stream).cuda_stream))
        return out

    def export_fixed(self, program, region, n_instances=1, d_inputs=None, layout=LAYOUT_COLUMNS, form=FORM_CANONICAL, stream=None):
        """h2e_export_fixed: fixed cells of one region on the device, [inst][cols][rows][4] (columns) or [inst][rows][cols][4]"""
        t = self.torch
        rows = (program.base_rows, program.range_rows, program.select_rows)[region]
        cols = (9, 2, 2)[region]
        shape = (n_instances, cols, rows, 4) if layout == LAYOUT_COLUMNS else (n_instances, rows, cols, 4)
        out = t.empty(shape, dtype=t.int64, device=f"cuda:{self.device}")
        _check(lib().h2e_export_fixed(self._h, program._h, region, layout, form, n_instances, d_inputs.data_ptr() if d_inputs is not None else None,
                                      out.data_ptr(), self._stream(stream).cuda_stream))
        return out

    def range_table(self, form=FORM_CANONICAL, stream=None):
        """h2e_range_table: [2][524287][4] (tag column, value column)"""
        t = self.torch
        out = t.empty((2, 524287, 4), dtype=t.int64, device=f"cuda:{self.device}")
        _check(lib().h2e_range_table(self._h, form, out.data_ptr(), self._stream(stream).cuda_stream))
        return out

    def export_copy_constraints(self, program, stream=None):
        """h2e_export_copy_constraints: int32 [n_permutations][4] = (column a, row a, column b, row b)"""
        t = self.torch
        out = t.empty((max(1, program.n_permutations), 4), dtype=t.int32, device=f"cuda:{self.device}")
        _check(lib().h2e_export_copy_constraints(self._h, program._h, out.data_ptr(), self._stream(stream).cuda_stream))
        return out[:program.n_permutations]

    def read_cell(self, base, ref, instance):
        """value of a base-chip cell reference (region << 30 | col << 27 | row) of one instance, as a Python int"""
        region, col, row = ref >> 30, (ref >> 27) & 7, ref & 0x3FFFFFF
        assert region == 0
        w = base[row, col, :, instance, :].reshape(4).cpu().numpy().view(np.uint64)
        return sum(int(w[k]) << (64 * k) for k in range(4))

    def set_option(self, option, value):
        _check(lib().h2e_ctx_set_option(self._h, option, value))

    def last_warning(self):
        """"" or why the last call of this thread will not perform as asked (h2e.h h2e_last_warning)"""
        return lib().h2e_last_warning().decode()

    def get_stat(self, stat):
        return int(lib().h2e_ctx_get_stat(self._h, stat))

    def set_profiling(self, on):
        _check(lib().h2e_set_profiling(self._h, int(on)))

    def last_run_launch_ms(self, cap=128):
        """per launched segment: (value-chain ms, expansion ms), from HIP events on the launching streams"""
        buf = (C.c_float * cap)()
        n = lib().h2e_last_run_launch_ms(self._h, buf, cap)
        if n < 0:
            _check(n)
        return [(buf[2 * i], buf[2 * i + 1]) for i in range(min(n, cap // 2))]

    def job_launch_ms(self, job, cap=128):
        """the same for the last run queued on job slot `job` (waits on the host until that run is complete)"""
        buf = (C.c_float * cap)()
        n = lib().h2e_job_launch_ms(self._h, job, buf, cap)
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
