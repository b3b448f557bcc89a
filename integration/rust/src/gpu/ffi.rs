//! `extern "C"` declarations of include/h2e.h.  One-to-one with the header; every function returns 0 on success and a
//! negative H2E_ERR_* otherwise (`h2e_last_error()` has text).  Device pointers are plain `*mut c_void` (hipMalloc).
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const H2E_FIELD_BN256_FQ: c_int = 0;
pub const H2E_FIELD_BLS12_381_FQ: c_int = 1;
pub const H2E_FIELD_BLS12_381_FR: c_int = 2;
pub const H2E_ST_ASSERT_FAILED: u32 = 1;
pub const H2E_ST_RETRY_ADD_SAME_OR_NEG_POINT: u32 = 2;
pub const H2E_ST_RETRY_ADD_IDENTITY: u32 = 4;
pub const H2E_ST_ARITH: u32 = 8;
pub const H2E_LAYOUT_ROWS: c_int = 0;
pub const H2E_LAYOUT_COLUMNS: c_int = 1;
/// flags of h2e_records_create's `emit_shape` argument
pub const H2E_RECORDS_EMIT_SHAPE: c_int = 1;
/// NativeScalarEccContext::new_without_select_chip (src/context.rs:201-205); with h2e_records_attach: msm_prefix0 = u64::MAX
pub const H2E_RECORDS_NO_SELECT_CHIP: c_int = 2;
/// classes of h2e_check (d_fail = [n_instances][2 * H2E_CHECK_CLASSES] u64: failing rows per class, then the lowest failing row)
pub const H2E_CHECK_BASE_GATE: u32 = 0;
pub const H2E_CHECK_RANGE_GATE: u32 = 1;
pub const H2E_CHECK_RANGE_LOOKUP: u32 = 2;
pub const H2E_CHECK_SELECT_LOOKUP: u32 = 3;
pub const H2E_CHECK_COPY: u32 = 4;
pub const H2E_CHECK_CLASSES: usize = 5;
pub const H2E_FORM_CANONICAL: c_int = 0;
pub const H2E_FORM_MONTGOMERY: c_int = 1;
pub const H2E_INT_ADD: c_int = 0;
pub const H2E_INT_SUB: c_int = 1;
pub const H2E_INT_MUL: c_int = 2;
pub const H2E_INT_DIV: c_int = 3;
pub const H2E_INT_REDUCE: c_int = 4;
pub const H2E_INT_NEG: c_int = 5;
pub const H2E_INT_SQUARE: c_int = 6;
pub const H2E_INT_UNSAFE_INVERT: c_int = 7;
pub const H2E_INT_IS_ZERO: c_int = 8;
pub const H2E_INT_IS_EQUAL: c_int = 9;
pub const H2E_INT_ASSERT_EQUAL: c_int = 10;
pub const H2E_FQ_ADD: c_int = 0;
pub const H2E_FQ_SUB: c_int = 1;
pub const H2E_FQ_MUL: c_int = 2;
pub const H2E_FQ_SQUARE: c_int = 3;
pub const H2E_FQ_NEG: c_int = 4;
pub const H2E_FQ_DOUBLE: c_int = 5;
pub const H2E_FQ_CONJUGATE: c_int = 6;
pub const H2E_FQ_UNSAFE_INVERT: c_int = 7;
pub const H2E_FQ_MUL_BY_NONRESIDUE: c_int = 8;
pub const H2E_FQ_FROBENIUS_MAP: c_int = 9;
pub const H2E_FQ_CYCLOTOMIC_SQUARE: c_int = 10;
pub const H2E_FQ_REDUCE: c_int = 11;
pub const H2E_FQ_ASSERT_EQUAL: c_int = 12;
pub const H2E_OPT_PIPELINE_DEPTH: c_int = 4;
pub const H2E_OPT_PREFAULT_HBM: c_int = 6;
pub const H2E_STAT_HW_QUEUES: c_int = 10;
pub const H2E_STAT_HW_QUEUES_WANTED: c_int = 11;

/// AssignedInteger (src/assign.rs:31-37) as cell references (region << 30 | col << 27 | row) + `times`
#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct h2e_int {
    pub limbs: [u32; 4],
    pub native: u32,
    pub times: u32,
}
/// AssignedPoint (src/assign.rs:46-51)
#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct h2e_point {
    pub x: h2e_int,
    pub y: h2e_int,
    pub z: u32,
}
/// AssignedPointWithCurvature (src/assign.rs:59-65)
#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct h2e_point_c {
    pub p: h2e_point,
    pub cv: h2e_int,
    pub cz: u32,
}
/// AssignedG2Affine (src/assign.rs:171-192)
#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct h2e_g2 {
    pub x0: h2e_int,
    pub x1: h2e_int,
    pub y0: h2e_int,
    pub y1: h2e_int,
    pub z: u32,
}
/// what `Records` holds besides advice values (src/context.rs:241-301); host arrays owned by the program / records
#[repr(C)]
pub struct h2e_shape {
    pub field_pair: c_int,
    pub slot_words: u32,
    pub n_input_slots: u32,
    pub base_offset: u64,
    pub range_offset: u64,
    pub select_offset: u64,
    pub base_height: u64,
    pub range_height: u64,
    pub select_height: u64,
    pub base_rows: u64,
    pub range_rows: u64,
    pub select_rows: u64,
    pub n_advice_cells: u64,
    pub n_permutations: u64,
    pub n_dict: u64,
    pub n_fixed_patches: u64,
    pub n_segments: u32,
    pub n_ops: u64,
    pub dict: *const u64,          // [n_dict][4] canonical values, entry 0 = None
    pub base_fix: *const u32,      // [base_rows][9] dictionary ids
    pub range_fix: *const u32,     // [range_rows][2]
    pub select_fix: *const u32,    // [select_rows][2]
    pub base_flags: *const u8,     // [base_rows][5]: bit 0 assigned, bit 1 permute
    pub range_flags: *const u8,    // [range_rows][3]
    pub select_flags: *const u8,   // [select_rows][2]
    pub permutations: *const u32,  // [n_permutations][2] cell refs
    pub fixed_patches: *const u32, // [n_fixed_patches][4]: row, fixed col, (op << 16 |) input slot, limb (-1: value mod n)
}

#[link(name = "h2e")]
extern "C" {
    pub fn h2e_last_error() -> *const c_char;
    pub fn h2e_version() -> *const c_char;
    /// "" or why the last call on this thread, though it succeeded, will not perform as asked (GPU_MAX_HW_QUEUES too small for the
    /// pipeline depth: the one process-wide knob the library cannot set for itself)
    pub fn h2e_last_warning() -> *const c_char;
    pub fn h2e_ctx_create(device: c_int, out: *mut *mut c_void) -> c_int;
    pub fn h2e_ctx_destroy(ctx: *mut c_void);
    pub fn h2e_ctx_set_option(ctx: *mut c_void, option: c_int, value: i64) -> c_int;
    pub fn h2e_ctx_get_stat(ctx: *mut c_void, stat: c_int) -> i64;
    // ---- programs: the reference's test bodies recorded once per shape ----
    pub fn h2e_program_int_mul_batch(field_pair: c_int, n: u32, emit_shape: c_int, out: *mut *mut c_void) -> c_int;
    pub fn h2e_program_integer_chip_st(field_pair: c_int, emit_shape: c_int, out: *mut *mut c_void) -> c_int;
    pub fn h2e_program_msm_bn256_tile(n_points: u32, emit_shape: c_int, out: *mut *mut c_void) -> c_int;
    pub fn h2e_program_msm_bn256_tile_no_select(n_points: u32, emit_shape: c_int, out: *mut *mut c_void) -> c_int;
    pub fn h2e_program_msm_bls12_381_tile(n_points: u32, emit_shape: c_int, out: *mut *mut c_void) -> c_int;
    pub fn h2e_program_pairing_check_bn256(emit_shape: c_int, out: *mut *mut c_void) -> c_int;
    pub fn h2e_program_pairing_check_bls12_381(emit_shape: c_int, out: *mut *mut c_void) -> c_int;
    pub fn h2e_program_pairing(curve: c_int, n_pairs: u32, with_expected: c_int, emit_shape: c_int, out: *mut *mut c_void) -> c_int;
    pub fn h2e_program_destroy(p: *mut c_void);
    pub fn h2e_program_shape(p: *const c_void, out: *mut h2e_shape) -> c_int;
    pub fn h2e_program_outputs(p: *const c_void, refs: *mut u32, cap: u32) -> c_int;
    /// one entry of 8 words per engine launch of a run: n_strands, n_ops, advice cells written per instance (0 unless emit_shape),
    /// per-strand Offset (base, range, select), n_params, first base row; returns the count
    pub fn h2e_program_launches(p: *const c_void, out: *mut u64, cap: u32) -> c_int;
    pub fn h2e_program_launch_rows(p: *const c_void, launch: u32, out: *mut u64) -> c_int;
    // diagnostics
    pub fn h2e_program_tape_opcodes(p: *const c_void, launch: u32, opcodes: *mut u16, cap: u32, subs: *mut u32, subs_cap: u32,
                                    n_subs: *mut u32) -> c_int;
    pub fn h2e_program_pack_order(p: *const c_void, launch: u32, groups_log2m1: u32, out: *mut u32, cap: u32) -> c_int;
    pub fn h2e_program_value_chain_kind(p: *const c_void, launch: u32, out3: *mut u32) -> c_int;
    // ---- execution ----
    pub fn h2e_run(ctx: *mut c_void, p: *mut c_void, n_instances: u32, d_inputs: *const c_void, d_base: *mut c_void,
                   d_range: *mut c_void, d_select: *mut c_void, d_status: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn h2e_submit(ctx: *mut c_void, p: *mut c_void, n_instances: u32, d_inputs: *const c_void, d_base: *mut c_void,
                      d_range: *mut c_void, d_select: *mut c_void, d_status: *mut c_void, stream: *mut c_void, job: *mut c_int) -> c_int;
    pub fn h2e_wait(ctx: *mut c_void, job: c_int, stream: *mut c_void) -> c_int;
    /// h2e_run whose expansion stores halo2's per-instance advice columns itself (`d_cols_*`: zeroed once, [instance][col][row][4 words]);
    /// base / range / select stay the engine's working copy
    pub fn h2e_run_columns(ctx: *mut c_void, p: *mut c_void, n_instances: u32, d_inputs: *const c_void, d_base: *mut c_void,
                           d_range: *mut c_void, d_select: *mut c_void, d_cols_base: *mut c_void, d_cols_range: *mut c_void,
                           d_cols_select: *mut c_void, form: c_int, d_status: *mut c_void, stream: *mut c_void) -> c_int;
    /// several caller batches (own inputs, arrays, status words each; `d_*` = host arrays of n_batches device pointers) as ONE run:
    /// a stream of small batches costs runs, not instances (include/h2e.h)
    pub fn h2e_run_batches(ctx: *mut c_void, p: *mut c_void, n_batches: u32, n_instances_each: u32, d_inputs: *const *const c_void,
                           d_base: *const *mut c_void, d_range: *const *mut c_void, d_select: *const *mut c_void,
                           d_status: *const *mut c_void, stream: *mut c_void) -> c_int;
    pub fn h2e_submit_batches(ctx: *mut c_void, p: *mut c_void, n_batches: u32, n_instances_each: u32, d_inputs: *const *const c_void,
                              d_base: *const *mut c_void, d_range: *const *mut c_void, d_select: *const *mut c_void,
                              d_status: *const *mut c_void, stream: *mut c_void, job: *mut c_int) -> c_int;
    /// the same with the stream digest as the consumer: d_digests = [3][n_instances][4] u64, accumulated by the expansion while it
    /// stores (a streaming job's consumer: no second pass over the cells; INTEGRATION.md "A stream of batches")
    pub fn h2e_run_digest(ctx: *mut c_void, p: *mut c_void, n_instances: u32, d_inputs: *const c_void, d_base: *mut c_void,
                          d_range: *mut c_void, d_select: *mut c_void, d_status: *mut c_void, d_digests: *mut c_void,
                          stream: *mut c_void) -> c_int;
    pub fn h2e_submit_digest(ctx: *mut c_void, p: *mut c_void, n_instances: u32, d_inputs: *const c_void, d_base: *mut c_void,
                             d_range: *mut c_void, d_select: *mut c_void, d_status: *mut c_void, d_digests: *mut c_void,
                             stream: *mut c_void, job: *mut c_int) -> c_int;
    // ---- the named entry points of SURVEY.md 8(b): program recorded and cached per shape inside the context ----
    /// IntegerChipOps::int_mul on n (a, b) pairs per instance (src/circuit/integer_chip.rs:466-483)
    pub fn h2e_int_mul_batch(ctx: *mut c_void, field_pair: c_int, n: u32, n_instances: u32, d_inputs: *const c_void,
                             d_base: *mut c_void, d_range: *mut c_void, d_select: *mut c_void, d_status: *mut c_void,
                             stream: *mut c_void) -> c_int;
    /// EccChipScalarOps::msm_unsafe on a tile of n_points (src/circuit/ecc_chip.rs:373-408), test body of
    /// src/tests/native_scalar_ecc_chip.rs:34-47 per tile
    pub fn h2e_msm_bn256_tile(ctx: *mut c_void, n_points: u32, n_tiles: u32, d_inputs: *const c_void, d_base: *mut c_void,
                              d_range: *mut c_void, d_select: *mut c_void, d_status: *mut c_void, stream: *mut c_void) -> c_int;
    /// PairingChipOps::check_pairing (src/circuit/pairing_chip.rs:173-176) on the reference's two-pair test shape
    pub fn h2e_pairing_check_bn256(ctx: *mut c_void, n_instances: u32, d_inputs: *const c_void, d_base: *mut c_void,
                                   d_range: *mut c_void, d_select: *mut c_void, d_status: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn h2e_pairing_check_bls12_381(ctx: *mut c_void, n_instances: u32, d_inputs: *const c_void, d_base: *mut c_void,
                                       d_range: *mut c_void, d_select: *mut c_void, d_status: *mut c_void, stream: *mut c_void) -> c_int;
    // ---- timing hooks (HIP events the engine records on its own streams) ----
    pub fn h2e_set_profiling(ctx: *mut c_void, enable: c_int) -> c_int;
    pub fn h2e_last_run_launch_ms(ctx: *mut c_void, ms: *mut f32, cap: u32) -> c_int;
    pub fn h2e_job_launch_ms(ctx: *mut c_void, job: c_int, ms: *mut f32, cap: u32) -> c_int;
    pub fn h2e_last_run_expansion_launches(ctx: *mut c_void, counts: *mut u32, cap: u32) -> c_int;
    // ---- h2e_ring: three runs in flight in 2.2 array sets (the rows of the program's biggest launch shared by runs k and k + 2) ----
    pub fn h2e_ring_create(ctx: *mut c_void, p: *mut c_void, n_instances: u32, depth: u32, out: *mut *mut c_void) -> c_int;
    pub fn h2e_ring_destroy(ring: *mut c_void);
    pub fn h2e_ring_arrays(ring: *const c_void, k: u64, d_base: *mut *mut c_void, d_range: *mut *mut c_void, d_select: *mut *mut c_void) -> c_int;
    pub fn h2e_ring_info(ring: *const c_void, out: *mut u64, cap: u32) -> c_int;
    /// the consumer's reads of run k end at this point of `stream`: run k + 2 (same physical rows of the shared launch) waits for it
    pub fn h2e_ring_release(ring: *mut c_void, k: u64, stream: *mut c_void) -> c_int;
    pub fn h2e_ring_submit(ring: *mut c_void, k: u64, d_inputs: *const c_void, d_status: *mut c_void, stream: *mut c_void, job: *mut c_int) -> c_int;
    pub fn h2e_ring_submit_digest(ring: *mut c_void, k: u64, d_inputs: *const c_void, d_status: *mut c_void, d_digests: *mut c_void,
                                  stream: *mut c_void, job: *mut c_int) -> c_int;
    /// test hook: the value chain's digit-row primitives on caller-chosen operands (tests/test_digit_rows_gpu.py)
    pub fn h2e_selftest_digit_rows(field_pair: c_int, op: u32, n_cases: u32, d_in: *const c_void, d_out: *mut c_void,
                                   stream: *mut c_void) -> c_int;
    // ---- hand-off ----
    pub fn h2e_export(ctx: *mut c_void, p: *mut c_void, n_instances: u32, region: c_int, layout: c_int, form: c_int,
                      d_batch: *const c_void, d_out: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn h2e_export_fixed(ctx: *mut c_void, p: *mut c_void, region: c_int, layout: c_int, form: c_int, n_instances: u32,
                            d_inputs: *const c_void, d_out: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn h2e_range_table(ctx: *mut c_void, form: c_int, d_out: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn h2e_export_copy_constraints(ctx: *mut c_void, p: *mut c_void, d_out: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn h2e_digest(ctx: *mut c_void, p: *mut c_void, n_instances: u32, region: c_int, d_batch: *const c_void,
                      d_digests: *mut c_void, stream: *mut c_void) -> c_int;
    /// the per-unit record table of a multi-GPU job's ONE collective: per instance of a finished run
    /// `h2e_unit_record_words(p)` i64 words {status, Offset (3), result point cells, 3 x 32-byte digest} at d_out + u * out_stride_words
    /// (INTEGRATION.md "The gather": the table, with a leading unit-index column, is the send buffer of the one ncclAllGather)
    pub fn h2e_unit_record_words(p: *const c_void) -> c_int;
    pub fn h2e_unit_records(ctx: *mut c_void, p: *const c_void, n_instances: u32, d_base: *const c_void, d_status: *const c_void,
                            d_digests: *const c_void, d_out: *mut c_void, out_stride_words: u32, stream: *mut c_void) -> c_int;
    /// MockProver's criterion (src/tests/mod.rs:117-132) over the arrays a run left on the device, for every instance at once:
    /// base gate, range gates + lookups, select lookup, copy constraints; `classes` = bit mask of H2E_CHECK_* (0: all)
    pub fn h2e_check(ctx: *mut c_void, p: *mut c_void, n_instances: u32, d_inputs: *const c_void, d_base: *const c_void,
                     d_range: *const c_void, d_select: *const c_void, classes: u32, d_fail: *mut c_void, stream: *mut c_void) -> c_int;
    // ---- operator API: a device-resident Context ----
    pub fn h2e_records_create(ctx: *mut c_void, field_pair: c_int, scalar_field: c_int, n_instances: u32, base_rows: u64,
                              range_rows: u64, select_rows: u64, emit_shape: c_int, out: *mut *mut c_void) -> c_int;
    /// the splice seam (ParallelClone, src/circuit/ecc_chip.rs:64-77): caller-allocated arrays, starting offsets, msm prefix
    pub fn h2e_records_attach(ctx: *mut c_void, field_pair: c_int, scalar_field: c_int, n_instances: u32, d_base: *mut c_void,
                              d_range: *mut c_void, d_select: *mut c_void, d_status: *mut c_void, capacity_rows: *const u64,
                              offset0: *const u64, msm_prefix0: u64, emit_shape: c_int, out: *mut *mut c_void) -> c_int;
    pub fn h2e_records_destroy(rec: *mut c_void);
    pub fn h2e_records_arrays(rec: *mut c_void, base: *mut *mut c_void, range: *mut *mut c_void, select: *mut *mut c_void,
                              status: *mut *mut c_void) -> c_int;
    pub fn h2e_records_shape(rec: *const c_void, out: *mut h2e_shape) -> c_int;
    pub fn h2e_op_assign_w(rec: *mut c_void, d_inputs: *const c_void, out: *mut h2e_int, stream: *mut c_void) -> c_int;
    pub fn h2e_op_assign(rec: *mut c_void, d_inputs: *const c_void, out_cell: *mut u32, stream: *mut c_void) -> c_int;
    pub fn h2e_op_int(rec: *mut c_void, which: c_int, a: *const h2e_int, b: *const h2e_int, out: *mut h2e_int, out_cond: *mut u32,
                      stream: *mut c_void) -> c_int;
    pub fn h2e_op_int_mul_small_constant(rec: *mut c_void, a: *const h2e_int, k: u64, out: *mut h2e_int, stream: *mut c_void) -> c_int;
    pub fn h2e_op_assign_int_constant(rec: *mut c_void, w_words: *const u64, out: *mut h2e_int, stream: *mut c_void) -> c_int;
    pub fn h2e_op_bisec_int(rec: *mut c_void, cond_cell: u32, a: *const h2e_int, b: *const h2e_int, out: *mut h2e_int,
                            stream: *mut c_void) -> c_int;
    pub fn h2e_op_fq(rec: *mut c_void, degree: c_int, which: c_int, a: *const h2e_int, b: *const h2e_int, imm: u64, out: *mut h2e_int,
                     stream: *mut c_void) -> c_int;
    pub fn h2e_op_assign_points(rec: *mut c_void, n: u32, d_inputs: *const c_void, out: *mut h2e_point, stream: *mut c_void) -> c_int;
    pub fn h2e_op_assign_scalars(rec: *mut c_void, n: u32, d_inputs: *const c_void, out: *mut h2e_int, stream: *mut c_void) -> c_int;
    pub fn h2e_op_msm_unsafe(rec: *mut c_void, n: u32, points: *const h2e_point, scalars: *const h2e_int, d_inputs: *const c_void,
                             out: *mut h2e_point, stream: *mut c_void) -> c_int;
    pub fn h2e_op_ecc_mul(rec: *mut c_void, a: *const h2e_point, scalar: *const h2e_int, d_inputs: *const c_void, out: *mut h2e_point,
                          stream: *mut c_void) -> c_int;
    pub fn h2e_op_ecc_assert_equal(rec: *mut c_void, a: *const h2e_point, b: *const h2e_point, stream: *mut c_void) -> c_int;
    pub fn h2e_op_to_point_with_curvature(rec: *mut c_void, a: *const h2e_point, out: *mut h2e_point_c, stream: *mut c_void) -> c_int;
    pub fn h2e_op_ecc_reduce_with_curvature(rec: *mut c_void, a: *const h2e_point, out: *mut h2e_point_c, stream: *mut c_void) -> c_int;
    pub fn h2e_op_ecc_double(rec: *mut c_void, a: *const h2e_point_c, out: *mut h2e_point, stream: *mut c_void) -> c_int;
    pub fn h2e_op_ecc_add(rec: *mut c_void, a: *const h2e_point_c, b: *const h2e_point, out: *mut h2e_point, stream: *mut c_void) -> c_int;
    pub fn h2e_op_ecc_neg(rec: *mut c_void, a: *const h2e_point, out: *mut h2e_point, stream: *mut c_void) -> c_int;
    pub fn h2e_op_ecc_encode(rec: *mut c_void, a: *const h2e_point, out_cells3: *mut u32, stream: *mut c_void) -> c_int;
    pub fn h2e_op_assign_constant_point(rec: *mut c_void, x_words: *const u64, y_words: *const u64, is_identity: c_int,
                                        out: *mut h2e_point, stream: *mut c_void) -> c_int;
    pub fn h2e_op_bisec_point_with_curvature(rec: *mut c_void, cond_cell: u32, a: *const h2e_point_c, b: *const h2e_point_c,
                                             out: *mut h2e_point_c, stream: *mut c_void) -> c_int;
    pub fn h2e_op_assign_cache_point(rec: *mut c_void, p: *const h2e_point_c, group: u64, selector: u64, stream: *mut c_void) -> c_int;
    pub fn h2e_op_assign_selected_point(rec: *mut c_void, n: u32, candidates: *const h2e_point_c, index_cell: u32, group: u64,
                                        out: *mut h2e_point_c, stream: *mut c_void) -> c_int;
    pub fn h2e_op_assign_g2_constant(rec: *mut c_void, d_inputs: *const c_void, out: *mut h2e_g2, stream: *mut c_void) -> c_int;
    pub fn h2e_op_check_pairing(rec: *mut c_void, n_pairs: u32, g1: *const h2e_point, g2: *const h2e_g2, stream: *mut c_void) -> c_int;
    pub fn h2e_op_pairing(rec: *mut c_void, n_pairs: u32, g1: *const h2e_point, g2: *const h2e_g2, out12: *mut h2e_int,
                          stream: *mut c_void) -> c_int;
}

// the HIP runtime calls the shim needs (device memory for inputs, copies back)
#[link(name = "amdhip64")]
extern "C" {
    pub fn hipMalloc(ptr: *mut *mut c_void, bytes: usize) -> c_int;
    pub fn hipFree(ptr: *mut c_void) -> c_int;
    pub fn hipMemcpy(dst: *mut c_void, src: *const c_void, bytes: usize, kind: c_int) -> c_int; // 1 = H2D, 2 = D2H
    pub fn hipMemset(dst: *mut c_void, value: c_int, bytes: usize) -> c_int;
    pub fn hipDeviceSynchronize() -> c_int;
}
