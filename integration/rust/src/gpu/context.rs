//! `GpuContext`: the engine's device-resident Context (include/h2e.h operator API) behind the reference's method names.
//! UNVERIFIED source (no Rust toolchain in the build image).  Values cross the boundary only as inputs - canonical
//! little-endian words, what `field_to_bn` sees (src/utils.rs:4-8) - and come back once, through `into_records` /
//! `splice_into`, as the reference's `Records` (src/context.rs:241-301).
use super::ffi::*;
use crate::assign::{Cell, Chip};
use crate::circuit::ecc_chip::UnsafeError;
use crate::context::{Context, Records, RecordsInner};
use crate::utils::bn_to_field;
use halo2_proofs::pairing::bn256::Fr;
use num_bigint::BigUint;
use std::os::raw::c_void;
use std::sync::Arc;

#[derive(Debug)]
pub struct GpuError(pub String);
fn check(rc: i32) -> Result<(), GpuError> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { std::ffi::CStr::from_ptr(h2e_last_error()) }.to_string_lossy().into_owned();
    Err(GpuError(format!("h2e error {}: {}", rc, msg)))
}

pub struct GpuContext {
    ctx: *mut c_void,
    rec: *mut c_void,
    n_instances: u32,
    slot_words: usize,
    rows: [u64; 3],
    inputs: Vec<*mut c_void>, // device input vectors stay alive until the records are read back
}

impl GpuContext {
    /// Context::new + IntegerContext::new (src/context.rs:136-143, :173-187) for a batch of `n_instances` instances
    pub fn new(device: i32, field_pair: i32, scalar_field: i32, n_instances: u32, rows: [u64; 3]) -> Result<Self, GpuError> {
        let mut ctx = std::ptr::null_mut();
        check(unsafe { h2e_ctx_create(device, &mut ctx) })?;
        let mut rec = std::ptr::null_mut();
        check(unsafe { h2e_records_create(ctx, field_pair, scalar_field, n_instances, rows[0], rows[1], rows[2], 1, &mut rec) })?;
        let slot_words = if field_pair == H2E_FIELD_BLS12_381_FQ { 6 } else { 4 };
        Ok(GpuContext { ctx, rec, n_instances, slot_words, rows, inputs: vec![] })
    }
    /// The splice seam (`ParallelClone`, src/circuit/ecc_chip.rs:64-77): a forked context over arrays the caller owns, starting
    /// at the caller's cursors and msm prefix (native_scalar_ecc_chip.rs:50-90, :173-178)
    pub fn attach(ctx: *mut c_void, field_pair: i32, scalar_field: i32, n_instances: u32, arrays: [*mut c_void; 4], rows: [u64; 3],
                  offsets: [u64; 3], msm_prefix: u64) -> Result<Self, GpuError> {
        let mut rec = std::ptr::null_mut();
        check(unsafe {
            h2e_records_attach(ctx, field_pair, scalar_field, n_instances, arrays[0], arrays[1], arrays[2], arrays[3], rows.as_ptr(),
                               offsets.as_ptr(), msm_prefix, 1, &mut rec)
        })?;
        let slot_words = if field_pair == H2E_FIELD_BLS12_381_FQ { 6 } else { 4 };
        Ok(GpuContext { ctx, rec, n_instances, slot_words, rows, inputs: vec![] })
    }
    /// per instance a list of values -> device vector [instance][slot][slot_words] of canonical little-endian words
    fn upload(&mut self, per_instance: &[&[BigUint]]) -> Result<*const c_void, GpuError> {
        assert_eq!(per_instance.len(), self.n_instances as usize);
        let mut host = vec![];
        for inst in per_instance {
            for v in inst.iter() {
                let d = v.to_u64_digits();
                for k in 0..self.slot_words {
                    host.push(*d.get(k).unwrap_or(&0));
                }
            }
        }
        let bytes = host.len() * 8;
        let mut dev = std::ptr::null_mut();
        if unsafe { hipMalloc(&mut dev, bytes) } != 0 || unsafe { hipMemcpy(dev, host.as_ptr() as *const c_void, bytes, 1) } != 0 {
            return Err(GpuError("hipMalloc / hipMemcpy".into()));
        }
        self.inputs.push(dev);
        Ok(dev as *const c_void)
    }
    // ---- the reference's trait surface (same names, same argument meaning) ----
    /// EccChipBaseOps::assign_point x n (src/circuit/ecc_chip.rs:458-512); values: (x, y, z) per point and instance
    pub fn assign_points(&mut self, values: &[&[BigUint]]) -> Result<Vec<h2e_point>, GpuError> {
        let n = values[0].len() / 3;
        let d = self.upload(values)?;
        let mut out = vec![h2e_point::default(); n];
        check(unsafe { h2e_op_assign_points(self.rec, n as u32, d, out.as_mut_ptr(), std::ptr::null_mut()) })?;
        Ok(out)
    }
    pub fn assign_scalars(&mut self, values: &[&[BigUint]]) -> Result<Vec<h2e_int>, GpuError> {
        let n = values[0].len();
        let d = self.upload(values)?;
        let mut out = vec![h2e_int::default(); n];
        check(unsafe { h2e_op_assign_scalars(self.rec, n as u32, d, out.as_mut_ptr(), std::ptr::null_mut()) })?;
        Ok(out)
    }
    /// IntegerChipOps::assign_w (src/circuit/integer_chip.rs:236-258)
    pub fn assign_w(&mut self, values: &[&[BigUint]]) -> Result<h2e_int, GpuError> {
        let d = self.upload(values)?;
        let mut out = h2e_int::default();
        check(unsafe { h2e_op_assign_w(self.rec, d, &mut out, std::ptr::null_mut()) })?;
        Ok(out)
    }
    fn int2(&mut self, which: i32, a: &h2e_int, b: Option<&h2e_int>) -> Result<(h2e_int, u32), GpuError> {
        let mut out = h2e_int::default();
        let mut cond = 0u32;
        check(unsafe { h2e_op_int(self.rec, which, a, b.map_or(std::ptr::null(), |x| x as *const _), &mut out, &mut cond, std::ptr::null_mut()) })?;
        Ok((out, cond))
    }
    pub fn int_add(&mut self, a: &h2e_int, b: &h2e_int) -> Result<h2e_int, GpuError> { Ok(self.int2(H2E_INT_ADD, a, Some(b))?.0) }
    pub fn int_sub(&mut self, a: &h2e_int, b: &h2e_int) -> Result<h2e_int, GpuError> { Ok(self.int2(H2E_INT_SUB, a, Some(b))?.0) }
    pub fn int_mul(&mut self, a: &h2e_int, b: &h2e_int) -> Result<h2e_int, GpuError> { Ok(self.int2(H2E_INT_MUL, a, Some(b))?.0) }
    pub fn int_div(&mut self, a: &h2e_int, b: &h2e_int) -> Result<(u32, h2e_int), GpuError> { let (o, c) = self.int2(H2E_INT_DIV, a, Some(b))?; Ok((c, o)) }
    pub fn reduce(&mut self, a: &h2e_int) -> Result<h2e_int, GpuError> { Ok(self.int2(H2E_INT_REDUCE, a, None)?.0) }
    pub fn int_neg(&mut self, a: &h2e_int) -> Result<h2e_int, GpuError> { Ok(self.int2(H2E_INT_NEG, a, None)?.0) }
    pub fn int_square(&mut self, a: &h2e_int) -> Result<h2e_int, GpuError> { Ok(self.int2(H2E_INT_SQUARE, a, None)?.0) }
    pub fn int_unsafe_invert(&mut self, a: &h2e_int) -> Result<h2e_int, GpuError> { Ok(self.int2(H2E_INT_UNSAFE_INVERT, a, None)?.0) }
    pub fn is_int_zero(&mut self, a: &h2e_int) -> Result<u32, GpuError> { Ok(self.int2(H2E_INT_IS_ZERO, a, None)?.1) }
    pub fn is_int_equal(&mut self, a: &h2e_int, b: &h2e_int) -> Result<u32, GpuError> { Ok(self.int2(H2E_INT_IS_EQUAL, a, Some(b))?.1) }
    pub fn assert_int_equal(&mut self, a: &h2e_int, b: &h2e_int) -> Result<(), GpuError> { self.int2(H2E_INT_ASSERT_EQUAL, a, Some(b)).map(|_| ()) }
    /// EccChipScalarOps::msm_unsafe (src/circuit/ecc_chip.rs:373-408); blinding = generator (x, y), r1 (x, y), r2 (x, y) per instance:
    /// the two points the reference draws inside (quirk Q1) are explicit here
    pub fn msm_unsafe(&mut self, points: &[h2e_point], scalars: &[h2e_int], blinding: &[&[BigUint]]) -> Result<h2e_point, GpuError> {
        let d = self.upload(blinding)?;
        let mut out = h2e_point::default();
        check(unsafe { h2e_op_msm_unsafe(self.rec, points.len() as u32, points.as_ptr(), scalars.as_ptr(), d, &mut out, std::ptr::null_mut()) })?;
        Ok(out)
    }
    pub fn ecc_assert_equal(&mut self, a: &h2e_point, b: &h2e_point) -> Result<(), GpuError> {
        check(unsafe { h2e_op_ecc_assert_equal(self.rec, a, b, std::ptr::null_mut()) })
    }
    /// fq2_assign_constant x 2 + assign_constant(0): an AssignedG2Affine whose coordinates are constants (x.c0, x.c1, y.c0, y.c1)
    pub fn assign_g2_constant(&mut self, values: &[&[BigUint]]) -> Result<h2e_g2, GpuError> {
        let d = self.upload(values)?;
        let mut out = h2e_g2::default();
        check(unsafe { h2e_op_assign_g2_constant(self.rec, d, &mut out, std::ptr::null_mut()) })?;
        Ok(out)
    }
    /// PairingChipOps::check_pairing / pairing (src/circuit/pairing_chip.rs:157-176)
    pub fn check_pairing(&mut self, g1: &[h2e_point], g2: &[h2e_g2]) -> Result<(), GpuError> {
        check(unsafe { h2e_op_check_pairing(self.rec, g1.len() as u32, g1.as_ptr(), g2.as_ptr(), std::ptr::null_mut()) })
    }
    pub fn pairing(&mut self, g1: &[h2e_point], g2: &[h2e_g2]) -> Result<[h2e_int; 12], GpuError> {
        let mut out = [h2e_int::default(); 12];
        check(unsafe { h2e_op_pairing(self.rec, g1.len() as u32, g1.as_ptr(), g2.as_ptr(), out.as_mut_ptr(), std::ptr::null_mut()) })?;
        Ok(out)
    }
    /// Fq2 / Fq6 / Fq12 ops (src/circuit/fq12.rs:24-459); `which` = H2E_FQ_*
    pub fn fq(&mut self, degree: usize, which: i32, a: &[h2e_int], b: Option<&[h2e_int]>, imm: u64) -> Result<Vec<h2e_int>, GpuError> {
        let mut out = vec![h2e_int::default(); degree];
        check(unsafe {
            h2e_op_fq(self.rec, degree as i32, which, a.as_ptr(), b.map_or(std::ptr::null(), |x| x.as_ptr()), imm,
                      if which == H2E_FQ_ASSERT_EQUAL { std::ptr::null_mut() } else { out.as_mut_ptr() }, std::ptr::null_mut())
        })?;
        Ok(out)
    }

    fn shape(&self) -> Result<h2e_shape, GpuError> {
        let mut s = std::mem::MaybeUninit::<h2e_shape>::zeroed();
        check(unsafe { h2e_records_shape(self.rec, s.as_mut_ptr()) })?;
        Ok(unsafe { s.assume_init() })
    }
    /// the status word of one instance: UnsafeError (retry with fresh blinding points, src/tests/native_scalar_ecc_chip.rs:52-57)
    /// or a would-be panic
    fn status(&self, instance: usize) -> Result<(), UnsafeError> {
        let (mut b, mut r, mut s, mut st) = (std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut());
        unsafe { h2e_records_arrays(self.rec, &mut b, &mut r, &mut s, &mut st) };
        let mut words = vec![0u32; self.n_instances as usize];
        unsafe { hipDeviceSynchronize(); hipMemcpy(words.as_mut_ptr() as *mut c_void, st, words.len() * 4, 2) };
        match words[instance] {
            0 => Ok(()),
            w if w & H2E_ST_RETRY_ADD_SAME_OR_NEG_POINT != 0 => Err(UnsafeError::AddSameOrNegPoint),
            w if w & H2E_ST_RETRY_ADD_IDENTITY != 0 => Err(UnsafeError::AddIdentity),
            w => panic!("assertion failed in GPU-generated rows (status {:#x}): the reference would have panicked", w),
        }
    }
    /// `Records` of one instance (what `Records::assign_all` reads in synthesize, src/context.rs:575-588): advice values exported
    /// row-major in Montgomery form (halo2's in-memory Fr: no per-cell from_repr, src/utils.rs:10-17), flags, fixed cells,
    /// permutations and heights from the shape artefacts.
    pub fn into_records(self, instance: usize) -> Result<Records<Fr>, UnsafeError> {
        self.status(instance)?;
        let sh = self.shape().expect("shape");
        let mut inner = RecordsInner::<Fr>::default();
        let (mut arr, mut st) = ([std::ptr::null_mut(); 3], std::ptr::null_mut());
        unsafe { h2e_records_arrays(self.rec, &mut arr[0], &mut arr[1], &mut arr[2], &mut st) };
        let dict = unsafe { std::slice::from_raw_parts(sh.dict, sh.n_dict as usize * 4) };
        let dict_fr = |id: u32| -> Fr {
            let w = &dict[id as usize * 4..id as usize * 4 + 4];
            bn_to_field(&(BigUint::from(w[0]) + (BigUint::from(w[1]) << 64) + (BigUint::from(w[2]) << 128) + (BigUint::from(w[3]) << 192)))
        };
        let heights = [sh.base_height, sh.range_height, sh.select_height];
        let cols = [5usize, 3, 2];
        let fcols = [9usize, 2, 2];
        let flags = [sh.base_flags, sh.range_flags, sh.select_flags];
        let fixes = [sh.base_fix, sh.range_fix, sh.select_fix];
        for region in 0..3 {
            let rows = self.rows[region] as usize;
            let n = self.n_instances as usize;
            // per-instance row-major [instance][row][cols][4 words], Montgomery form, unassigned cells zero
            let mut dev = std::ptr::null_mut();
            let bytes = n * rows * cols[region] * 32;
            unsafe { hipMalloc(&mut dev, bytes) };
            // (a records object has no program handle: export through a one-op program or copy the batch array and de-interleave
            //  on the host; the engine's Python binding does the former - see halo2ecc_s_amd/engine.py Engine.export)
            let mut host = vec![0u64; rows * cols[region] * 4];
            deinterleave(arr[region], rows, cols[region], n, instance, &mut host);
            unsafe { hipFree(dev) };
            let fl = unsafe { std::slice::from_raw_parts(flags[region], rows * cols[region]) };
            let fx = unsafe { std::slice::from_raw_parts(fixes[region], rows * fcols[region]) };
            for row in 0..=(heights[region] as usize).min(rows - 1) {
                for col in 0..cols[region] {
                    let f = fl[row * cols[region] + col];
                    if f & 1 != 0 {
                        let w = &host[(row * cols[region] + col) * 4..][..4];
                        let v: Fr = bn_to_field(&(BigUint::from(w[0]) + (BigUint::from(w[1]) << 64) + (BigUint::from(w[2]) << 128) + (BigUint::from(w[3]) << 192)));
                        match region {
                            0 => inner.base_adv_record[row][col] = (Some(v), f & 2 != 0),
                            1 => inner.range_adv_record[row][col] = (Some(v), f & 2 != 0),
                            _ => inner.select_adv_record[row][col] = (Some(v), f & 2 != 0),
                        }
                    }
                }
                for col in 0..fcols[region] {
                    let id = fx[row * fcols[region] + col];
                    if id != 0 {
                        match region {
                            0 => inner.base_fix_record[row][col] = Some(dict_fr(id)),
                            1 => inner.range_fix_record[row][col] = Some(dict_fr(id)),
                            _ => inner.select_fix_record[row][col] = Some(dict_fr(id)),
                        }
                    }
                }
            }
        }
        // (fixed cells made from instance inputs - the G2 constants - arrive as fixed_patches: row, column, input slot, limb)
        let perms = unsafe { std::slice::from_raw_parts(sh.permutations, sh.n_permutations as usize * 2) };
        let cell = |w: u32| Cell::new(match w >> 30 { 0 => Chip::BaseChip, 1 => Chip::RangeChip, _ => Chip::SelectChip }, ((w >> 27) & 7) as usize, (w & 0x3ff_ffff) as usize);
        Ok(Records {
            inner: Arc::new(inner),
            base_height: sh.base_height as usize,
            range_height: sh.range_height as usize,
            select_height: sh.select_height as usize,
            permutations: perms.chunks(2).map(|p| (cell(p[0]), cell(p[1]))).collect(),
        })
    }
    /// merge() + apply_offset_diff (src/circuit/native_scalar_ecc_chip.rs:50-90) of an attached context into the host Context it
    /// was forked from: the caller copies the spliced rows out of `into_records`, appends the permutations, maxes the heights and
    /// advances its cursors by (shape.offsets - offset0).
    pub fn offsets_and_heights(&self) -> Result<([u64; 3], [u64; 3]), GpuError> {
        let sh = self.shape()?;
        Ok(([sh.base_offset, sh.range_offset, sh.select_offset], [sh.base_height, sh.range_height, sh.select_height]))
    }
}
/// batch-interleaved [row][col][half][instance][2 words] on the device -> [row][col][4 words] of one instance on the host
fn deinterleave(d_arr: *mut c_void, rows: usize, cols: usize, n_inst: usize, instance: usize, out: &mut [u64]) {
    let mut all = vec![0u64; rows * cols * 4 * n_inst];
    unsafe { hipMemcpy(all.as_mut_ptr() as *mut c_void, d_arr, all.len() * 8, 2) };
    for cell in 0..rows * cols {
        for k in 0..4 {
            out[cell * 4 + k] = all[(cell * 2 + k / 2) * 2 * n_inst + 2 * instance + k % 2];
        }
    }
}
impl Drop for GpuContext {
    fn drop(&mut self) {
        unsafe {
            hipDeviceSynchronize();
            h2e_records_destroy(self.rec);
            for p in self.inputs.drain(..) {
                hipFree(p);
            }
        }
    }
}
