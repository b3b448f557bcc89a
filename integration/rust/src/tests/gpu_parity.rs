//! Pins the oracle (and through it the GPU engine) to the REAL crate.  UNVERIFIED source (no Rust toolchain in the build image).
//!
//! For every fixture `tests/golden/pyref/<name>.json` of the engine's repo - inputs as canonical little-endian 64-bit words
//! plus the summary of the `Records` the independent Python restatement (`oracle/pyref.py`) produced for them - this runs the
//! reference's own chips on the same inputs (the bodies of `src/tests/*.rs`, with the random inputs replaced by the fixture's)
//! and compares the same summary computed from the crate's `Records`:
//!   offsets, heights, number of assigned advice cells, the 32-byte digest of each advice array (include/h2e.h `h2e_digest`),
//!   SHA-256 of the permutation list in order, SHA-256 of the (cell, assigned | permute) flags per array.
//! Equal summaries = every advice value, every flag and every copy constraint equal (up to hash collisions).
//!
//! Fixtures with blinding points (the MSM tiles) need the two `C::generator() * C::Scalar::rand()` draws of
//! `EccChipScalarOps::msm_unsafe` (src/circuit/ecc_chip.rs:378-379, quirk Q1) replaced by the fixture's r1 / r2: build with
//! `--cfg h2e_fixed_blinding` and the three-line patch quoted at `fixed_blinding` below.
use crate::assign::{AssignedCondition, AssignedG2Affine, Cell, Chip};
use crate::circuit::base_chip::BaseChipOps;
use crate::circuit::ecc_chip::{EccChipBaseOps, EccChipScalarOps};
use crate::circuit::fq12::{Fq12ChipOps, Fq2ChipOps};
use crate::circuit::integer_chip::IntegerChipOps;
use crate::circuit::pairing_chip::PairingChipOps;
use crate::context::{Context, GeneralScalarEccContext, IntegerContext, NativeScalarEccContext, Records};
use crate::utils::{bn_to_field, field_to_bn};
use halo2_proofs::arithmetic::{BaseExt, CurveAffine, FieldExt};
use halo2_proofs::pairing::bn256::Fr;
use num_bigint::BigUint;
use sha2::{Digest, Sha256};
use std::cell::RefCell;
use std::rc::Rc;

const ADV_COLS: [usize; 3] = [5, 3, 2];

fn fixtures_dir() -> String {
    std::env::var("H2E_FIXTURES").expect("H2E_FIXTURES = <engine repo>/tests/golden/pyref")
}
fn load(name: &str) -> serde_json::Value {
    let text = std::fs::read_to_string(format!("{}/{}.json", fixtures_dir(), name)).unwrap();
    serde_json::from_str(&text).unwrap()
}
/// inputs_hex: [[w0, w1, ..], ..] canonical little-endian words of each input slot -> BigUint per slot
fn inputs(doc: &serde_json::Value) -> Vec<BigUint> {
    doc["inputs_hex"]
        .as_array()
        .unwrap()
        .iter()
        .map(|slot| {
            let mut v = BigUint::from(0u64);
            for (k, w) in slot.as_array().unwrap().iter().enumerate() {
                let w = u64::from_str_radix(w.as_str().unwrap().trim_start_matches("0x"), 16).unwrap();
                v += BigUint::from(w) << (64 * k);
            }
            v
        })
        .collect()
}
fn fe<F: BaseExt>(x: &BigUint) -> F {
    bn_to_field(x)
}

// ---- the summary of oracle/pyref.py:1931-1985 over the crate's Records ---------------------------------------------------
fn sm64(z: u64) -> u64 {
    let mut z = z.wrapping_add(0x9E3779B97F4A7C15);
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
    z ^ (z >> 31)
}
fn words(x: &Fr) -> [u64; 4] {
    let bn = field_to_bn(x);
    let d = bn.to_u64_digits();
    let mut w = [0u64; 4];
    for (i, v) in d.iter().enumerate() {
        w[i] = *v;
    }
    w
}
struct Summary {
    offsets: [usize; 3],
    heights: [usize; 3],
    n_advice_cells: usize,
    adv_digest: [[u64; 4]; 3],
    flags_sha256: [String; 3],
    n_permutations: usize,
    permutations_sha256: String,
}
fn cell_word(c: &Cell) -> u32 {
    let region = match c.region {
        Chip::BaseChip => 0u32,
        Chip::RangeChip => 1,
        Chip::SelectChip => 2,
    };
    (region << 30) | ((c.col as u32) << 27) | c.row as u32
}
fn summarize(ctx: &Context<Fr>) -> Summary {
    let rec: &Records<Fr> = &ctx.records;
    let mut adv_digest = [[0u64; 4]; 3];
    let mut flags_sha256: [String; 3] = Default::default();
    let mut n_cells = 0usize;
    macro_rules! region {
        ($idx:expr, $arr:expr, $height:expr) => {{
            let mut h = Sha256::new();
            for (row, cells) in $arr.iter().enumerate().take($height + 1) {
                for (col, (v, permute)) in cells.iter().enumerate() {
                    if let Some(v) = v {
                        n_cells += 1;
                        let cell = (row * ADV_COLS[$idx] + col) as u64;
                        let t = sm64(cell);
                        let w = words(v);
                        for j in 0..4 {
                            adv_digest[$idx][j] =
                                adv_digest[$idx][j].wrapping_add(sm64(w[j] ^ t ^ (j as u64).wrapping_mul(0xA24BAED4963EE407)));
                        }
                        h.update(cell.to_le_bytes());
                        h.update([1u8 | if *permute { 2 } else { 0 }]);
                    }
                }
            }
            flags_sha256[$idx] = hex::encode(h.finalize());
        }};
    }
    region!(0, rec.inner.base_adv_record, rec.base_height);
    region!(1, rec.inner.range_adv_record, rec.range_height);
    region!(2, rec.inner.select_adv_record, rec.select_height);
    let mut ph = Sha256::new();
    for (a, b) in rec.permutations.iter() {
        ph.update(cell_word(a).to_le_bytes());
        ph.update(cell_word(b).to_le_bytes());
    }
    Summary {
        offsets: [ctx.base_offset, ctx.range_offset, ctx.select_offset],
        heights: [rec.base_height, rec.range_height, rec.select_height],
        n_advice_cells: n_cells,
        adv_digest,
        flags_sha256,
        n_permutations: rec.permutations.len(),
        permutations_sha256: hex::encode(ph.finalize()),
    }
}
fn check(name: &str, s: &Summary, doc: &serde_json::Value) {
    let p = &doc["pyref"];
    let arr3 = |k: &str| -> Vec<usize> { p[k].as_array().unwrap().iter().map(|x| x.as_u64().unwrap() as usize).collect() };
    assert_eq!(s.offsets.to_vec(), arr3("offsets"), "{}: offsets", name);
    assert_eq!(s.heights.to_vec(), arr3("heights"), "{}: heights", name);
    assert_eq!(s.n_advice_cells as u64, p["n_advice_cells"].as_u64().unwrap(), "{}: advice cells", name);
    assert_eq!(s.n_permutations as u64, p["n_permutations"].as_u64().unwrap(), "{}: permutations", name);
    assert_eq!(s.permutations_sha256, p["permutations_sha256"].as_str().unwrap(), "{}: permutation list", name);
    for region in 0..3 {
        let want: Vec<u64> = p["adv_digest"][region].as_array().unwrap().iter().map(|x| x.as_u64().unwrap()).collect();
        assert_eq!(s.adv_digest[region].to_vec(), want, "{}: advice digest of region {}", name, region);
        assert_eq!(s.flags_sha256[region], p["assigned_flags_sha256"][region].as_str().unwrap(), "{}: flags of region {}", name, region);
    }
    println!("{}: the reference's Records equal the fixture ({} advice cells)", name, s.n_advice_cells);
}

// ---- the reference's test bodies on fixture inputs -------------------------------------------------------------------------
/// src/tests/integer_chip.rs:11-55 (add / sub / mul / div + division by zero); inputs a, b, a+b, a-b, a*b, a/b
fn integer_chip_st<W: BaseExt>(doc: &serde_json::Value) -> Context<Fr> {
    let v = inputs(doc);
    let ctx = Rc::new(RefCell::new(Context::new()));
    let mut ctx = IntegerContext::<W, Fr>::new(ctx);
    let a = ctx.assign_w(&v[0]);
    let b = ctx.assign_w(&v[1]);
    let c1 = ctx.assign_w(&v[2]);
    let c2 = ctx.int_add(&a, &b);
    ctx.assert_int_equal(&c1, &c2);
    let d1 = ctx.assign_w(&v[3]);
    let d2 = ctx.int_sub(&a, &b);
    ctx.assert_int_equal(&d1, &d2);
    let e1 = ctx.assign_w(&v[4]);
    let e2 = ctx.int_mul(&a, &b);
    ctx.assert_int_equal(&e1, &e2);
    let f1 = ctx.assign_w(&v[5]);
    let (_, f2) = ctx.int_div(&a, &b);
    ctx.assert_int_equal(&f1, &f2);
    let zero = ctx.int_sub(&a, &a);
    let (g1, _) = ctx.int_div(&a, &zero);
    ctx.ctx.borrow_mut().assert_true(&g1);
    Rc::try_unwrap(ctx.ctx).ok().unwrap().into_inner()
}

/// second block of src/tests/native_scalar_pairing_chip.rs:67-97: check_pairing([(a, b), (-a, b)]), G2 as constants.
/// inputs: b.x.c0, b.x.c1, b.y.c0, b.y.c1, (-a).x, (-a).y, (-a).z, a.x, a.y, a.z
fn pairing_check_bn256(doc: &serde_json::Value) -> Context<Fr> {
    use halo2_proofs::pairing::bn256::{Fq, G1Affine};
    let v = inputs(doc);
    let ctx = Rc::new(RefCell::new(Context::new()));
    let ctx = IntegerContext::<Fq, Fr>::new(ctx);
    let mut ctx = NativeScalarEccContext::<G1Affine>(ctx, 0);
    let bx = ctx.fq2_assign_constant((fe::<Fq>(&v[0]), fe::<Fq>(&v[1])));
    let by = ctx.fq2_assign_constant((fe::<Fq>(&v[2]), fe::<Fq>(&v[3])));
    let b = AssignedG2Affine::new(bx, by, AssignedCondition(ctx.0.ctx.borrow_mut().assign_constant(Fr::zero())));
    let neg_a = G1Affine::from_xy(fe(&v[4]), fe(&v[5])).unwrap();
    let a = G1Affine::from_xy(fe(&v[7]), fe(&v[8])).unwrap();
    let neg_a = ctx.assign_point(&neg_a.to_curve());
    let a = ctx.assign_point(&a.to_curve());
    ctx.check_pairing(&[(&a, &b), (&neg_a, &b)]);
    let c: Context<Fr> = ctx.into();
    c
}

/// second block of src/tests/general_scalar_pairing_chip.rs:74-105: check_pairing([(ac, b), (-a, bc)]).
/// inputs: b (4), bc (4), (-a) (x, y, z), ac (x, y, z); 6-word slots
fn pairing_check_bls12_381(doc: &serde_json::Value) -> Context<Fr> {
    use halo2_proofs::pairing::bls12_381::{Fq, G1Affine};
    let v = inputs(doc);
    let ctx = Rc::new(RefCell::new(Context::new()));
    let mut ctx = GeneralScalarEccContext::<G1Affine, Fr>::new(ctx);
    let g2 = |ctx: &mut GeneralScalarEccContext<G1Affine, Fr>, k: usize| {
        let x = ctx.fq2_assign_constant((fe::<Fq>(&v[k]), fe::<Fq>(&v[k + 1])));
        let y = ctx.fq2_assign_constant((fe::<Fq>(&v[k + 2]), fe::<Fq>(&v[k + 3])));
        AssignedG2Affine::new(x, y, AssignedCondition(ctx.native_ctx.borrow_mut().assign_constant(Fr::zero())))
    };
    let b = g2(&mut ctx, 0);
    let bc = g2(&mut ctx, 4);
    let neg_a = G1Affine::from_xy(fe(&v[8]), fe(&v[9])).unwrap();
    let ac = G1Affine::from_xy(fe(&v[11]), fe(&v[12])).unwrap();
    let neg_a = ctx.assign_point(&neg_a.to_curve());
    let ac = ctx.assign_point(&ac.to_curve());
    ctx.check_pairing(&[(&ac, &b), (&neg_a, &bc)]);
    let c: Context<Fr> = ctx.into();
    c
}

/// first block of src/tests/native_scalar_pairing_chip.rs:20-65: pairing([(a, b)]) == the expected Fq12 constant.
/// inputs: b (4), expected (12), a (x, y, z)
fn pairing_bn256_expected(doc: &serde_json::Value) -> Context<Fr> {
    use halo2_proofs::pairing::bn256::{Fq, G1Affine};
    let v = inputs(doc);
    let ctx = Rc::new(RefCell::new(Context::new()));
    let ctx = IntegerContext::<Fq, Fr>::new(ctx);
    let mut ctx = NativeScalarEccContext::<G1Affine>(ctx, 0);
    let bx = ctx.fq2_assign_constant((fe::<Fq>(&v[0]), fe::<Fq>(&v[1])));
    let by = ctx.fq2_assign_constant((fe::<Fq>(&v[2]), fe::<Fq>(&v[3])));
    let b = AssignedG2Affine::new(bx, by, AssignedCondition(ctx.0.ctx.borrow_mut().assign_constant(Fr::zero())));
    let e = |k: usize| (fe::<Fq>(&v[4 + 2 * k]), fe::<Fq>(&v[5 + 2 * k]));
    let expected = ctx.fq12_assign_constant(((e(0), e(1), e(2)), (e(3), e(4), e(5))));
    let a = G1Affine::from_xy(fe(&v[16]), fe(&v[17])).unwrap();
    let a = ctx.assign_point(&a.to_curve());
    let res = ctx.pairing(&[(&a, &b)]);
    ctx.fq12_assert_eq(&expected, &res);
    let c: Context<Fr> = ctx.into();
    c
}

/// src/tests/native_scalar_ecc_chip.rs:34-47 for one tile of n points.  inputs: (x, y, z) x n, n scalars, generator (x, y),
/// r1 (x, y), r2 (x, y), expected (x, y, z).  Needs the blinding points of msm_unsafe fixed to r1 / r2:
///
/// ```ignore
/// // src/circuit/ecc_chip.rs:378-379, under #[cfg(h2e_fixed_blinding)]
/// let (r1, r2) = crate::tests::gpu_parity::fixed_blinding::<C>();   // instead of C::generator() * C::Scalar::rand()
/// ```
#[cfg(h2e_fixed_blinding)]
pub mod fixed_blinding_hook {
    use std::cell::RefCell;
    thread_local! { pub static BLINDING: RefCell<Option<[num_bigint::BigUint; 4]>> = RefCell::new(None); }
}
#[cfg(h2e_fixed_blinding)]
pub fn fixed_blinding<C: CurveAffine>() -> (C::Curve, C::Curve) {
    fixed_blinding_hook::BLINDING.with(|b| {
        let b = b.borrow();
        let w = b.as_ref().expect("blinding points not set");
        let p = |x: &BigUint, y: &BigUint| C::from_xy(bn_to_field(x), bn_to_field(y)).unwrap().to_curve();
        (p(&w[0], &w[1]), p(&w[2], &w[3]))
    })
}
#[cfg(h2e_fixed_blinding)]
fn msm_bn256_tile(doc: &serde_json::Value, with_select: bool) -> Context<Fr> {
    use halo2_proofs::pairing::bn256::{Fq, G1Affine, G1};
    use halo2_proofs::pairing::group::Group;
    let v = inputs(doc);
    let n = (v.len() - 9) / 4;
    fixed_blinding_hook::BLINDING.with(|b| *b.borrow_mut() = Some([v[4 * n + 2].clone(), v[4 * n + 3].clone(), v[4 * n + 4].clone(), v[4 * n + 5].clone()]));
    let point = |x: &BigUint, y: &BigUint, z: &BigUint| -> G1 {
        if *z != BigUint::from(0u64) { G1::identity() } else { G1Affine::from_xy(fe(x), fe(y)).unwrap().to_curve() }
    };
    let ctx = Rc::new(RefCell::new(Context::new()));
    let ctx = IntegerContext::<Fq, Fr>::new(ctx);
    let mut ctx = if with_select { NativeScalarEccContext::new_with_select_chip(ctx) } else { NativeScalarEccContext::new_without_select_chip(ctx) };
    let points: Vec<_> = (0..n).map(|k| ctx.assign_point(&point(&v[3 * k], &v[3 * k + 1], &v[3 * k + 2]))).collect();
    let scalars: Vec<_> = (0..n).map(|k| ctx.0.ctx.borrow_mut().assign(fe::<Fr>(&v[3 * n + k]))).collect();
    let res = ctx.msm_unsafe(&points, &scalars).expect("UnsafeError on fixture inputs");
    let expect = ctx.assign_point(&point(&v[4 * n + 6], &v[4 * n + 7], &v[4 * n + 8]));
    ctx.ecc_assert_equal(&res, &expect);
    let c: Context<Fr> = ctx.into();
    c
}

#[test]
fn gpu_parity_reference_records_equal_fixtures() {
    use halo2_proofs::pairing::{bls12_381, bn256};
    for (name, fp) in [("integer_chip_st_fp0", 0), ("integer_chip_st_fp1", 1), ("integer_chip_st_fp2", 2)] {
        let doc = load(name);
        let ctx = match fp {
            0 => integer_chip_st::<bn256::Fq>(&doc),
            1 => integer_chip_st::<bls12_381::Fq>(&doc),
            _ => integer_chip_st::<bls12_381::Fr>(&doc),
        };
        check(name, &summarize(&ctx), &doc);
    }
    for (name, f) in [
        ("pairing_check_bn256_i1", pairing_check_bn256 as fn(&serde_json::Value) -> Context<Fr>),
        ("pairing_check_bls12_381_i1", pairing_check_bls12_381),
        ("pairing_bn256_1pair_expected", pairing_bn256_expected),
    ] {
        let doc = load(name);
        check(name, &summarize(&f(&doc)), &doc);
    }
    #[cfg(h2e_fixed_blinding)]
    for (name, with_select) in [("msm_bn256_tile_n33", true), ("msm_bn256_tile_n12_no_select", false), ("msm_bn256_tile_n1024", true)] {
        let doc = load(name);
        check(name, &summarize(&msm_bn256_tile(&doc, with_select)), &doc);
    }
}

/// The same inputs through the GPU engine's operator API (src/gpu/context.rs): `Records` rebuilt from the engine's arrays equal
/// the crate's own, cell for cell (needs an MI355X and libh2e.so).
#[test]
#[ignore = "needs a GPU: cargo test -- --ignored"]
fn gpu_parity_engine_records_equal_reference() {
    use crate::gpu::context::GpuContext;
    let doc = load("pairing_check_bn256_i1");
    let reference = pairing_check_bn256(&doc);
    let v = inputs(&doc);
    let mut gpu = GpuContext::new(0, crate::gpu::ffi::H2E_FIELD_BN256_FQ, -1, 1, [1 << 21, 1 << 21, 1]).unwrap();
    let b = gpu.assign_g2_constant(&[&v[0..4]]).unwrap();
    let neg_a = gpu.assign_points(&[&v[4..7]]).unwrap()[0];
    let a = gpu.assign_points(&[&v[7..10]]).unwrap()[0];
    gpu.check_pairing(&[a, neg_a], &[b, b]).unwrap();
    let rec = gpu.into_records(0).expect("status word");
    let want = summarize(&reference);
    assert_eq!((rec.base_height, rec.range_height), (reference.records.base_height, reference.records.range_height));
    let got = summarize(&Context { records: rec, base_offset: want.offsets[0], range_offset: want.offsets[1], select_offset: want.offsets[2], ..Context::new() });
    assert_eq!(got.adv_digest, want.adv_digest);
    assert_eq!(got.flags_sha256, want.flags_sha256);
    assert_eq!(got.permutations_sha256, want.permutations_sha256);
}

/// One case through each NAMED entry point of SURVEY.md 8(b) - `h2e_int_mul_batch`, `h2e_msm_bn256_tile`, `h2e_pairing_check_bn256`,
/// `h2e_pairing_check_bls12_381` (include/h2e.h: program recorded and cached per shape inside the context) - on the fixtures' inputs:
/// the 32-byte digest of every advice array the engine left on the device (`h2e_digest`) equals the crate's own `Records`' digest.
/// UNVERIFIED source like the rest of this file (no Rust toolchain in the build image); the same four comparisons run against the
/// C++ oracle and the pyref fixtures in the engine's `-m gpu` suite (tests/test_parity_gpu.py::test_named_entry_points,
/// tests/test_pyref_gpu.py).
#[test]
#[ignore = "needs a GPU: cargo test -- --ignored"]
fn gpu_parity_named_entry_points_equal_reference() {
    use crate::gpu::ffi::*;
    use std::os::raw::c_void;
    use std::ptr::null_mut;
    // device buffers of a run: inputs (canonical words of every slot), the three batch-interleaved arrays for ONE instance, status
    struct Dev(*mut c_void);
    impl Dev {
        fn zeroed(bytes: usize) -> Dev {
            let mut p = null_mut();
            unsafe {
                assert_eq!(hipMalloc(&mut p, bytes.max(16)), 0);
                assert_eq!(hipMemset(p, 0, bytes.max(16)), 0);
            }
            Dev(p)
        }
        fn upload(words: &[u64]) -> Dev {
            let d = Dev::zeroed(words.len() * 8);
            unsafe { assert_eq!(hipMemcpy(d.0, words.as_ptr() as *const c_void, words.len() * 8, 1), 0) };
            d
        }
        fn words(&self, n: usize) -> Vec<u64> {
            let mut v = vec![0u64; n];
            unsafe {
                assert_eq!(hipDeviceSynchronize(), 0);
                assert_eq!(hipMemcpy(v.as_mut_ptr() as *mut c_void, self.0, n * 8, 2), 0);
            }
            v
        }
    }
    impl Drop for Dev {
        fn drop(&mut self) {
            unsafe { hipFree(self.0) };
        }
    }
    let slot_words = |doc: &serde_json::Value| -> Vec<u64> {
        doc["inputs_hex"].as_array().unwrap().iter().flat_map(|slot| {
            slot.as_array().unwrap().iter().map(|w| u64::from_str_radix(w.as_str().unwrap().trim_start_matches("0x"), 16).unwrap()).collect::<Vec<_>>()
        }).collect()
    };
    let mut ctx = null_mut();
    unsafe { assert_eq!(h2e_ctx_create(0, &mut ctx), 0) };
    // (name of the fixture, the crate's own run of the same body, program constructor for the shape, the named entry point)
    type Named = Box<dyn Fn(*mut c_void, *const c_void, *mut c_void, *mut c_void, *mut c_void, *mut c_void) -> i32>;
    let cases: Vec<(&str, Context<Fr>, Box<dyn Fn(*mut *mut c_void) -> i32>, Named)> = vec![
        ("pairing_check_bn256_i1", pairing_check_bn256(&load("pairing_check_bn256_i1")),
         Box::new(|p| unsafe { h2e_program_pairing_check_bn256(1, p) }),
         Box::new(|c, i, b, r, s, st| unsafe { h2e_pairing_check_bn256(c, 1, i, b, r, s, st, null_mut()) })),
        ("pairing_check_bls12_381_i1", pairing_check_bls12_381(&load("pairing_check_bls12_381_i1")),
         Box::new(|p| unsafe { h2e_program_pairing_check_bls12_381(1, p) }),
         Box::new(|c, i, b, r, s, st| unsafe { h2e_pairing_check_bls12_381(c, 1, i, b, r, s, st, null_mut()) })),
        ("msm_bn256_tile_n33", msm_bn256_tile(&load("msm_bn256_tile_n33"), true),
         Box::new(|p| unsafe { h2e_program_msm_bn256_tile(33, 1, p) }),
         Box::new(|c, i, b, r, s, st| unsafe { h2e_msm_bn256_tile(c, 33, 1, i, b, r, s, st, null_mut()) })),
    ];
    for (name, reference, make, run) in cases {
        let doc = load(name);
        let mut prog = null_mut();
        assert_eq!(make(&mut prog), 0, "{}", name);
        let mut shape: h2e_shape = unsafe { std::mem::zeroed() };
        unsafe { assert_eq!(h2e_program_shape(prog, &mut shape), 0) };
        let d_in = Dev::upload(&slot_words(&doc));
        let rows = [shape.base_rows as usize, shape.range_rows as usize, shape.select_rows as usize];
        let arr: Vec<Dev> = (0..3).map(|k| Dev::zeroed(rows[k] * ADV_COLS[k] * 32)).collect();
        let status = Dev::zeroed(4);
        assert_eq!(run(ctx, d_in.0, arr[0].0, arr[1].0, arr[2].0, status.0), 0, "{}", name);
        assert_eq!(status.words(1)[0] & 0xffff_ffff, 0, "{}: status word", name);
        let want = summarize(&reference);
        for region in 0..3 {
            let dg = Dev::zeroed(32);
            unsafe { assert_eq!(h2e_digest(ctx, prog, 1, region as i32, arr[region].0, dg.0, null_mut()), 0) };
            assert_eq!(dg.words(4), want.adv_digest[region].to_vec(), "{}: advice array {}", name, region);
        }
        unsafe { h2e_program_destroy(prog) };
    }
    // h2e_int_mul_batch: n (a, b) pairs of the integer chip's fixture inputs through IntegerChipOps::int_mul on both sides
    {
        let doc = load("integer_chip_st_fp0");
        let v = inputs(&doc);
        let ctx_ref = Rc::new(RefCell::new(Context::<Fr>::new()));
        let mut ictx = IntegerContext::<halo2_proofs::pairing::bn256::Fq, Fr>::new(ctx_ref.clone());
        let a = ictx.assign_w(&v[0]);
        let b = ictx.assign_w(&v[1]);
        ictx.int_mul(&a, &b);
        drop(ictx);
        let reference = Rc::try_unwrap(ctx_ref).ok().unwrap().into_inner();
        let mut prog = null_mut();
        unsafe { assert_eq!(h2e_program_int_mul_batch(H2E_FIELD_BN256_FQ, 1, 1, &mut prog), 0) };
        let mut shape: h2e_shape = unsafe { std::mem::zeroed() };
        unsafe { assert_eq!(h2e_program_shape(prog, &mut shape), 0) };
        let words: Vec<u64> = slot_words(&doc)[..2 * shape.slot_words as usize].to_vec();
        let d_in = Dev::upload(&words);
        let rows = [shape.base_rows as usize, shape.range_rows as usize, shape.select_rows as usize];
        let arr: Vec<Dev> = (0..3).map(|k| Dev::zeroed(rows[k] * ADV_COLS[k] * 32)).collect();
        let status = Dev::zeroed(4);
        unsafe { assert_eq!(h2e_int_mul_batch(ctx, H2E_FIELD_BN256_FQ, 1, 1, d_in.0, arr[0].0, arr[1].0, arr[2].0, status.0, null_mut()), 0) };
        let want = summarize(&reference);
        for region in 0..2 {
            let dg = Dev::zeroed(32);
            unsafe { assert_eq!(h2e_digest(ctx, prog, 1, region as i32, arr[region].0, dg.0, null_mut()), 0) };
            assert_eq!(dg.words(4), want.adv_digest[region].to_vec(), "int_mul_batch: advice array {}", region);
        }
        unsafe { h2e_program_destroy(prog) };
    }
    unsafe { h2e_ctx_destroy(ctx) };
}
