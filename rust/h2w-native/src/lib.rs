//! Drop-in `NativeChip` / `ContextWrapper` for `shuklaayush/halo2-plonky2-verifier`, backed by `libh2w.so`.
//!
//! The reference's `verifier/src/field/native.rs:11-194` forwards every call to halo2-base's `GateChip` / `RangeChip` with
//! `ctx.ctx: &mut Context<F>`; here the same methods (names, argument order, return types) forward to the eager C ABI of
//! `include/h2w.h`, so every chip above (`field/goldilocks/*`, `hash/*`, `merkle`, `challenger`, `fri`, `stark`, `witness`)
//! compiles unchanged against `use h2w_native::{NativeChip, ContextWrapper}`.  The advice cells are materialised on the GPU
//! (`h2w_ctx_advice_device` / `h2w_ctx_download`); handle VALUES are available at once, as the reference's hint code
//! (`field/goldilocks/base.rs:27-35,349,382`) needs.  Where the reference panics, so does this (`ck`).
//!
//! SHIPPED AS SOURCE: there is no Rust toolchain in this repository's build environment, so this crate has not
//! been compiled; the raw declarations in `h2w-sys` are generated from the header and checked against it by
//! `tests/test_rust_bindings.py`.
use h2w_sys::*;
use halo2_base::{
    halo2_proofs::plonk::Assigned,
    utils::BigPrimeField,
    AssignedValue, ContextCell,
};
use std::ffi::{CStr, CString};
use std::marker::PhantomData;

fn ck(rc: i32) {
    if rc != 0 {
        panic!("libh2w: {}", unsafe { CStr::from_ptr(h2w_last_error()) }.to_string_lossy());
    }
}

/// canonical little-endian limbs <-> field element (`to_repr` / `from_repr` of halo2curves, which are canonical LE bytes)
fn to_fr<F: BigPrimeField>(a: &F) -> H2wFr {
    let bytes = a.to_bytes_le();
    let mut l = [0u64; 4];
    for (i, chunk) in bytes.chunks(8).take(4).enumerate() {
        let mut w = [0u8; 8];
        w[..chunk.len()].copy_from_slice(chunk);
        l[i] = u64::from_le_bytes(w);
    }
    H2wFr { l }
}
fn from_fr<F: BigPrimeField>(a: &H2wFr) -> F {
    let mut bytes = [0u8; 32];
    for i in 0..4 {
        bytes[8 * i..8 * i + 8].copy_from_slice(&a.l[i].to_le_bytes());
    }
    F::from_bytes_le(&bytes)
}
fn to_h2w<F: BigPrimeField>(a: &AssignedValue<F>) -> H2wAssigned {
    match a.cell {
        Some(c) => H2wAssigned { value: to_fr(a.value()), offset: c.offset as u64, ctx_id: c.context_id as u32, has_cell: 1 },
        None => H2wAssigned { value: to_fr(a.value()), offset: 0, ctx_id: 0, has_cell: 0 },
    }
}
fn from_h2w<F: BigPrimeField>(a: &H2wAssigned) -> AssignedValue<F> {
    AssignedValue {
        value: Assigned::Trivial(from_fr(&a.value)),
        cell: (a.has_cell != 0).then(|| ContextCell::new("h2w", a.ctx_id as usize, a.offset as usize)),
    }
}

/// `util/context_wrapper.rs:11-33`: the reference wraps `&mut Context<F>` and a cell-count tree; here the context lives in the
/// library (one HIP stream per handle; not thread-safe, like `&mut Context`).
pub struct ContextWrapper<F: BigPrimeField> {
    pub h2w: *mut H2wCtx,
    _f: PhantomData<F>,
}
impl<F: BigPrimeField> ContextWrapper<F> {
    /// `lookup_bits` = `k - 1` of `base_test().k(k)`; `witness_gen_only` as halo2-base's `Context::witness_gen_only()`
    pub fn new(lookup_bits: usize, witness_gen_only: bool, device_id: i32) -> Self {
        let h2w = unsafe { h2w_ctx_new(lookup_bits as i32, witness_gen_only as i32, device_id) };
        assert!(!h2w.is_null(), "h2w_ctx_new failed");
        Self { h2w, _f: PhantomData }
    }
    /// `ctx.advice.len()` (`context_wrapper.rs:24-26`)
    pub fn num_cells(&self) -> usize {
        unsafe { h2w_num_cells(self.h2w) as usize }
    }
    pub fn push_context(&mut self, _level: log::Level, ctx: &str) {
        let s = CString::new(ctx).unwrap();
        ck(unsafe { h2w_push_context(self.h2w, s.as_ptr()) });
    }
    pub fn pop_context(&mut self) {
        ck(unsafe { h2w_pop_context(self.h2w) });
    }
    /// the advice stream (canonical `Fr`), expanded on the GPU and copied back
    pub fn advice(&mut self) -> Vec<F> {
        let n = self.num_cells();
        let mut raw = vec![H2wFr::default(); n];
        ck(unsafe { h2w_ctx_download(self.h2w, 0, n as u64, raw.as_mut_ptr()) });
        raw.iter().map(from_fr).collect()
    }
}
impl<F: BigPrimeField> Drop for ContextWrapper<F> {
    fn drop(&mut self) {
        unsafe { h2w_ctx_free(self.h2w) }
    }
}

/// `field/native.rs:11-194`, method for method.
#[derive(Clone, Debug, Default)]
pub struct NativeChip<F: BigPrimeField> {
    _f: PhantomData<F>,
}
type Ctx<'a, F> = &'a mut ContextWrapper<F>;
impl<F: BigPrimeField> NativeChip<F> {
    pub fn new() -> Self {
        Self { _f: PhantomData }
    }
    fn unary(f: unsafe extern "C" fn(*mut H2wCtx, *const H2wFr, *mut H2wAssigned) -> i32, ctx: Ctx<F>, a: &F) -> AssignedValue<F> {
        let mut out = H2wAssigned::default();
        ck(unsafe { f(ctx.h2w, &to_fr(a), &mut out) });
        from_h2w(&out)
    }
    pub fn load_constant(&self, ctx: Ctx<F>, a: F) -> AssignedValue<F> { Self::unary(h2w_load_constant, ctx, &a) }          // :28-31
    pub fn load_witness(&self, ctx: Ctx<F>, a: F) -> AssignedValue<F> { Self::unary(h2w_load_witness, ctx, &a) }            // :43-46
    pub fn load_zero(&self, ctx: Ctx<F>) -> AssignedValue<F> {                                                              // :33-36
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_load_zero(ctx.h2w, &mut out) });
        from_h2w(&out)
    }
    pub fn load_constants(&self, ctx: Ctx<F>, c: &[F]) -> Vec<AssignedValue<F>> {                                           // :38-41
        let cs: Vec<H2wFr> = c.iter().map(to_fr).collect();
        let mut out = vec![H2wAssigned::default(); c.len()];
        ck(unsafe { h2w_load_constants(ctx.h2w, cs.as_ptr(), cs.len(), out.as_mut_ptr()) });
        out.iter().map(from_h2w).collect()
    }
    pub fn add(&self, ctx: Ctx<F>, a: AssignedValue<F>, b: AssignedValue<F>) -> AssignedValue<F> {                         // :48-57
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_add(ctx.h2w, &to_h2w(&a), &to_h2w(&b), &mut out) });
        from_h2w(&out)
    }
    pub fn mul(&self, ctx: Ctx<F>, a: AssignedValue<F>, b: AssignedValue<F>) -> AssignedValue<F> {                         // :59-68
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_mul(ctx.h2w, &to_h2w(&a), &to_h2w(&b), &mut out) });
        from_h2w(&out)
    }
    pub fn mul_add(&self, ctx: Ctx<F>, a: AssignedValue<F>, b: AssignedValue<F>, c: AssignedValue<F>) -> AssignedValue<F> { // :70-80
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_mul_add(ctx.h2w, &to_h2w(&a), &to_h2w(&b), &to_h2w(&c), &mut out) });
        from_h2w(&out)
    }
    pub fn select(&self, ctx: Ctx<F>, a: AssignedValue<F>, b: AssignedValue<F>, sel: AssignedValue<F>) -> AssignedValue<F> { // :82-93
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_select(ctx.h2w, &to_h2w(&a), &to_h2w(&b), &to_h2w(&sel), &mut out) });
        from_h2w(&out)
    }
    pub fn select_from_idx(&self, ctx: Ctx<F>, arr: &[AssignedValue<F>], idx: AssignedValue<F>) -> AssignedValue<F> {       // :95-104
        let a: Vec<H2wAssigned> = arr.iter().map(to_h2w).collect();
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_select_from_idx(ctx.h2w, a.as_ptr(), a.len(), &to_h2w(&idx), &mut out) });
        from_h2w(&out)
    }
    pub fn select_array_by_indicator(&self, ctx: Ctx<F>, array2d: &[Vec<AssignedValue<F>>], indicator: &[AssignedValue<F>]) -> Vec<AssignedValue<F>> { // :106-115
        let w = array2d.first().map_or(0, |r| r.len());
        let flat: Vec<H2wAssigned> = array2d.iter().flat_map(|r| r.iter().map(to_h2w)).collect();
        let ind: Vec<H2wAssigned> = indicator.iter().map(to_h2w).collect();
        let mut out = vec![H2wAssigned::default(); w];
        ck(unsafe { h2w_select_array_by_indicator(ctx.h2w, flat.as_ptr(), array2d.len(), w, ind.as_ptr(), out.as_mut_ptr()) });
        out.iter().map(from_h2w).collect()
    }
    pub fn idx_to_indicator(&self, ctx: Ctx<F>, idx: AssignedValue<F>, len: usize) -> Vec<AssignedValue<F>> {               // :117-126
        let mut out = vec![H2wAssigned::default(); len];
        ck(unsafe { h2w_idx_to_indicator(ctx.h2w, &to_h2w(&idx), len, out.as_mut_ptr()) });
        out.iter().map(from_h2w).collect()
    }
    pub fn num_to_bits(&self, ctx: Ctx<F>, a: AssignedValue<F>, range_bits: usize) -> Vec<AssignedValue<F>> {               // :128-137
        let mut out = vec![H2wAssigned::default(); range_bits];
        ck(unsafe { h2w_num_to_bits(ctx.h2w, &to_h2w(&a), range_bits, out.as_mut_ptr()) });
        out.iter().map(from_h2w).collect()
    }
    pub fn bits_to_num(&self, ctx: Ctx<F>, bits: &[AssignedValue<F>]) -> AssignedValue<F> {                                 // :139-148
        let b: Vec<H2wAssigned> = bits.iter().map(to_h2w).collect();
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_bits_to_num(ctx.h2w, b.as_ptr(), b.len(), &mut out) });
        from_h2w(&out)
    }
    pub fn decompose_le(&self, ctx: Ctx<F>, num: AssignedValue<F>, limb_bits: usize, num_limbs: usize) -> Vec<AssignedValue<F>> { // :150-160
        let mut out = vec![H2wAssigned::default(); num_limbs];
        ck(unsafe { h2w_decompose_le(ctx.h2w, &to_h2w(&num), limb_bits, num_limbs, out.as_mut_ptr()) });
        out.iter().map(from_h2w).collect()
    }
    pub fn limbs_to_num(&self, ctx: Ctx<F>, limbs: &[AssignedValue<F>], limb_bits: usize) -> AssignedValue<F> {             // :162-171
        let l: Vec<H2wAssigned> = limbs.iter().map(to_h2w).collect();
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_limbs_to_num(ctx.h2w, l.as_ptr(), l.len(), limb_bits, &mut out) });
        from_h2w(&out)
    }
    pub fn check_less_than_safe(&self, ctx: Ctx<F>, a: AssignedValue<F>, b: u64) {                                          // :173-177
        ck(unsafe { h2w_check_less_than_safe(ctx.h2w, &to_h2w(&a), b) });
    }
    pub fn range_check(&self, ctx: Ctx<F>, a: AssignedValue<F>, range_bits: usize) {                                        // :179-183
        ck(unsafe { h2w_range_check(ctx.h2w, &to_h2w(&a), range_bits) });
    }
    pub fn assert_equal(&self, ctx: Ctx<F>, a: AssignedValue<F>, b: AssignedValue<F>) {                                     // :185-192
        ck(unsafe { h2w_constrain_equal(ctx.h2w, &to_h2w(&a), &to_h2w(&b)) });
    }
}
