//! Drop-in `NativeChip` / `ContextWrapper` (and the few halo2-base items they are built from) for
//! `shuklaayush/halo2-plonky2-verifier`, backed by `libh2w.so`.
//!
//! The reference's `verifier/src/field/native.rs:11-194` forwards every call to halo2-base's `GateChip` / `RangeChip` with
//! `ctx.ctx: &mut Context<F>`; here the same items with the same shapes forward to the eager C ABI of `include/h2w.h`:
//!
//! | reference call shape | file:line | here |
//! |---|---|---|
//! | `ContextWrapper::new(ctx)` with `ctx: &'ctx mut Context<F>`; `pub ctx` field | `util/context_wrapper.rs:11-22` | [`ContextWrapper`] |
//! | `ctx.ctx.advice.len()` | `util/context_wrapper.rs:24-26` | [`Context::advice`] ([`Advice::len`]) |
//! | `ctx.ctx.load_constant(..)` / `load_zero()` / `load_constants(..)` / `load_witness(..)` / `constrain_equal(&a, &b)` | `field/native.rs:28-46,185-193`, `hash/poseidon_bn254/permutation.rs:88-104` | [`Context`] |
//! | `NativeChip::new(range.clone())`, `gate_chip()`, `range_chip()` | `field/native.rs:15-25`, `stark/mod.rs:485-488` | [`NativeChip`], [`RangeChip`], [`GateChip`] |
//! | `base_test().k(k).run(\|ctx, range\| ..)` | `stark/mod.rs:427-456,483-518` | [`base_test`] |
//!
//! What a port of the reference changes: the `use halo2_base::{gates::{GateChip, RangeChip, ..}, Context, utils::testing::base_test}`
//! lines of `field/native.rs`, `util/context_wrapper.rs` and the test modules become `use h2w_native::{..}`; `AssignedValue`,
//! `Assigned`, `BigPrimeField`, `fe_to_biguint` stay halo2-base's, so the chips above (`field/goldilocks/*`, `hash/*`, `merkle`,
//! `challenger`, `fri`, `stark`, `witness`) need no other edit.  `field/native.rs` itself is replaced by [`NativeChip`] below
//! (its bodies were one-line forwards to `GateInstructions` / `RangeInstructions`).  The advice cells are materialised on the GPU
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

/// halo2-base `Context<F>` as far as the reference touches it directly (`ctx.ctx.*`): the advice lives in the library (one handle =
/// one `&mut Context`: not thread-safe), `advice.len()` is the reference's cell counter (`util/context_wrapper.rs:24-26`).
pub struct Context<F: BigPrimeField> {
    h2w: *mut H2wCtx,
    /// `ctx.advice.len()`
    pub advice: Advice,
    _f: PhantomData<F>,
}
/// stands in for `Context::advice: Vec<Assigned<F>>` where only its length is read
pub struct Advice {
    h2w: *mut H2wCtx,
}
impl Advice {
    pub fn len(&self) -> usize {
        unsafe { h2w_num_cells(self.h2w) as usize }
    }
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }
}
impl<F: BigPrimeField> Context<F> {
    /// `lookup_bits` = `k - 1` of `base_test().k(k)`; `witness_gen_only` as halo2-base's `Context::witness_gen_only()`
    pub fn new(lookup_bits: usize, witness_gen_only: bool, device_id: i32) -> Self {
        let h2w = unsafe { h2w_ctx_new(lookup_bits as i32, witness_gen_only as i32, device_id) };
        assert!(!h2w.is_null(), "h2w_ctx_new failed");
        Self { h2w, advice: Advice { h2w }, _f: PhantomData }
    }
    pub fn load_constant(&mut self, c: F) -> AssignedValue<F> {                     // native.rs:28-31, poseidon_bn254/permutation.rs:88-104
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_load_constant(self.h2w, &to_fr(&c), &mut out) });
        from_h2w(&out)
    }
    pub fn load_witness(&mut self, w: F) -> AssignedValue<F> {                      // native.rs:43-46
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_load_witness(self.h2w, &to_fr(&w), &mut out) });
        from_h2w(&out)
    }
    pub fn load_zero(&mut self) -> AssignedValue<F> {                               // native.rs:33-36
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_load_zero(self.h2w, &mut out) });
        from_h2w(&out)
    }
    pub fn load_constants(&mut self, c: &[F]) -> Vec<AssignedValue<F>> {            // native.rs:38-41
        let cs: Vec<H2wFr> = c.iter().map(to_fr).collect();
        let mut out = vec![H2wAssigned::default(); c.len()];
        ck(unsafe { h2w_load_constants(self.h2w, cs.as_ptr(), cs.len(), out.as_mut_ptr()) });
        out.iter().map(from_h2w).collect()
    }
    pub fn constrain_equal(&mut self, a: &AssignedValue<F>, b: &AssignedValue<F>) { // native.rs:189-192
        ck(unsafe { h2w_constrain_equal(self.h2w, &to_h2w(a), &to_h2w(b)) });
    }
    /// the context as new, its host memory kept: for the next proof through the same context
    pub fn reset(&mut self) {
        ck(unsafe { h2w_ctx_reset(self.h2w) });
    }
    /// (block records, literal cells) the run so far appended: the sizes for `reserve` on a new context of the same shape
    pub fn footprint(&self) -> (u64, u64) {
        let mut out = [0u64; 2];
        ck(unsafe { h2w_ctx_footprint(self.h2w, out.as_mut_ptr()) });
        (out[0], out[1])
    }
    /// Vec::with_capacity for a new context: its host vectors sized and mapped ahead of the first run
    pub fn reserve(&mut self, n_records: u64, n_literal_cells: u64) {
        ck(unsafe { h2w_ctx_reserve(self.h2w, n_records, n_literal_cells) });
    }
    // ---- record and replay (include/h2w.h 2d): ONE run of the unchanged chips on this context becomes a plan the GPU replays
    /// start recording the op tape (on a fresh context)
    pub fn trace_begin(&mut self) {
        ck(unsafe { h2w_ctx_trace_begin(self.h2w) });
    }
    /// the next `load_witness` / `load_constant` takes its value from proof word(s) `[word, word + n_words)` of the flat layout (witness/mod.rs:48-225);
    /// a no-op unless the context is tracing
    pub fn trace_input(&mut self, word: u64, n_words: u32) {
        ck(unsafe { h2w_trace_input(self.h2w, word, n_words) });
    }
    /// the hint of `GoldilocksQuadExtChip::inv` (extension.rs:320-340), computed inside the library
    pub fn gl_ext_inv_witness(&mut self, a: &[AssignedValue<F>; 2]) -> [AssignedValue<F>; 2] {
        let (ain, mut out) = ([to_h2w(&a[0]), to_h2w(&a[1])], [H2wAssigned::default(); 2]);
        ck(unsafe { h2w_gl_ext_inv_witness(self.h2w, ain.as_ptr(), out.as_mut_ptr()) });
        [from_h2w(&out[0]), from_h2w(&out[1])]
    }
    /// the recorded run as a plan for `h2w_fri_witness_batch`; `parallel_scopes`: the `#[count]` scopes whose instances are independent
    /// (`"verify_query_round"`, fri/mod.rs:488-501; `"verify_proof_to_cap_with_cap_index"`, merkle/mod.rs:57-78) - checked by the library
    pub fn plan_from_trace(&mut self, proof_words: u64, parallel_scopes: &[&std::ffi::CStr], device_id: i32) -> *mut H2wPlan {
        let names: Vec<*const std::os::raw::c_char> = parallel_scopes.iter().map(|s| s.as_ptr()).collect();
        let plan = unsafe { h2w_plan_from_trace(self.h2w, proof_words, names.as_ptr(), names.len(), device_id) };
        if plan.is_null() { panic!("{}", unsafe { std::ffi::CStr::from_ptr(h2w_last_error()) }.to_string_lossy()); }
        plan
    }
    /// the advice stream (canonical `Fr`), expanded on the GPU and copied back
    pub fn advice_values(&mut self) -> Vec<F> {
        let n = self.advice.len();
        let mut raw = vec![H2wFr::default(); n];
        ck(unsafe { h2w_ctx_download(self.h2w, 0, n as u64, raw.as_mut_ptr()) });
        raw.iter().map(from_fr).collect()
    }
}
impl<F: BigPrimeField> Drop for Context<F> {
    fn drop(&mut self) {
        unsafe { h2w_ctx_free(self.h2w) }
    }
}

/// halo2-base `GateChip<F>` / `RangeChip<F>` as far as the reference holds them (`NativeChip::new(range.clone())`,
/// `gate_chip()`, `range_chip()`): the templates themselves run inside the library, so these carry only `lookup_bits`.
#[derive(Clone, Debug, Default)]
pub struct GateChip<F: BigPrimeField> {
    _f: PhantomData<F>,
}
#[derive(Clone, Debug)]
pub struct RangeChip<F: BigPrimeField> {
    gate: GateChip<F>,
    lookup_bits: usize,
}
impl<F: BigPrimeField> RangeChip<F> {
    pub fn new(lookup_bits: usize) -> Self {
        Self { gate: GateChip { _f: PhantomData }, lookup_bits }
    }
    pub fn gate(&self) -> &GateChip<F> {
        &self.gate
    }
    pub fn lookup_bits(&self) -> usize {
        self.lookup_bits
    }
}

/// `halo2_base::utils::testing::base_test().k(k).run(|ctx, range| ..)` (`stark/mod.rs:427-456,483-518`): `lookup_bits = k - 1`,
/// a keygen-mode context (`witness_gen_only = false`), the closure, then the advice is materialised on the GPU.
pub struct BaseTester {
    k: u32,
    device_id: i32,
}
pub fn base_test() -> BaseTester {
    BaseTester { k: 10, device_id: 0 }
}
impl BaseTester {
    pub fn k(mut self, k: u32) -> Self {
        self.k = k;
        self
    }
    pub fn device(mut self, device_id: i32) -> Self {
        self.device_id = device_id;
        self
    }
    pub fn run<F: BigPrimeField, R>(&self, f: impl FnOnce(&mut Context<F>, &RangeChip<F>) -> R) -> R {
        let lookup_bits = (self.k - 1) as usize;
        let mut ctx = Context::<F>::new(lookup_bits, false, self.device_id);
        let range = RangeChip::<F>::new(lookup_bits);
        let r = f(&mut ctx, &range);
        let mut dev: *mut std::ffi::c_void = std::ptr::null_mut();
        ck(unsafe { h2w_ctx_advice_device(ctx.h2w, &mut dev) });          // every cell is produced (on the GPU) before the tester returns
        r
    }
}

/// `util/context_wrapper.rs:11-33`: the reference wraps `&'ctx mut Context<F>` and a cell-count tree (`#[count]` drives
/// `push_context` / `pop_context`); the tree lives in the library (`h2w_push_context` / `h2w_pop_context` / `h2w_context_dump`).
pub struct ContextWrapper<'ctx, F: BigPrimeField> {
    pub ctx: &'ctx mut Context<F>,
}
impl<'ctx, F: BigPrimeField> ContextWrapper<'ctx, F> {
    pub fn new(ctx: &'ctx mut Context<F>) -> Self {
        Self { ctx }
    }
    /// `self.ctx.advice.len()` (`context_wrapper.rs:24-26`)
    pub fn num_cells(&self) -> usize {
        self.ctx.advice.len()
    }
    pub fn push_context(&mut self, _level: log::Level, ctx: &str) {
        let s = CString::new(ctx).unwrap();
        ck(unsafe { h2w_push_context(self.ctx.h2w, s.as_ptr()) });
    }
    pub fn pop_context(&mut self) {
        ck(unsafe { h2w_pop_context(self.ctx.h2w) });
    }
    /// `print_cell_counts` (`context_wrapper.rs:36-54`): the collapsed-stack lines `a;b;c <cells>` of the scope tree
    pub fn print_cell_counts(&self) {
        let need = unsafe { h2w_context_dump(self.ctx.h2w, std::ptr::null_mut(), 0) };
        let mut buf = vec![0u8; need + 1];
        unsafe { h2w_context_dump(self.ctx.h2w, buf.as_mut_ptr() as *mut std::os::raw::c_char, buf.len()) };
        print!("{}", String::from_utf8_lossy(&buf[..need]));
    }
}

/// `field/native.rs:11-194`, method for method.
#[derive(Clone, Debug)]
pub struct NativeChip<F: BigPrimeField> {
    range_chip: RangeChip<F>,
}
type Ctx<'a, 'ctx, F> = &'a mut ContextWrapper<'ctx, F>;
impl<F: BigPrimeField> NativeChip<F> {
    pub fn new(range_chip: RangeChip<F>) -> Self {                                   // :15-17
        Self { range_chip }
    }
    pub fn gate_chip(&self) -> &GateChip<F> {                                         // :19-21
        self.range_chip.gate()
    }
    pub fn range_chip(&self) -> &RangeChip<F> {                                       // :23-25
        &self.range_chip
    }
    pub fn load_constant(&self, ctx: Ctx<'_, '_, F>, a: F) -> AssignedValue<F> { ctx.ctx.load_constant(a) }                       // :28-31
    pub fn load_zero(&self, ctx: Ctx<'_, '_, F>) -> AssignedValue<F> { ctx.ctx.load_zero() }                                        // :33-36
    pub fn load_constants(&self, ctx: Ctx<'_, '_, F>, c: &[F]) -> Vec<AssignedValue<F>> { ctx.ctx.load_constants(c) }               // :38-41
    pub fn load_witness(&self, ctx: Ctx<'_, '_, F>, a: F) -> AssignedValue<F> { ctx.ctx.load_witness(a) }                           // :43-46
    pub fn add(&self, ctx: Ctx<'_, '_, F>, a: AssignedValue<F>, b: AssignedValue<F>) -> AssignedValue<F> {                         // :48-57
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_add(ctx.ctx.h2w, &to_h2w(&a), &to_h2w(&b), &mut out) });
        from_h2w(&out)
    }
    pub fn mul(&self, ctx: Ctx<'_, '_, F>, a: AssignedValue<F>, b: AssignedValue<F>) -> AssignedValue<F> {                         // :59-68
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_mul(ctx.ctx.h2w, &to_h2w(&a), &to_h2w(&b), &mut out) });
        from_h2w(&out)
    }
    pub fn mul_add(&self, ctx: Ctx<'_, '_, F>, a: AssignedValue<F>, b: AssignedValue<F>, c: AssignedValue<F>) -> AssignedValue<F> { // :70-80
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_mul_add(ctx.ctx.h2w, &to_h2w(&a), &to_h2w(&b), &to_h2w(&c), &mut out) });
        from_h2w(&out)
    }
    pub fn select(&self, ctx: Ctx<'_, '_, F>, a: AssignedValue<F>, b: AssignedValue<F>, sel: AssignedValue<F>) -> AssignedValue<F> { // :82-93
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_select(ctx.ctx.h2w, &to_h2w(&a), &to_h2w(&b), &to_h2w(&sel), &mut out) });
        from_h2w(&out)
    }
    pub fn select_from_idx(&self, ctx: Ctx<'_, '_, F>, arr: &[AssignedValue<F>], idx: AssignedValue<F>) -> AssignedValue<F> {       // :95-104
        let a: Vec<H2wAssigned> = arr.iter().map(to_h2w).collect();
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_select_from_idx(ctx.ctx.h2w, a.as_ptr(), a.len(), &to_h2w(&idx), &mut out) });
        from_h2w(&out)
    }
    pub fn select_array_by_indicator(&self, ctx: Ctx<'_, '_, F>, array2d: &[Vec<AssignedValue<F>>], indicator: &[AssignedValue<F>]) -> Vec<AssignedValue<F>> { // :106-115
        let w = array2d.first().map_or(0, |r| r.len());
        let flat: Vec<H2wAssigned> = array2d.iter().flat_map(|r| r.iter().map(to_h2w)).collect();
        let ind: Vec<H2wAssigned> = indicator.iter().map(to_h2w).collect();
        let mut out = vec![H2wAssigned::default(); w];
        ck(unsafe { h2w_select_array_by_indicator(ctx.ctx.h2w, flat.as_ptr(), array2d.len(), w, ind.as_ptr(), out.as_mut_ptr()) });
        out.iter().map(from_h2w).collect()
    }
    pub fn idx_to_indicator(&self, ctx: Ctx<'_, '_, F>, idx: AssignedValue<F>, len: usize) -> Vec<AssignedValue<F>> {               // :117-126
        let mut out = vec![H2wAssigned::default(); len];
        ck(unsafe { h2w_idx_to_indicator(ctx.ctx.h2w, &to_h2w(&idx), len, out.as_mut_ptr()) });
        out.iter().map(from_h2w).collect()
    }
    pub fn num_to_bits(&self, ctx: Ctx<'_, '_, F>, a: AssignedValue<F>, range_bits: usize) -> Vec<AssignedValue<F>> {               // :128-137
        let mut out = vec![H2wAssigned::default(); range_bits];
        ck(unsafe { h2w_num_to_bits(ctx.ctx.h2w, &to_h2w(&a), range_bits, out.as_mut_ptr()) });
        out.iter().map(from_h2w).collect()
    }
    pub fn bits_to_num(&self, ctx: Ctx<'_, '_, F>, bits: &[AssignedValue<F>]) -> AssignedValue<F> {                                 // :139-148
        let b: Vec<H2wAssigned> = bits.iter().map(to_h2w).collect();
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_bits_to_num(ctx.ctx.h2w, b.as_ptr(), b.len(), &mut out) });
        from_h2w(&out)
    }
    pub fn decompose_le(&self, ctx: Ctx<'_, '_, F>, num: AssignedValue<F>, limb_bits: usize, num_limbs: usize) -> Vec<AssignedValue<F>> { // :150-160
        let mut out = vec![H2wAssigned::default(); num_limbs];
        ck(unsafe { h2w_decompose_le(ctx.ctx.h2w, &to_h2w(&num), limb_bits, num_limbs, out.as_mut_ptr()) });
        out.iter().map(from_h2w).collect()
    }
    pub fn limbs_to_num(&self, ctx: Ctx<'_, '_, F>, limbs: &[AssignedValue<F>], limb_bits: usize) -> AssignedValue<F> {             // :162-171
        let l: Vec<H2wAssigned> = limbs.iter().map(to_h2w).collect();
        let mut out = H2wAssigned::default();
        ck(unsafe { h2w_limbs_to_num(ctx.ctx.h2w, l.as_ptr(), l.len(), limb_bits, &mut out) });
        from_h2w(&out)
    }
    pub fn check_less_than_safe(&self, ctx: Ctx<'_, '_, F>, a: AssignedValue<F>, b: u64) {                                          // :173-177
        ck(unsafe { h2w_check_less_than_safe(ctx.ctx.h2w, &to_h2w(&a), b) });
    }
    pub fn range_check(&self, ctx: Ctx<'_, '_, F>, a: AssignedValue<F>, range_bits: usize) {                                        // :179-183
        ck(unsafe { h2w_range_check(ctx.ctx.h2w, &to_h2w(&a), range_bits) });
    }
    pub fn assert_equal(&self, ctx: Ctx<'_, '_, F>, a: AssignedValue<F>, b: AssignedValue<F>) {                                     // :185-192
        ctx.ctx.constrain_equal(&a, &b)
    }
}
