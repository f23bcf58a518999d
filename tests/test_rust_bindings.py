"""the Rust side of the boundary is shipped as source (no Rust toolchain here).  What can be checked without
rustc: the generated `extern "C"` block of rust/h2w-sys declares every function of include/h2w.h with the same number of
parameters and is up to date with the header; the hand-written NativeChip shim calls only declared functions and covers every
method of the reference's NativeChip (field/native.rs:28-193)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip(src):
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def test_sys_crate_is_generated_from_the_header():
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_bindings.py"), "--check"])
    hdr = _strip(open(os.path.join(ROOT, "include", "h2w.h")).read())
    hdr = hdr[hdr.index('extern "C" {'):]
    c_protos = {m.group(1): m.group(2) for m in re.finditer(r"\b(h2w_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S)}
    rs = open(os.path.join(ROOT, "rust", "h2w-sys", "src", "lib.rs")).read()
    rs_protos = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (h2w_[a-z0-9_]+)\((.*?)\)(?: -> [^;]+)?;", rs)}
    assert set(c_protos) == set(rs_protos), set(c_protos) ^ set(rs_protos)
    for name, params in c_protos.items():
        n_c = 0 if params.strip() in ("", "void") else params.count(",") + 1
        n_rs = 0 if not rs_protos[name].strip() else rs_protos[name].count(",") + 1
        assert n_c == n_rs, (name, params, rs_protos[name])


def test_native_chip_shim_covers_the_reference_interface():
    rs = open(os.path.join(ROOT, "rust", "h2w-native", "src", "lib.rs")).read()
    sys_fns = set(re.findall(r"pub fn (h2w_[a-z0-9_]+)\(", open(os.path.join(ROOT, "rust", "h2w-sys", "src", "lib.rs")).read()))
    used = set(re.findall(r"\b(h2w_[a-z0-9_]+)\s*[\(,]", rs)) - {"h2w_native", "h2w_sys"}
    assert used and used <= sys_fns, used - sys_fns
    methods = set(re.findall(r"pub fn ([a-z_0-9]+)\(", rs))
    reference = {"load_constant", "load_zero", "load_constants", "load_witness", "add", "mul", "mul_add", "select", "select_from_idx",
                 "select_array_by_indicator", "idx_to_indicator", "num_to_bits", "bits_to_num", "decompose_le", "limbs_to_num",
                 "check_less_than_safe", "range_check", "assert_equal", "num_cells", "push_context", "pop_context"}
    assert reference <= methods, reference - methods


# the shapes the reference's sources actually use at the boundary (file:line in /root/reference/verifier/src), as patterns the shim
# must contain; kept as data here because the shim cannot be compiled in this environment (no Rust toolchain)
CALL_SHAPES = [
    ("util/context_wrapper.rs:11-14  pub struct ContextWrapper<'ctx, F> { pub ctx: &'ctx mut Context<F>, .. }",
     r"pub struct ContextWrapper<'ctx, F: BigPrimeField>\s*\{\s*pub ctx: &'ctx mut Context<F>,"),
    ("util/context_wrapper.rs:17     pub fn new(ctx: &'ctx mut Context<F>) -> Self", r"pub fn new\(ctx: &'ctx mut Context<F>\) -> Self"),
    ("util/context_wrapper.rs:24-26  self.ctx.advice.len()", r"self\.ctx\.advice\.len\(\)"),
    ("util/context_wrapper.rs:28     pub fn push_context(&mut self, level: log::Level, ctx: &str)", r"pub fn push_context\(&mut self, _?level: log::Level, ctx: &str\)"),
    ("util/context_wrapper.rs:32     pub fn pop_context(&mut self)", r"pub fn pop_context\(&mut self\)"),
    ("field/native.rs:15-17          pub fn new(range_chip: RangeChip<F>) -> Self", r"pub fn new\(range_chip: RangeChip<F>\) -> Self"),
    ("field/native.rs:19-21          pub fn gate_chip(&self) -> &GateChip<F>", r"pub fn gate_chip\(&self\) -> &GateChip<F>"),
    ("field/native.rs:23-25          pub fn range_chip(&self) -> &RangeChip<F>", r"pub fn range_chip\(&self\) -> &RangeChip<F>"),
    ("field/native.rs:28-31          ctx.ctx.load_constant(a)", r"ctx\.ctx\.load_constant\(a\)"),
    ("field/native.rs:33-36          ctx.ctx.load_zero()", r"ctx\.ctx\.load_zero\(\)"),
    ("field/native.rs:38-41          ctx.ctx.load_constants(c)", r"ctx\.ctx\.load_constants\(c\)"),
    ("field/native.rs:43-46          ctx.ctx.load_witness(a)", r"ctx\.ctx\.load_witness\(a\)"),
    ("field/native.rs:189-192        ctx.ctx.constrain_equal(&a, &b)", r"ctx\.ctx\.constrain_equal\(&a, &b\)"),
    ("hash/poseidon_bn254/permutation.rs:88-104  ctx.ctx.load_constant(from_fr(..)): Context::load_constant(&mut self, c: F)", r"pub fn load_constant\(&mut self, c: F\) -> AssignedValue<F>"),
    ("stark/mod.rs:427-428,483-484   base_test().k(k).run(|ctx, range| ..)", r"pub fn base_test\(\) -> BaseTester"),
    ("stark/mod.rs:483-484           .run(|ctx, range| ..): closure over (&mut Context<F>, &RangeChip<F>)", r"pub fn run<F: BigPrimeField, R>\(&self, f: impl FnOnce\(&mut Context<F>, &RangeChip<F>\) -> R\) -> R"),
    ("stark/mod.rs:488               NativeChip::<Fr>::new(range.clone()): RangeChip is Clone", r"#\[derive\(Clone, Debug\)\]\s*pub struct RangeChip<F: BigPrimeField>"),
]


def test_shim_has_the_reference_call_shapes():
    rs = open(os.path.join(ROOT, "rust", "h2w-native", "src", "lib.rs")).read()
    missing = [what for what, pat in CALL_SHAPES if not re.search(pat, rs)]
    assert not missing, missing
    cargo = open(os.path.join(ROOT, "rust", "h2w-native", "Cargo.toml")).read()
    assert re.search(r"^log\s*=", cargo, flags=re.M)                     # log::Level is part of push_context's signature
    # every NativeChip method takes `&mut ContextWrapper<F>` (the alias Ctx<'_, '_, F>) as the reference's do
    assert "type Ctx<'a, 'ctx, F> = &'a mut ContextWrapper<'ctx, F>;" in rs
