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
