"""The lazy nine-limb Montgomery arithmetic of the values pass (csrc/field.h fr9_t), checked on the host against the canonical product:
products of un-reduced representatives up to (and past) the sizes the pass forms, sums of two as operands, normalisation, canonicalisation.
field.h compiles as plain C++; the GPU parity tests check the pass end to end, this pins its arithmetic and its bounds."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lazy_limb_products_agree_with_the_canonical_product(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = os.path.join(str(tmp_path), "fr9_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "halo2-plonky2-verifier_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "fr9_check.cpp"), "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout
    bits = int(r.stdout.split("widest product:")[1].split()[0])
    assert bits <= 261


def test_lane_cooperative_products_and_permutation_on_simulated_lanes(tmp_path):
    """csrc/rowfr.h + csrc/rowperm.h (the values pass with one 29-bit limb per lane, k_merkle_bn_values_row) with the 64 lanes of a wavefront
    simulated on the host: products of lazy operands against the canonical product, limb and value bounds, no wrapped column sum, the carry rule of
    the exact division, and the whole PoseidonBN254 permutation (output state and the 168 S-box values) against the reference walk."""
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = os.path.join(str(tmp_path), "rowfr_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "halo2-plonky2-verifier_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "rowfr_check.cpp"), "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    assert int(r.stdout.split("widest limb:")[1].split()[0]) < (1 << 29) + 8


def test_fri_constants_table_equals_the_per_call_formulas(tmp_path):
    """chips.h FriTab (what h2w_plan_compile tabulates for the device strands) against the formulas the reference evaluates at every call."""
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = os.path.join(str(tmp_path), "fri_tab_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "halo2-plonky2-verifier_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "fri_tab_check.cpp"), "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_partial_round_accumulator_tables_reproduce_the_fast_partial_rounds(tmp_path):
    """csrc/glptab.h (what h2w_plan_compile uploads behind the Poseidon constants for the values phase, csrc/glperm.h): the partial rounds with one
    accumulator per round instead of one row sum per round give the state the reference walk gives, on tiny-entry and full-width tables."""
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = os.path.join(str(tmp_path), "glperm_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "halo2-plonky2-verifier_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "glperm_check.cpp"), "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_assembly_reductions_restated_on_the_host(tmp_path):
    """csrc/glperm.h glq_reduce / glq_reduce96 (written in assembly for the device) restated instruction by instruction in integer arithmetic: the algorithm -
    one multiply-add folds the 2^64 word, its carry and the 2^96 word's borrow become one correction, no second wrap - against x mod p on edge values, on the rare
    borrow and carry-and-borrow cases, on products and on the sums of the MDS / linear layers."""
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = os.path.join(str(tmp_path), "glq_reduce_check")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "glq_reduce_check.cpp"), "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
