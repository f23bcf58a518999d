"""Code-generation invariants of the two hot kernels, checked on the cross-compiled gfx950 assembly (no GPU needed).

On this family loads and stores retire in order on one counter (vmcnt): a scratch or flat access inside a round loop makes the wavefront
wait for every cell store it has in flight.  Twice in round 2 a source change that "did not touch the arithmetic" (a struct select, a
refactoring of the Montgomery product into helper functions) put spill reloads into the PoseidonBN254 partial-round loop and cost 3-8 % of the
chain kernel; this test is the guard."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "halo2-plonky2-verifier_amd", "csrc")
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-unused-value -x hip -S --cuda-device-only".split()


def _asm(src, tmp_path):
    if not shutil.which("hipcc"):
        pytest.skip("no hipcc")
    out = os.path.join(str(tmp_path), src + ".s")
    subprocess.run(["hipcc", *FLAGS, os.path.join(CSRC, src), "-o", out], check=True, capture_output=True, cwd=CSRC)
    return open(out).read().split("\n")


def _function(lines, mangled_prefix):
    start = next(i for i, l in enumerate(lines) if re.match("^" + re.escape(mangled_prefix) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if "s_setpc_b64" in lines[i] or "s_endpgm" in lines[i])
    return start, end


def _loops(lines, lo, hi):
    """(first, last) line of every natural loop = backward branch (conditional or not) to a label inside [lo, hi)."""
    labels = {m.group(1): i for i in range(lo, hi) for m in [re.match(r"^(\.LBB\d+_\d+):", lines[i])] if m}
    for i in range(lo, hi):
        m = re.search(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            yield labels[m.group(1)], i


def _count(lines, a, b, pat):
    return sum(1 for l in lines[a:b + 1] if re.match(r"^\s+" + pat, l))


def _clean(lines, a, b, what):
    bad = _count(lines, a, b, "scratch_") + _count(lines, a, b, "flat_")
    assert bad == 0, f"{bad} scratch/flat accesses in the {what} (asm lines {a}-{b})"


def test_chain_round_loops_have_no_scratch_or_flat_accesses(tmp_path):
    lines = _asm("batch.hip", tmp_path)
    # one-pass kernel (k_merkle_bn_fused): the emitter walks the S-box chain - four products (644 multiply-adds) per partial round
    lo, hi = _function(lines, "_ZN3h2w9QuadSinkTILb0ELi3EE13bn_emit_cellsILb0EE")
    loops = list(_loops(lines, lo, hi))
    partial = [(a, b) for a, b in loops if 630 <= _count(lines, a, b, "v_mad_u64_u32") <= 660]
    assert partial, "partial-round loop (four products, 644 multiply-adds) not found"
    a, b = min(partial, key=lambda ab: ab[1] - ab[0])
    _clean(lines, a, b, "one-pass partial-round loop")
    n = sum(1 for l in lines[a:b + 1] if re.match(r"^\s+[a-z]", l))
    assert n <= 1950, f"partial round grew to {n} instructions"
    full = [(c, d) for c, d in loops if (d < a or c > b) and 700 <= _count(lines, c, d, "v_mad_u64_u32") <= 1100]
    assert full, "full-round loop not found"
    for c, d in full:
        _clean(lines, c, d, "one-pass full-round loop")
    # emission kernel of the two-pass paths (k_merkle_bn_emit): two products per partial round, rounds in pairs (644 multiply-adds per trip),
    # the S-box values requested a pair ahead: no vector-memory wait but the two that admit the next pair's loads.  (The kernel lives in glue.hip.)
    glue = _asm("glue.hip", tmp_path)
    lo, hi = _function(glue, "_ZN3h2w9QuadSinkTILb0ELi2EE13bn_emit_cellsILb1EE")
    loops = list(_loops(glue, lo, hi))
    pair = [(a, b) for a, b in loops if 630 <= _count(glue, a, b, "v_mad_u64_u32") <= 660]
    assert pair, "two-round loop (four products) not found"
    a, b = min(pair, key=lambda ab: ab[1] - ab[0])
    _clean(glue, a, b, "emission partial-round loop")
    waits = [int(m.group(1)) for l in glue[a:b + 1] for m in [re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)] if m]
    assert waits and min(waits) >= 30, f"a vector-memory wait in the emission loop drains the store queue: vmcnt {waits}"
    # values kernel (k_merkle_bn_values, glue.hip too): no scratch in its round loops either
    lo, hi = _function(glue, "_ZN3h2w9QuadSinkTILb0ELi1EE9bn_values")
    for c, d in _loops(glue, lo, hi):
        if _count(glue, c, d, "v_mad_u64_u32") >= 400:
            _clean(glue, c, d, "values-pass round loop")


def test_expand_fast_has_no_scratch(tmp_path):
    lines = _asm("expand.hip", tmp_path)
    for variant in ("_ZN3h2w11expand_fastILi21ELb0ELb0EE", "_ZN3h2w11expand_fastILi21ELb1ELb0EE", "_ZN3h2w11expand_fastILi21ELb1ELb1EE"):      # static grid / roaming wavefronts
        lo, hi = _function(lines, variant)
        assert _count(lines, lo, hi, "scratch_") == 0
        assert _count(lines, lo, hi, "flat_store") == 0      # every cell store is a global_store


def test_row_cooperative_values_pass_round_loops(tmp_path):
    """k_merkle_bn_values_row (glue.hip; rowfr.h / rowperm.h): the partial-round loop is three lane-cooperative products (3 x 29 multiply-adds) without a scratch or
    flat access and without a vector-memory wait (its only memory traffic are the three S-box stores); its table entries are read a round ahead (the LDS reads
    stand at the top of the loop); it stays near the instruction count its 1.3 ms correspond to."""
    glue = _asm("glue.hip", tmp_path)
    lo, hi = _function(glue, "_ZN3h2w22k_merkle_bn_values_row")      # (permute_unit is inlined into the flattened kernel since round 4: no call, no entry wait for the previous unit's state store)
    loops = list(_loops(glue, lo, hi))
    partial = [(a, b) for a, b in loops if 85 <= _count(glue, a, b, "v_mad_u64_u32") <= 90]
    assert partial, "partial-round loop (three products of 29 multiply-adds) not found"
    a, b = min(partial, key=lambda ab: ab[1] - ab[0])
    _clean(glue, a, b, "row-cooperative partial-round loop")
    assert _count(glue, a, b, "global_load") == 0
    assert not [l for l in glue[a:b + 1] if re.search(r"s_waitcnt.*vmcnt\(0\)", l)], "the row-cooperative partial round waits for its S-box stores"
    n = sum(1 for l in glue[a:b + 1] if re.match(r"^\s+[a-z]", l))
    assert n <= 540, f"partial round grew to {n} instructions"
    first_mad = next(i for i in range(a, b + 1) if re.match(r"^\s+v_mad_u64_u32", glue[i]))
    assert _count(glue, a, first_mad, "ds_read") >= 6, "the round's table entries are no longer read ahead of its first product"
    full = [(c, d) for c, d in loops if (d < a or c > b) and 195 <= _count(glue, c, d, "v_mad_u64_u32") <= 215]
    assert full, "full-round loop (seven products) not found"
    for c, d in full:
        _clean(glue, c, d, "row-cooperative full-round loop")


def test_replay_interpreter_reads_its_tape_with_scalar_loads(tmp_path):
    """k_replay (replay.hip): the tape pointer must stay uniform - without the readfirstlane on the template's fields every tape word was a vector load behind the
    record stores in flight (2.5 us per op) and every branch a lane mask; operands come from LDS (the ring and the pools), far ones through the one out-of-line path."""
    lines = _asm("replay.hip", tmp_path)
    lo, hi = _function(lines, "_ZN3h2w8k_replay")
    end = next(i for i in range(lo, len(lines)) if "s_endpgm" in lines[i])
    assert _count(lines, lo, end, "s_load_dword") >= 50, "the interpreter no longer reads its tape with scalar loads"
    assert _count(lines, lo, end, "global_load") + _count(lines, lo, end, "flat_load") <= 40, "vector loads crept into the interpreter (tape words? operands outside LDS?)"
    assert _count(lines, lo, end, "ds_read") >= 50


def test_goldilocks_values_permutation_stays_at_its_instruction_count(tmp_path):
    """glp_permute_lanes (glperm.h: the values phase's Goldilocks permutation, one wavefront, the floor of the prologue): a lone wavefront pays ~4.5 cycles per
    instruction AND per wait state, so the count is the time.  The partial-round loop (two rounds per trip) is four products and two lane swaps a round;
    the function touches memory twice (the list word on the way in, nothing else), never scratch; a row of the MDS layer reads its 24 lane values
    without a wait state between them (the hand-scheduled blocks: left to the compiler there were 24); the linear layer before the partial rounds is three
    such blocks of twelve."""
    lines = _asm("glue.hip", tmp_path)
    lo, hi = _function(lines, "_ZN3h2w17glp_permute_lanes")
    _clean(lines, lo, hi, "Goldilocks values permutation")
    assert _count(lines, lo, hi, "global_store") == 1 and _count(lines, lo, hi, "global_load") == 0
    assert not [l for l in lines[lo + 4:hi - 2] if re.search(r"s_waitcnt.*vmcnt", l)], "a vector-memory wait inside the permutation"
    loops = list(_loops(lines, lo, hi))
    partial = [(a, b) for a, b in loops if _count(lines, a, b, "v_permlane16_swap") == 4 and _count(lines, a, b, "v_readlane") <= 8]
    assert partial, "partial-round loop (two rounds: four lane swaps, four v_readlane) not found"
    a, b = min(partial, key=lambda ab: ab[1] - ab[0])
    valu = _count(lines, a, b, "v_")
    waits = sum(int(m.group(1)) + 1 for l in lines[a:b + 1] for m in [re.match(r"^\s+s_nop\s+(\d+)", l)] if m)
    assert valu <= 200 and valu + waits <= 240, f"two partial rounds grew to {valu} vector instructions and {waits} wait states"
    starts = [i for i in range(lo, hi) if re.match(r"^\s+v_readlane_b32 s20, v\d+, 0", lines[i])]
    ends = [next(j for j in range(i, hi + 1) if "#ASMEND" in lines[j]) for i in starts]
    rows = [sum(1 for l in lines[i:e] if re.match(r"^\s+v_readlane", l)) for i, e in zip(starts, ends)]
    assert rows.count(24) >= 2, "the hand-scheduled MDS rows (first and second half) are gone"
    assert rows.count(12) == 3, "the three limb passes of the linear layer before the partial rounds (glq_dense12) are gone"
    for i, e in zip(starts, ends):
        last = max(j for j in range(i, e) if re.match(r"^\s+v_readlane", lines[j]))
        assert sum(1 for l in lines[i:last] if re.match(r"^\s+s_nop", l)) == 0, "a wait state between the v_readlanes of a hand-scheduled block"
