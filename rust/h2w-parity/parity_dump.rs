//! Value-level parity hand-off (SOURCE ONLY: this repository's build environment has no Rust toolchain, the reference's
//! dependencies are unpinned git branches - SURVEY.md 8c - so this file has never met a compiler).
//!
//! What it is for: the advice stream libh2w produces is byte-identical to the CPU oracle's (oracle/oracle.c), and the oracle is
//! pinned to the reference by cell COUNTS only (every frame of verifier/profile/*.svg).  The ORDER of cells inside five halo2-base
//! templates (`select`, `is_zero`, `check_less_than`, `decompose_le`, `sub`) is restated from recollection.  This module,
//! dropped into the reference crate, dumps what the reference's own chips append to `Context::advice` for one proof, in a form
//! `tools/compare_advice.py` of this repository diffs against the oracle's (and, with `--gpu`, libh2w's) stream cell by cell.
//!
//! How to run (someone with cargo and the reference checked out):
//!   1. copy this file to `verifier/src/stark/parity_dump.rs` and add `#[cfg(test)] mod parity_dump;` at the end of
//!      `verifier/src/stark/mod.rs` (it uses that module's test imports: `super::*`);
//!   2. `H2W_PARITY_DIR=/tmp/parity cargo test --release -- --nocapture parity_dump`  (writes bn254/ and gl/ below that directory);
//!   3. in this repository: `python tools/compare_advice.py /tmp/parity/bn254 [--gpu]` and the same for `gl`.
//! A report "IDENTICAL" closes the pin for that hash mode; a mismatch names the #[count] call stack of the first differing cell, and
//! INTEGRATION.md ("Closing the parity pin") lists, per template, the three places that encode its cell order.
//!
//! The flow is `test_fibonacci_stark_bn254` / `_gl` of `verifier/src/stark/mod.rs:405-518` unchanged, plus the three writes at the end.
//! The proof is serialised in the order `WitnessChip::load_proof_with_pis` reads it (`verifier/src/witness/mod.rs:236-294`) = the flat
//! layout of INTEGRATION.md; a hash is 4 little-endian u64 words (Goldilocks-Poseidon: its 4 elements; PoseidonBN254: the canonical
//! little-endian bytes of the field element, `hash/poseidon_bn254/hash.rs:19-21`).
use super::*;

use std::io::Write;

use halo2_base::halo2_proofs::halo2curves::bn256::Fr;
use halo2_base::halo2_proofs::halo2curves::ff::PrimeField;
use halo2_base::utils::testing::base_test;
use plonky2::field::extension::quadratic::QuadraticExtension;
use plonky2::field::goldilocks_field::GoldilocksField;
use plonky2::field::types::{Field as Field_plonky2, PrimeField64};
use plonky2::hash::hash_types::HashOut;
use plonky2::plonk::config::{GenericConfig, GenericHashOut, PoseidonGoldilocksConfig};
use plonky2::util::timing::TimingTree;
use plonky2x::backend::wrapper::plonky2_config::PoseidonBN128GoldilocksConfig;
use starky::config::StarkConfig;
use starky::proof::StarkProofWithPublicInputs;
use starky::prover::prove;

use crate::field::goldilocks::base::GoldilocksChip;
use crate::field::native::NativeChip;
use crate::hash::poseidon::hash::PoseidonChip;
use crate::hash::poseidon_bn254::hash::PoseidonBN254Chip;
use crate::hash::PermutationChip;
use crate::merkle::MerkleTreeChip;
use crate::test_util::fibonacci_stark::FibonacciStark;
use crate::witness::WitnessChip;

type GF = GoldilocksField;
const D: usize = 2;

fn ext(words: &mut Vec<u64>, e: &QuadraticExtension<GF>) {
    words.push(e.0[0].to_canonical_u64());
    words.push(e.0[1].to_canonical_u64());
}

/// One hash = 4 words.  `to_bytes()` of both hash-out types is 32 little-endian bytes: HashOut<GF> = its 4 elements (8 bytes each),
/// PoseidonBN128HashOut = the field element (what `hash_to_fr` feeds to `F::from_bytes_le`).
fn hash<H: GenericHashOut<GF>>(words: &mut Vec<u64>, h: &H) {
    let b = h.to_bytes();
    assert_eq!(b.len(), 32);
    for c in b.chunks(8) {
        words.push(u64::from_le_bytes(c.try_into().unwrap()));
    }
}

/// The proof in `WitnessChip::load_proof_with_pis` order (witness/mod.rs:236-294).
fn flat_words<C: GenericConfig<D, F = GF>>(p: &StarkProofWithPublicInputs<GF, C, D>) -> Vec<u64> {
    let mut w = Vec::new();
    let pr = &p.proof;
    for h in &pr.trace_cap.0 { hash(&mut w, h); }                                   // load_cap(trace_cap)            :245
    for h in &pr.quotient_polys_cap.0 { hash(&mut w, h); }                          // load_cap(quotient_polys_cap)   :246
    let o = &pr.openings;                                                           // load_openings_set              :129-147
    for e in &o.local_values { ext(&mut w, e); }
    for e in &o.next_values { ext(&mut w, e); }
    if let Some(z) = &o.permutation_zs { for e in z { ext(&mut w, e); } }
    if let Some(z) = &o.permutation_zs_next { for e in z { ext(&mut w, e); } }
    for e in &o.quotient_polys { ext(&mut w, e); }
    if let Some(cap) = &pr.permutation_zs_cap { for h in &cap.0 { hash(&mut w, h); } }   // :251-254
    let f = &pr.opening_proof;                                                      // load_fri_proof                 :150-233
    w.push(f.pow_witness.to_canonical_u64());
    for e in &f.final_poly.coeffs { ext(&mut w, e); }
    for cap in &f.commit_phase_merkle_caps { for h in &cap.0 { hash(&mut w, h); } }
    for q in &f.query_round_proofs {
        for (evals, proof) in &q.initial_trees_proof.evals_proofs {
            for x in evals { w.push(x.to_canonical_u64()); }
            for h in &proof.siblings { hash(&mut w, h); }
        }
        for step in &q.steps {
            for e in &step.evals { ext(&mut w, e); }
            for h in &step.merkle_proof.siblings { hash(&mut w, h); }
        }
    }
    for x in &p.public_inputs { w.push(x.to_canonical_u64()); }                     // :285-288
    w
}

fn write_case(dir: &str, hash_mode: usize, degree_bits: usize, config: &StarkConfig, lookup_bits: usize, words: &[u64], advice: &[u8]) {
    std::fs::create_dir_all(dir).unwrap();
    let (arity_bits, final_poly_bits) = match config.fri_config.reduction_strategy {
        plonky2::fri::reduction_strategies::FriReductionStrategy::ConstantArityBits(a, f) => (a, f),
        _ => panic!("the shape struct of include/h2w.h describes ConstantArityBits only"),
    };
    // Fibonacci STARK (test_util/fibonacci_stark.rs:60-61,129-131): 4 columns, one permutation pair in batches of 1 -> 2 Z polys, 2 quotient
    // polys (num_challenges x quotient_degree_factor 1), 3 public inputs
    let case = format!(
        "{{\"degree_bits\": {}, \"rate_bits\": {}, \"cap_height\": {}, \"num_queries\": {}, \"pow_bits\": {}, \"num_challenges\": {}, \
          \"arity_bits\": {}, \"final_poly_bits\": {}, \"n_cols\": 4, \"n_perm_z\": 2, \"n_quotient\": 2, \"n_pis\": 3, \"perm_batch_size\": 1, \
          \"hash_mode\": {}, \"lookup_bits\": {}, \"witness_load_range_check\": 1}}",
        degree_bits, config.fri_config.rate_bits, config.fri_config.cap_height, config.fri_config.num_query_rounds,
        config.fri_config.proof_of_work_bits, config.num_challenges, arity_bits, final_poly_bits, hash_mode, lookup_bits
    );
    std::fs::write(format!("{dir}/case.json"), case).unwrap();
    let mut f = std::fs::File::create(format!("{dir}/proof.words")).unwrap();
    for w in words { f.write_all(&w.to_le_bytes()).unwrap(); }
    std::fs::write(format!("{dir}/advice.bin"), advice).unwrap();
}

/// `ctx.advice` as canonical little-endian bytes (the metric of the reference is its length: util/context_wrapper.rs:24-26).
fn advice_bytes(ctx: &ContextWrapper<Fr>) -> Vec<u8> {
    let mut out = Vec::with_capacity(ctx.ctx.advice.len() * 32);
    for a in ctx.ctx.advice.iter() {
        out.extend_from_slice(a.evaluate().to_repr().as_ref());       // Assigned<Fr> -> Fr (Rational cells: the is_zero inverse hints) -> canonical LE
    }
    out
}

fn fibonacci<F: Field_plonky2>(n: usize, x0: F, x1: F) -> F {
    (0..n).fold((x0, x1), |x, _| (x.1, x.0 + x.1)).1
}

fn out_dir(sub: &str) -> String {
    format!("{}/{}", std::env::var("H2W_PARITY_DIR").unwrap_or_else(|_| "parity".to_string()), sub)
}

/// `log2(num_rows)`: 3 is the reference's own test size (no FRI fold step, empty Merkle paths); 7 and up exercise the fold loop
/// (`fri/mod.rs:403-438`), which no reference test reaches: 7 has one arity-16 step, 11 has TWO (the BASELINE arity with a second, shorter
/// Merkle path per query and `x <- x^16` applied between them: `fri/mod.rs:433-437`).
const DEGREE_BITS: &[usize] = &[3, 7, 11];

#[test]
fn parity_dump_bn254() {
    type C = PoseidonBN128GoldilocksConfig;
    type F = <C as GenericConfig<D>>::F;
    type S = FibonacciStark<F, D>;
    for &db in DEGREE_BITS {
        let config = StarkConfig::standard_fast_config();
        let num_rows = 1 << db;
        let public_inputs = [F::ZERO, F::ONE, fibonacci(num_rows - 1, F::ZERO, F::ONE)];
        let stark = S::new(num_rows);
        let trace = stark.generate_trace(public_inputs[0], public_inputs[1]);
        let proof_with_pis = prove::<F, C, S, D>(stark, &config, trace, &public_inputs, &mut TimingTree::default()).unwrap();
        let words = flat_words(&proof_with_pis);
        let k = 22;
        base_test().k(k).run(|ctx, range| {
            let ctx = &mut ContextWrapper::new(ctx);
            let native = NativeChip::<Fr>::new(range.clone());
            let goldilocks_chip = GoldilocksChip::new(native.clone());
            let extension_chip = GoldilocksQuadExtChip::new(goldilocks_chip.clone());
            let poseidon_chip = PoseidonChip::new(goldilocks_chip.clone());
            let poseidon_bn254_chip = PoseidonBN254Chip::new(native.clone());
            let merkle_chip = MerkleTreeChip::new(goldilocks_chip.clone(), poseidon_bn254_chip.clone());
            let permutation_chip = poseidon_chip.permutation_chip();
            let state = permutation_chip.load_zero(ctx);
            let challenger_chip = ChallengerChip::new(permutation_chip.clone(), state);
            let fri_chip = FriChip::new(extension_chip, merkle_chip);
            let mut stark_chip = StarkChip::new(challenger_chip, fri_chip);
            let witness_chip = WitnessChip::new(goldilocks_chip, poseidon_bn254_chip);
            let proof_wire = witness_chip.load_proof_with_pis(ctx, proof_with_pis.clone());
            stark_chip.verify_proof(ctx, stark, proof_wire, &config);
            write_case(&out_dir(&format!("bn254_d{db}")), 1, db, &config, k - 1, &words, &advice_bytes(ctx));
        });
    }
}

#[test]
fn parity_dump_gl() {
    type C = PoseidonGoldilocksConfig;
    type F = <C as GenericConfig<D>>::F;
    type S = FibonacciStark<F, D>;
    for &db in DEGREE_BITS {
        let config = StarkConfig::standard_fast_config();
        let num_rows = 1 << db;
        let public_inputs = [F::ZERO, F::ONE, fibonacci(num_rows - 1, F::ZERO, F::ONE)];
        let stark = S::new(num_rows);
        let trace = stark.generate_trace(public_inputs[0], public_inputs[1]);
        let proof_with_pis = prove::<F, C, S, D>(stark, &config, trace, &public_inputs, &mut TimingTree::default()).unwrap();
        let words = flat_words(&proof_with_pis);
        let k = 22;
        base_test().k(k).run(|ctx, range| {
            let ctx = &mut ContextWrapper::new(ctx);
            let native = NativeChip::<Fr>::new(range.clone());
            let goldilocks_chip = GoldilocksChip::new(native.clone());
            let extension_chip = GoldilocksQuadExtChip::new(goldilocks_chip.clone());
            let poseidon_chip = PoseidonChip::new(goldilocks_chip.clone());
            let merkle_chip = MerkleTreeChip::new(goldilocks_chip.clone(), poseidon_chip.clone());
            let permutation_chip = poseidon_chip.permutation_chip();
            let state = permutation_chip.load_zero(ctx);
            let challenger_chip = ChallengerChip::new(permutation_chip.clone(), state);
            let fri_chip = FriChip::new(extension_chip, merkle_chip);
            let mut stark_chip = StarkChip::new(challenger_chip, fri_chip);
            let witness_chip = WitnessChip::new(goldilocks_chip, poseidon_chip);
            let proof_wire = witness_chip.load_proof_with_pis(ctx, proof_with_pis.clone());
            stark_chip.verify_proof(ctx, stark, proof_wire, &config);
            write_case(&out_dir(&format!("gl_d{db}")), 0, db, &config, k - 1, &words, &advice_bytes(ctx));
        });
    }
}

// (unused import guards for the variants of halo2-base / plonky2 that re-export these under other paths)
#[allow(dead_code)]
fn _types(_: HashOut<GF>, _: Fr) {}
#[allow(dead_code)]
fn _repr_is_canonical_le() { let _ = <Fr as PrimeField>::NUM_BITS; }
