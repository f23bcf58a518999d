#!/usr/bin/env python3
"""bench.py — FRI-verifier witness-generation throughput on MI355X (BASELINE.json metric).

One *step* = one pass of the hot path (h2w_fri_witness_batch: value kernels + expansion kernel) over one batch of
synthetic proofs that are already resident in HBM.  Default workload = BASELINE.json configs[2], the north-star
target config (2^20-row Fibonacci STARK, 28 FRI queries, cap_height 4, PoseidonBN254 Merkle caps), `--batch`
proofs per GPU per step.  `--config cfg2|cfg1`, `--hash gl` select the other configs.

N>1 (launched by torch.distributed.run, one rank per GPU): rank 0 builds the proofs and broadcasts the proof block
over RCCL (the only collective: SURVEY §8e); every rank then generates the witness of its own shard of proofs —
no data-path collective — so per-GPU work is fixed ("weak" scaling) and value = all ranks' cells / max-rank time.

The synthetic inputs (proof words, Poseidon tables) are drawn here with numpy; the CPU oracle (oracle/) is imported by the
`cpu_baseline` leg only (rank 0, N=1), never by the measured path.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# several batches are in flight on separate HIP streams; ROCm maps streams onto 4 hardware queues by default, which would
# serialise them pairwise.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

CONFIGS = {
    # name: (degree_bits, num_queries, rate_bits, description)
    "cfg1": (10, 4, 1, "2^10-row Fibonacci STARK, 4 FRI queries, rate_bits=1, cap_height=4"),
    "cfg2": (16, 28, 2, "2^16-row Fibonacci STARK, 28 FRI queries, rate_bits=2, cap_height=4"),
    "cfg3": (20, 28, 1, "2^20-row Fibonacci STARK, 28 FRI queries, rate_bits=1, cap_height=4"),
    "cfg5": (20, 84, 1, "2^20-row Fibonacci STARK, 84 FRI queries, rate_bits=1, cap_height=4"),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(shape_args, hash_mode, lookup_bits, budget_s=12.0):
    """The oracle (CPU restatement, kind "port") timed single-threaded on this host, on the same workload shape."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as O
    sh = O.fibonacci_shape(shape_args[0], shape_args[1], rate_bits=shape_args[2], hash_mode=hash_mode, lookup_bits=lookup_bits)
    k = O.published_consts()
    n, cells, t0 = 0, 0, time.perf_counter()
    while True:
        pr = O.synth_proof(sh, 0xF1B00000 + n)
        ctx = O.Ctx(lookup_bits)
        if n > 0:
            ctx.reserve(cells // n)          # known size after the first proof: no realloc copies in the timed run
        t1 = time.perf_counter()
        O.verify_stark(ctx, sh, k, pr)
        dt = time.perf_counter() - t1
        cells += ctx.num_cells(); n += 1
        ctx.close()
        if n == 1:
            first = dt
        if time.perf_counter() - t0 > budget_s or n >= 64:
            break
    total = time.perf_counter() - t0
    return {"value": cells / total, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": f"{n} proof(s) of the same shape ({cells} cells) through oracle/liboracle.so, 1 thread, {total:.1f} s wall incl. proof synthesis",
            "proofs_per_s": n / total}


def cpu_baseline_all_cores(shape_args, hash_mode, lookup_bits, cells_per_proof, budget_s=10.0):
    """The same oracle loop in one process per host core (one proof per core; the reference itself is single-threaded,
    so this is the most a user of it could get from the box).  Workers are fresh interpreters that never touch the GPU; their number
    is bounded by memory (a context holds its whole advice stream)."""
    import subprocess
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 32 << 30
    share = int(os.environ.get("H2W_CPU_SHARE", "16"))      # the GPU box gives one GPU's job a 16-core CPU share, whatever its affinity mask shows
    workers = max(1, min(ncores, share, int(0.5 * avail // (cells_per_proof * 40))))
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", json.dumps([list(shape_args), hash_mode, lookup_bits, budget_s])]
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True) for _ in range(workers)]
    res = [json.loads(p.communicate(timeout=budget_s * 6 + 120)[0].strip().splitlines()[-1]) for p in procs]
    wall = time.perf_counter() - t0
    cells = sum(r["cells"] for r in res); n = sum(r["proofs"] for r in res); span = max(r["seconds"] for r in res)
    return {"value": cells / span, "unit": "cells/s", "cores": workers, "host_cores_visible": ncores, "cpu_share": share, "kind": "port",
            "sample": f"{n} proof(s) of the same shape over {workers} single-threaded oracle processes ({span:.1f} s of work each, {wall:.1f} s wall incl. start-up)",
            "proofs_per_s": n / span}


def _cpu_worker(spec):
    shape_args, hash_mode, lookup_bits, budget_s = json.loads(spec)
    r = cpu_baseline(tuple(shape_args), hash_mode, lookup_bits, budget_s)
    m = re.match(r"(\d+) proof", r["sample"])
    n = int(m.group(1))
    print(json.dumps({"cells": r["value"] * n / r["proofs_per_s"], "proofs": n, "seconds": n / r["proofs_per_s"]}))


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-worker":
        return _cpu_worker(sys.argv[2])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--hash", default="bn254", choices=["bn254", "gl"])
    ap.add_argument("--batch", type=int, default=0, help="proofs per GPU per step (0 = auto)")
    ap.add_argument("--lookup-bits", type=int, default=21)
    ap.add_argument("--streams", type=int, default=0, help="batches in flight, each on its own HIP stream with its own advice/workspace buffers (0 = auto)")
    ap.add_argument("--advice-cap-gb", type=float, default=200.0, help="upper bound on the advice buffers of all batches in flight (the stream count is reduced to fit)")
    ap.add_argument("--calib", type=int, default=5, help="isolated single-stream launches after the timed region for the roofline numbers")
    ap.add_argument("--proofs", default="valid", choices=["valid", "random"], help="synthetic inputs: valid FRI instances generated on the GPU by the ingest rank (h2w_prove_fri_batch, SURVEY 8d variant A) or uniform random words (variant B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fork", action="store_true", help="experiment: PoseidonBN254 chain kernels on the caller's stream (H2W_OPT_FORK_CHAINS = 0)")
    ap.add_argument("--cu-split", type=int, default=0, help="experiment: CU-masked streams, value strands on the first N CUs, expansion on the rest")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for rehearsing the N>1 logic on one GPU)")
    ap.add_argument("--share-gpu0", action="store_true", help="rehearsal only: every rank uses cuda:0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    h2w = importlib.import_module("halo2-plonky2-verifier_amd")
    api = importlib.import_module("halo2-plonky2-verifier_amd.api")
    D = importlib.import_module("halo2-plonky2-verifier_amd.distributed")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.share_gpu0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    d, q, rb, desc = CONFIGS[args.config]
    hash_mode = 1 if args.hash == "bn254" else 0
    shape = h2w.fibonacci_shape(d, q, rate_bits=rb, hash_mode=hash_mode, lookup_bits=args.lookup_bits)
    # Poseidon constants: the published plonky2 / circomlib parameter sets the reference links in (h2w_poseidon_published;
    # pinned to published known-answer vectors in tests/test_poseidon_published.py).  The oracle is not involved in the
    # product path: it is imported by the cpu_baseline leg only.
    import ctypes as C
    import numpy as np
    consts = h2w.published_consts()
    plan = api.Plan(shape, consts, local_rank)
    if args.no_fork:
        plan.configure(1, 0)

    cell_bytes = plan.num_cells * 32
    if args.batch > 0:
        B = args.batch
    else:   # auto: ~29.5 GB (BN254 Merkle) / ~58 GB (GL Merkle) of advice per GPU per step
        B = max(1, min(64, int((29.5e9 if hash_mode == 1 else 58e9) // cell_bytes)))
    total_proofs = B * world

    # ---- inputs: rank 0 synthesises all proofs, one RCCL broadcast moves the proof block
    words = plan.proof_words
    if args.proofs == "valid" and args.backend != "gloo":
        # valid FRI instances (SURVEY 8d variant (A)): random committed polynomials, proved on the ingest rank's GPU in lockstep batches;
        # the proofs never visit the host.  Every distinct proof is used (no repetition inside a step).
        all_proofs = torch.zeros(total_proofs * words, dtype=torch.int64, device=dev)
        gen_seconds = None
        if rank == 0:
            t_gen = time.perf_counter()
            pr = api.Prover(shape, consts, local_rank)
            assert pr.proof_words == words
            gen = torch.Generator(device=dev); gen.manual_seed(0xF1B00000)
            chunk = max(1, min(total_proofs, (8 << 21) >> (d + rb)))              # ~5 GB of prover scratch at a time
            for first in range(0, total_proofs, chunk):
                nb = min(chunk, total_proofs - first)
                coefs = torch.randint(0, 1 << 62, (nb * pr.num_polys << d,), dtype=torch.int64, device=dev, generator=gen)   # < 2^62 < p: canonical
                pr.prove_batch(coefs.data_ptr(), [1, 1, 2] * nb, all_proofs[first * words:].data_ptr(), nb, torch.cuda.current_stream(dev).cuda_stream)
            torch.cuda.synchronize(dev)
            gen_seconds = time.perf_counter() - t_gen
            pr.close(); del coefs
            torch.cuda.empty_cache()
        D.broadcast_proofs(all_proofs, src=0)       # the only collective (RCCL over xGMI): ingest rank -> all ranks
    else:
        host = torch.empty(total_proofs * words, dtype=torch.int64)
        if rank == 0:      # uniform random proof words (witness generation does not branch on validity, SURVEY 8d variant (B)); every word < 2^60:
            prng = np.random.default_rng(0xF1B00000)     # Goldilocks words canonical (< p), every 4-word hash a canonical Fr (< 2^252 < r)
            host[:] = torch.from_numpy(prng.integers(0, 1 << 60, total_proofs * words, dtype=np.int64))
        if args.backend == "gloo":                      # rehearsal: broadcast on the host, then upload
            D.broadcast_proofs(host, src=0)
            all_proofs = host.to(dev)
        else:
            all_proofs = host.to(dev)
            D.broadcast_proofs(all_proofs, src=0)       # the only collective (RCCL over xGMI): ingest rank -> all ranks
    lo, hi = D.shard_range(total_proofs, world, rank)
    assert hi - lo == B
    my_proofs = all_proofs[lo * words:hi * words]

    # S batches in flight: step k runs on stream k % S into its own advice / workspace buffers, so the latency-bound
    # value strands of one batch (serial Fiat-Shamir sponge, Merkle chains) overlap the HBM-bound kernels of another.
    S = args.streams if args.streams > 0 else (6 if hash_mode == 1 else 3)
    while S > 1 and S * B * cell_bytes > args.advice_cap_gb * 1e9:   # stay inside the 288 GB of HBM
        S -= 1
    advices = [torch.empty(B * cell_bytes, dtype=torch.uint8, device=dev) for _ in range(S)]
    wss = [torch.empty(plan.workspace_bytes(B), dtype=torch.uint8, device=dev) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    emit_streams = [None] * S
    if args.cu_split > 0:      # hipExtStreamCreateWithCUMask: value strands on CUs [0, N), streaming kernel on [N, 256)
        hip = C.CDLL("libamdhip64.so")
        def masked(lo, hi):
            words = (C.c_uint32 * 8)()
            for cu in range(lo, hi):
                words[cu // 32] |= 1 << (cu % 32)
            st = C.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
            assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
            return torch.cuda.ExternalStream(st.value, device=dev)
        streams = [masked(0, args.cu_split) for _ in range(S)]
        emit_streams = [masked(args.cu_split, 256) for _ in range(S)]
    torch.cuda.synchronize()
    counter = [0]

    def step():
        i = counter[0] % S; counter[0] += 1
        plan.run(my_proofs.data_ptr(), B, advices[i].data_ptr(), wss[i].data_ptr(), streams[i].cuda_stream,
                 emit_streams[i].cuda_stream if emit_streams[i] is not None else None)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = D.max_over_ranks(elapsed, torch.device("cpu") if args.backend == "gloo" else dev)
    for i in range(S):
        status = plan.status(wss[i].data_ptr(), B, streams[i].cuda_stream)
        assert status == [0] * B or os.environ.get("H2W_DBG_SKIP_KERNELS") or os.environ.get("H2W_DBG_SKIP_ALT"), f"device status {status}"   # (the skip hooks exist in H2W_DEBUG_HOOKS builds only)

    # per-kernel timing from the HIP events the library records on the launch stream around every kernel group.
    # (a) over the timed region (batches overlap each other there, so these intervals include time-sharing);
    nback = min(args.steps, 64)
    tim = [plan.timing(i) for i in range(nback)]
    overl = [sum(t[k] for t in tim) / nback for k in range(5)]
    # (b) roofline calibration: the same batch call launched ALONE (one stream, synchronised), so the expansion kernel's
    #     duration is its own: `--calib` launches, timed by the same library events.
    iso = []
    for _ in range(args.calib):
        torch.cuda.synchronize()
        plan.run(my_proofs.data_ptr(), B, advices[0].data_ptr(), wss[0].data_ptr(), streams[0].cuda_stream)
        torch.cuda.synchronize()
        iso.append(plan.timing(0))
    iso = iso[1:] if len(iso) > 1 else iso
    isol = [sum(t[k] for t in iso) / len(iso) for k in range(5)] if iso else overl
    exp_ms = isol[3]
    # (c) the same kernel with its launches back to back: every stream re-expands the records of its last batch
    #     (h2w_fri_expand_records: expansion kernel only), three rounds over all streams; aggregate bytes / wall time.
    b2b_gbs = None
    if args.calib > 0 and not args.cu_split:
        torch.cuda.synchronize()
        tb = time.perf_counter()
        rounds = 3
        for _ in range(rounds):
            for i in range(S):
                plan.expand_records(B, advices[i].data_ptr(), wss[i].data_ptr(), streams[i].cuda_stream)
        torch.cuda.synchronize()
        b2b_gbs = rounds * S * B * plan.num_record_cells * 32 / (time.perf_counter() - tb) / 1e9

    # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes (profiles/), when they cover this workload
    traffic = None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_cfg3_bn254_b32.json")))
        if pj.get("workload") == f"{args.config}/{args.hash}/B{B}/L{args.lookup_bits}":
            traffic = pj["kernels"]["expand_kernel"]["hbm_bytes"]
    except Exception:
        traffic = None

    if rank == 0:
        total_cells = plan.num_cells * total_proofs * args.steps
        value = total_cells / elapsed
        rec_bytes = plan.num_records * 40      # record (32 B) + meta (8 B) read per block
        exp_cells = B * plan.num_record_cells   # cells the expansion kernel writes (the rest: direct cells of the value kernels)
        achieved = (exp_cells * 32) / (exp_ms * 1e-3) / 1e9
        out = {
            "metric": "FRI-verifier witness cells/sec", "value": value, "unit": "cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{args.config}: {desc}, {'PoseidonBN254' if hash_mode else 'Goldilocks-Poseidon'} Merkle, lookup_bits={args.lookup_bits}",
                       "proofs_per_gpu_per_step": B, "batches_in_flight": S, "cells_per_proof": plan.num_cells, "includes_witness_load_cells": True,
                       "proofs": "valid FRI instances of random polynomials, generated on the GPU by the ingest rank (h2w_prove_fri_batch)" if args.proofs == "valid" and args.backend != "gloo" else "uniform random words of the proof's shape",
                       "parallelism": f"proof-sharded x{world}, no data-path collective"},
            "proofs_per_s": total_proofs * args.steps / elapsed,
            "input_generation": ({"proofs": total_proofs, "seconds": round(gen_seconds, 3), "proofs_per_s": round(total_proofs / gen_seconds, 1), "where": "GPU of rank 0, outside the timed region"} if args.proofs == "valid" and args.backend != "gloo" and gen_seconds else None),
            "advice_GBps": value * 32 / 1e9,
            "kernel_ms_isolated": {"prologue": isol[0], "strands": isol[1], "bn254_units": isol[2], "expand": isol[3], "batch": isol[4]},
            "kernel_ms_timed_region": {"prologue": overl[0], "strands": overl[1], "bn254_units": overl[2], "expand": overl[3], "batch": overl[4]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes": exp_cells * 32, "kernel": "expand_kernel",
                         "whole_job_frac": value * 32 / 1e9 / HBM_PEAK_GBS,
                         "achieved_back_to_back": b2b_gbs, "frac_back_to_back": (b2b_gbs / HBM_PEAK_GBS if b2b_gbs else None),
                         "note": f"expand_kernel launched alone ({len(iso)} calibration launches after the timed region, library HIP events on the launch stream): 32 B x {exp_cells} cells per launch ({plan.num_record_cells} of {plan.num_cells} cells/proof come from block records; the others are PoseidonBN254 permutation units / direct cells); record+meta reads {B * rec_bytes / 1e6:.1f} MB extra. whole_job_frac = value x 32 B / peak (all kernels, overlapped batches); achieved_back_to_back = the same kernel alone with its launches back to back on the bench's streams (h2w_fri_expand_records, 3 rounds), aggregate bytes / wall time"},
        }
        if not args.no_cpu_baseline and world == 1:      # the CPU legs run at N = 1 only (rank 0's host)
            out["cpu_baseline"] = cpu_baseline((d, q, rb), hash_mode, args.lookup_bits)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores((d, q, rb), hash_mode, args.lookup_bits, plan.num_cells)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
