"""N>1 host logic on CPU: world_size-2 gloo processes broadcast the proof block and take disjoint proof shards whose
union is the whole batch (bench.py --gpus N does the same over RCCL).  No GPU compute here."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, words, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import importlib
    D = importlib.import_module("halo2-plonky2-verifier_amd.distributed")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    proofs = torch.zeros(total * words, dtype=torch.int64)
    if rank == 0:
        import pyoracle as O
        sh = O.fibonacci_shape(6, 2)
        assert O.lib().orc_proof_words(sh) == words
        for i in range(total):
            proofs[i * words:(i + 1) * words] = torch.frombuffer(bytearray(bytes(O.synth_proof(sh, 100 + i))), dtype=torch.int64)
    D.broadcast_proofs(proofs, src=0)
    lo, hi = D.shard_range(total, world, rank)
    digest = int(proofs.sum().item())
    mx = D.max_over_ranks(float(rank + 1), torch.device("cpu"))
    out.put((rank, lo, hi, digest, int(proofs[lo * words].item()), mx))
    dist.barrier(); dist.destroy_process_group()


def test_broadcast_and_disjoint_shards():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as O
    words = O.lib().orc_proof_words(O.fibonacci_shape(6, 2))
    world, total = 2, 5
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, words, out)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(out.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert res[0][3] == res[1][3]                      # both ranks hold the same proof block after the broadcast
    assert [(r[1], r[2]) for r in res] == [(0, 3), (3, 5)]   # disjoint, covering, balanced
    assert res[1][4] != 0                              # rank 1 sees rank 0's data
    assert all(r[5] == 2.0 for r in res)               # max-over-ranks reduction


def test_shard_range_properties():
    sys.path.insert(0, ROOT)
    import importlib
    D = importlib.import_module("halo2-plonky2-verifier_amd.distributed")
    for total in (0, 1, 7, 256):
        for world in (1, 2, 4, 8):
            rs = [D.shard_range(total, world, r) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1


def test_query_unit_partition_is_exact():
    """SURVEY §8e partitioning used by h2w_fri_witness_batch_shard: (proof, query) units round-robin over the ranks — every unit
    has exactly one owner, shares differ by at most one unit, for any world size (the device kernels use the same formula)."""
    import importlib
    sys.path.insert(0, ROOT)
    D = importlib.import_module("halo2-plonky2-verifier_amd.distributed")
    for n_proofs, nq, world in [(1, 84, 8), (256, 28, 8), (3, 5, 4), (7, 1, 2), (2, 28, 3), (5, 4, 1)]:
        shares = [D.my_units(n_proofs, nq, r, world) for r in range(world)]
        flat = [u for s_ in shares for u in s_]
        assert sorted(flat) == [(p, q) for p in range(n_proofs) for q in range(nq)]
        assert max(map(len, shares)) - min(map(len, shares)) <= 1


def test_packed_shard_layout_arithmetic():
    """h2w_plan_shard_cells / h2w_plan_shard_block (host arithmetic, no device): a rank's packed buffer holds exactly the blocks the round-robin deal
    gives it (distributed.shard_ranges), back to back without overlap; over the ranks every cell of every proof has one owner.  The device side of
    the same arithmetic (batchargs.h block_out, expand.hip fast_tile_range) is compared with the oracle in tests/test_gpu_batch.py::test_packed_shard_layout."""
    import importlib
    sys.path.insert(0, ROOT)
    h2w = importlib.import_module("halo2-plonky2-verifier_amd"); api = importlib.import_module("halo2-plonky2-verifier_amd.api")
    D = importlib.import_module("halo2-plonky2-verifier_amd.distributed")
    for mode in (1, 0):
        sh = h2w.fibonacci_shape(6, 5, hash_mode=mode)
        plan = api.Plan(sh, h2w.published_consts())
        layout = plan.strand_layout()
        for n, world in [(3, 2), (5, 3), (2, 8), (7, 1), (9, 4)]:
            seen = {}
            for rank in range(world):
                cells = plan.shard_cells(n, rank, world)
                blocks = []
                for p in range(n):
                    for q in range(-1, sh.num_queries):
                        b = plan.shard_block(rank, world, p, q)
                        if b is not None:
                            blocks.append((b[0], b[1], p, b[2]))
                blocks.sort()
                assert all(blocks[i][0] + blocks[i][1] <= blocks[i + 1][0] for i in range(len(blocks) - 1))      # no overlap, in (proof, block) order
                assert blocks[-1][0] + blocks[-1][1] <= cells and blocks[0][0] == 0
                assert cells - sum(b[1] for b in blocks) <= n * sh.num_queries                                    # slack: at most one cell per query slot
                assert sorted((p, g, c) for _, c, p, g in blocks) == sorted(D.shard_ranges(n, sh.num_queries, rank, world, layout))
                for _, c, p, g in blocks:
                    for key in ((p, g), (p, g + c - 1)):
                        assert key not in seen; seen[key] = rank
            assert sum(plan.shard_cells(n, r, world) for r in range(world)) >= n * plan.num_cells
        plan.close()


def _shard_worker(rank, world, port, n, out):
    """One rank of a query-sharded run, on the CPU: the broadcast, the host-side partition (Plan.strand_layout + shard_ranges:
    what h2w_fri_witness_batch_shard's kernels follow on the device) and this rank's cells - cut out of the oracle's stream, which
    stands in for the device here - gathered over gloo."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import importlib
    import pyoracle as O
    h2w = importlib.import_module("halo2-plonky2-verifier_amd")
    api = importlib.import_module("halo2-plonky2-verifier_amd.api")
    D = importlib.import_module("halo2-plonky2-verifier_amd.distributed")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    osh = O.fibonacci_shape(6, 3, hash_mode=1); sh = h2w.fibonacci_shape(6, 3, hash_mode=1)
    plan = api.Plan(sh, h2w.published_consts())           # layout queries only: no device is touched (and none exists in this test)
    words = plan.proof_words
    proofs = torch.zeros(n * words, dtype=torch.int64)
    if rank == 0:
        for i in range(n):
            proofs[i * words:(i + 1) * words] = torch.frombuffer(bytearray(bytes(O.synth_proof(osh, 700 + i))), dtype=torch.int64)
    D.broadcast_proofs(proofs, src=0)
    ko = O.published_consts()
    mine = torch.zeros(n * plan.num_cells * 4, dtype=torch.int64).view(n, plan.num_cells, 4)
    wrote = torch.zeros(n, plan.num_cells, dtype=torch.int32)
    ranges = D.shard_ranges(n, sh.num_queries, rank, world, plan.strand_layout())
    for p in sorted({r[0] for r in ranges}):
        ctx = O.Ctx(21)
        pw = (O.C.c_uint64 * words).from_buffer(bytearray(proofs[p * words:(p + 1) * words].numpy().tobytes()))
        assert O.verify_stark(ctx, osh, ko, pw) == 0
        full = torch.frombuffer(bytearray(ctx.advice_bytes()), dtype=torch.int64).view(plan.num_cells, 4); ctx.close()
        for pp, first, length in ranges:
            if pp == p:
                mine[p, first:first + length] = full[first:first + length]; wrote[p, first:first + length] += 1
    cells = int(wrote.sum().item())
    assert cells == sum(r[2] for r in ranges) == D.shard_cells(n, sh.num_queries, rank, world, *plan.strand_layout()[:3])
    dist.all_reduce(wrote, op=dist.ReduceOp.SUM)          # how many ranks wrote each cell
    dist.all_reduce(mine, op=dist.ReduceOp.SUM)           # disjoint writers -> the sum IS the union
    if rank == 0:
        ok = bool((wrote == 1).all().item())
        want = []
        for p in range(n):
            ctx = O.Ctx(21)
            pw = (O.C.c_uint64 * words).from_buffer(bytearray(proofs[p * words:(p + 1) * words].numpy().tobytes()))
            assert O.verify_stark(ctx, osh, ko, pw) == 0
            want.append(torch.frombuffer(bytearray(ctx.advice_bytes()), dtype=torch.int64).view(plan.num_cells, 4)); ctx.close()
        out.put((ok, bool((torch.stack(want) == mine).all().item()), cells))
    else:
        out.put((True, True, cells))
    plan.close()
    dist.barrier(); dist.destroy_process_group()


def test_query_sharded_witness_union_over_two_ranks():
    """world_size 2 over gloo: every cell of every proof is produced by exactly one rank and the union is the oracle's stream."""
    world, n = 2, 3
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + ((os.getpid() + 7) % 500)
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, n, out)) for r in range(world)]
    [p.start() for p in procs]
    res = [out.get(timeout=300) for _ in range(world)]
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert all(r[0] and r[1] for r in res)
    shares = sorted(r[2] for r in res)
    assert shares[0] > 0 and shares[1] < 2 * shares[0]      # both ranks carry a real share
