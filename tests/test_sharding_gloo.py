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
