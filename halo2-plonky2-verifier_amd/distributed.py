"""Multi-GPU plumbing for the batched hot path: the path shards by proof - or, for few large proofs, by (proof, query)
unit - with no data-path collective.  One process per GPU; the ONLY collective is one broadcast of the flat proof block from
the ingest rank (torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(total_proofs, world, rank):
    """Contiguous, balanced shard [lo, hi) of `total_proofs` for `rank`."""
    base, rem = divmod(total_proofs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_proofs(proofs, src=0):
    """In-place broadcast of the int64 proof block [total_proofs * proof_words] from `src` (no-op when not distributed)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(proofs, src=src)
    return proofs


def max_over_ranks(value, device):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([value], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return value


def unit_owner(proof, query, num_queries, world):
    """SURVEY §8e partitioning: unit = (proof, query), round-robin over the ranks (h2w_fri_witness_batch_shard)."""
    return (proof * num_queries + query) % world


def my_units(n_proofs, num_queries, rank, world):
    return [(p, q) for p in range(n_proofs) for q in range(num_queries) if unit_owner(p, q, num_queries, world) == rank]


def prologue_owner(proof, world):
    """The rank that EMITS the prologue block of a proof in a query-sharded run (every rank computes its values)."""
    return proof % world


def shard_cells(n_proofs, num_queries, rank, world, prologue_cells, query_cells_first, query_cells_rest):
    """Advice cells rank `rank` writes in a query-sharded run of n_proofs proofs: its query blocks (query 0 of a context holds the one
    cached load_zero cell more than the others) plus the prologue blocks it owns.  The shares of all ranks add up to the stream."""
    cells = sum(query_cells_first if q == 0 else query_cells_rest for _, q in my_units(n_proofs, num_queries, rank, world))
    return cells + prologue_cells * sum(1 for p in range(n_proofs) if prologue_owner(p, world) == rank)


def shard_ranges(n_proofs, num_queries, rank, world, layout):
    """Cell ranges (proof, first cell, length) that rank `rank` writes in a query-sharded run; `layout` = Plan.strand_layout()."""
    pro, q0, qn, total = layout
    assert pro + q0 + (num_queries - 1) * qn == total
    out = []
    for p in range(n_proofs):
        if prologue_owner(p, world) == rank:
            out.append((p, 0, pro))
        for q in range(num_queries):
            if unit_owner(p, q, num_queries, world) == rank:
                out.append((p, pro, q0) if q == 0 else (p, pro + q0 + (q - 1) * qn, qn))
    return out


class Comm:
    """The C-ABI ingest communicator (include/h2w.h h2w_comm_*: RCCL resolved inside libh2w.so), for callers without torch.distributed.
    `id128` comes from Comm.unique_id() on one rank and reaches the others by any out-of-band channel."""

    def __init__(self, lib, id128, rank, world, device_id=0):
        import ctypes as C
        self.L = lib
        self._id = (C.c_ubyte * 128).from_buffer_copy(bytes(id128))
        self.p = lib.h2w_comm_init(self._id, rank, world, device_id)
        if not self.p:
            raise RuntimeError(lib.h2w_last_error().decode())

    @staticmethod
    def unique_id(lib):
        import ctypes as C
        buf = (C.c_ubyte * 128)()
        if lib.h2w_comm_unique_id(buf) != 0:
            raise RuntimeError(lib.h2w_last_error().decode())
        return bytes(buf)

    def broadcast_proofs(self, proofs, root=0, stream=0):
        """In place, int64 device tensor."""
        if self.L.h2w_comm_broadcast_proofs(self.p, proofs.data_ptr(), proofs.numel(), root, stream) != 0:
            raise RuntimeError(self.L.h2w_last_error().decode())
        return proofs

    def allgather_digests(self, digest4, out, stream=0):
        if self.L.h2w_comm_allgather_digests(self.p, digest4.data_ptr(), out.data_ptr(), stream) != 0:
            raise RuntimeError(self.L.h2w_last_error().decode())
        return out

    def close(self):
        if self.p:
            self.L.h2w_comm_free(self.p); self.p = None
