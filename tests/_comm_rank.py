"""One rank of tests/test_gpu_batch.py::test_two_ranks_over_rccl (a fresh process per rank, started before anything touches a GPU):
h2w_comm_* over RCCL with world 2 - rank 0 broadcasts the proof block, every rank generates its (proof, query) shard into its packed buffer,
the ranks all-gather the digests of their buffers.  Prints one JSON line: the gathered digests and this rank's status words.
usage: _comm_rank.py RANK WORLD ID_FILE MODE N_PROOFS"""
import importlib, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    rank, world, id_file, mode, n = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    import torch
    torch.cuda.set_device(rank)
    h2w = importlib.import_module("halo2-plonky2-verifier_amd"); api = importlib.import_module("halo2-plonky2-verifier_amd.api")
    D = importlib.import_module("halo2-plonky2-verifier_amd.distributed")
    import pyoracle as O
    L = h2w.lib()
    if rank == 0:
        ident = D.Comm.unique_id(L)
        with open(id_file + ".tmp", "wb") as f:
            f.write(ident)
        os.replace(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            assert time.time() - t0 < 120, "rank 0 never published the communicator id"
            time.sleep(0.05)
        ident = open(id_file, "rb").read()
    comm = D.Comm(L, ident, rank, world, rank)
    sh = h2w.fibonacci_shape(7, 5, rate_bits=2, hash_mode=mode); osh = O.fibonacci_shape(7, 5, rate_bits=2, hash_mode=mode)
    plan = api.Plan(sh, h2w.published_consts(), rank)
    words = plan.proof_words
    proofs = torch.zeros(n * words, dtype=torch.int64, device=f"cuda:{rank}")
    if rank == 0:      # the ingest rank holds the proofs; the others learn them from the one broadcast
        host = torch.empty(n * words, dtype=torch.int64)
        for i in range(n):
            host[i * words:(i + 1) * words] = torch.frombuffer(bytearray(bytes(O.synth_proof(osh, 900 + i))), dtype=torch.int64)
        proofs.copy_(host)
    st = torch.cuda.current_stream().cuda_stream
    comm.broadcast_proofs(proofs, 0, st)
    cells = plan.shard_cells(n, rank, world)
    buf = torch.zeros(cells * 4, dtype=torch.int64, device=f"cuda:{rank}")
    ws = torch.zeros(plan.workspace_bytes(n), dtype=torch.uint8, device=f"cuda:{rank}")
    plan.run_shard_compact(proofs.data_ptr(), n, buf.data_ptr(), ws.data_ptr(), rank, world, st)
    dig = torch.zeros(4, dtype=torch.int64, device=f"cuda:{rank}"); allg = torch.zeros(4 * world, dtype=torch.int64, device=f"cuda:{rank}")
    plan.advice_digest(buf.data_ptr(), cells, dig.data_ptr(), st)
    comm.allgather_digests(dig, allg, st)
    torch.cuda.synchronize()
    print(json.dumps({"rank": rank, "status": plan.status(ws.data_ptr(), n, st), "digests": [int(x) & 0xFFFFFFFFFFFFFFFF for x in allg.cpu().tolist()],
                      "proof_checksum": int(proofs.sum().item()) & 0xFFFFFFFFFFFFFFFF}))
    comm.close(); plan.close()


if __name__ == "__main__":
    main()
