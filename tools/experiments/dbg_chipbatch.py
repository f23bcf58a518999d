import importlib, sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
h2w = importlib.import_module("halo2-plonky2-verifier_amd")
L = h2w.lib()
print("lib loaded", flush=True)
P = 0xFFFFFFFF00000001
for lb in (21, 13):
  for op in range(9):
    h = L.h2w_chipbatch_new(op, lb, 0)
    nw, nc = int(L.h2w_chipbatch_num_operands(h)), int(L.h2w_chipbatch_num_cells(h))
    print("op", op, "L", lb, "nw", nw, "nc", nc, flush=True)
    n = 200
    rnd = random.Random(op)
    ops = [[rnd.randrange(1, P) for _ in range(nw)] for _ in range(n)]
    d_ops = torch.tensor(np.array(ops, dtype=np.uint64).view(np.int64).reshape(-1), dtype=torch.int64, device="cuda")
    advice = torch.zeros(n * nc * 32, dtype=torch.uint8, device="cuda")
    status = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    rc = L.h2w_chipbatch_run(h, d_ops.data_ptr(), n, advice.data_ptr(), status.data_ptr(), 0)
    print(" run rc", rc, h2w.last_error() if rc else "", flush=True)
    torch.cuda.synchronize()
    print(" synced", status[:4].tolist(), flush=True)
    L.h2w_chipbatch_free(h)
