"""Both expansion kernels produce the same bytes: the batched hot path takes expand_fast (record-owning lanes, lookup_bits 21 / 13 / 8),
every other lookup_bits and the eager contexts take the generic expand_kernel_t.  lookup_bits 20 has the same cell layout as 21
(4 limbs), so the same proof goes through both kernels; each stream is compared with the oracle's."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [1, 0])
def test_generic_and_fast_expansion_kernels_agree_with_the_oracle(h2w, h2w_api, oracle, consts, mode):
    import torch
    ko, kh = consts
    for L in (21, 20, 13, 12, 8):
        sh = h2w.fibonacci_shape(7, 2, hash_mode=mode, lookup_bits=L); osh = oracle.fibonacci_shape(7, 2, hash_mode=mode, lookup_bits=L)
        plan = h2w_api.Plan(sh, kh)
        pr = oracle.synth_proof(osh, 900 + L)
        d_proofs = torch.frombuffer(bytearray(bytes(pr)), dtype=torch.int64).cuda()
        advice = torch.zeros(plan.num_cells * 32, dtype=torch.uint8, device="cuda")
        ws = torch.zeros(plan.workspace_bytes(1), dtype=torch.uint8, device="cuda")
        plan.run(d_proofs.data_ptr(), 1, advice.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ctx = oracle.Ctx(L, track_scopes=False)
        assert oracle.verify_stark(ctx, osh, ko, pr) == 0
        assert advice.cpu().numpy().tobytes() == ctx.advice_bytes(), f"lookup_bits {L}"
        ctx.close(); plan.close()
