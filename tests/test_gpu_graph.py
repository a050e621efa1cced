"""TaskHeads.capture: the step as a HIP graph returns what the eager step returns, on every replay's own inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(B, L, seed, dev):
    from phonlp_amd import synthetic as S
    g = torch.Generator(device="cpu").manual_seed(seed)
    enc_dep, enc_tag = S.make_encoder_outputs(B, L, seed=seed)
    ld = torch.randint(max(L // 4, 2), L + 1, (B,), generator=g).sort(descending=True).values.to(torch.int32)
    return enc_dep.to(dev), ld.to(dev), enc_tag.to(dev), (ld - 1).to(dev)


@pytest.mark.parametrize("B,L", [(1, 24), (8, 64), (32, 128), (4, 160)])  # 160: the unfused label path, MST in global memory
def test_captured_step_matches_eager(B, L):
    from phonlp_amd import HeadWeights, TaskHeads, synthetic as S
    dev = torch.device("cuda:0")
    heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(), dev))
    step = heads.capture(B, L)
    for seed in (11, 12, 13):  # three different batches through the same graph
        ed, ld, et, lt = _batch(B, L, seed, dev)
        want = heads.annotate_batch(ed, ld, et, lt)
        torch.cuda.synchronize()
        want = {k: want[k].clone() for k in ("pos", "ner", "heads", "rels", "mst_status")}
        step.enc_dep.copy_(ed); step.lens_dep.copy_(ld); step.enc_tag.copy_(et); step.lens_tag.copy_(lt)
        got = step.replay()
        torch.cuda.synchronize()
        for k, v in want.items():
            assert torch.equal(v, got[k]), (seed, k)
        assert int(got["mst_status"].abs().max()) == 0
