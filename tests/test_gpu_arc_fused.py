"""Engine 2: the arc scorers' bilinear stages fused into one kernel on the image their first layer writes
(csrc/arc_f16x3.hip, ref biaffine.py:21-37 + annotate_model.py:192-214) against (a) the CPU oracle and (b) the
unfused engine-2 form (stage A as a GEMM, stage B on fp32 features), padded and packed, at head sizes that instantiate
every tile count of the kernel."""
import numpy as np
import pytest
import torch

from oracle import heads_oracle as O

pytestmark = pytest.mark.gpu


def _finite_close(a, b, rtol, atol):
    a, b = a.cpu(), b.cpu()
    fin = torch.isfinite(b)
    assert torch.equal(fin, torch.isfinite(a))
    torch.testing.assert_close(a[fin], b[fin], rtol=rtol, atol=atol)


@pytest.mark.parametrize("E,TE,H,U,O_,T", [(768, 100, 400, 24, 33, 9), (64, 12, 48, 7, 6, 5), (96, 20, 100, 9, 5, 4),
                                           (128, 16, 200, 11, 4, 6)])
@pytest.mark.parametrize("B,L,ragged", [(16, 128, False), (9, 128, True), (7, 100, True), (5, 40, False), (3, 17, True),
                                        (2, 2, False), (4, 200, True), (3, 300, False), (2, 512, True), (3, 129, False)])
def test_fused_arc_vs_oracle_and_unfused(E, TE, H, U, O_, T, B, L, ragged):
    from phonlp_amd import HeadWeights, TaskHeads, ops
    from phonlp_amd import synthetic as S
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    ops.set_gemm_engine(2)
    sd = S.make_state_dict(E, TE, H, U, O_, T, seed=H + L)
    enc_dep, enc_tag = S.make_encoder_outputs(B, L, E, seed=B * 5 + L)
    lens1 = S.make_sentlens(B, L, ragged)
    lens = [max(l - 1, 1) for l in lens1]
    heads = TaskHeads(HeadWeights.from_state_dict(sd, dev))
    assert heads.w.arc_KC in (64, 128, 256, 416)
    od = O.dep_heads(enc_dep, lens1, sd, keep_label_scores=False)
    got = {}
    for fused in (True, False):
        heads.cfg["arc_fused"] = fused
        for pack_pct in ((90, 0) if ragged else (0,)):
            heads.cfg["pack_pct"] = pack_pct
            out = heads.annotate_batch(enc_dep.to(dev), lens1, enc_tag[:, :max(L - 1, 1)].to(dev), lens)
            torch.cuda.synchronize()
            assert (out["mst_status"] == 0).all()
            arc = out["arc_logp"].cpu()
            # the packed form leaves the rows of padded dependents unwritten: compare the live rows
            for b in range(B):
                _finite_close(arc[b, :lens1[b]], od["arc_logp"][b, :lens1[b]], rtol=2e-4, atol=2e-5)
            got[(fused, pack_pct)] = (arc, out["heads"].cpu())
    base_arc, base_heads = got[(False, 0)]
    for key, (arc, hd) in got.items():
        for b in range(B):
            _finite_close(arc[b, :lens1[b]], base_arc[b, :lens1[b]], rtol=1e-4, atol=1e-5)
        # decoded heads: equal wherever the scores leave a margin (they do on these inputs)
        assert torch.equal(hd, base_heads), key
