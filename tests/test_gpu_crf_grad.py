"""GPU parity of the CRF NLL gradient (phonlp_crf_nll_grad) through the C-ABI: against the
gradients the reference's own CRFLoss produced under autograd (tests/golden/crf.npz) and against autograd through the
CPU oracle on larger batches.  Tolerance 1e-4 relative (+1e-6 absolute): probabilities formed in fp32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import heads_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from phonlp_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def test_crf_nll_grad_reference_fixture(dev, golden):
    from phonlp_amd import ops
    z = golden("crf.npz")
    for c in range(int(z["n"])):
        x, m, t, tr = (torch.from_numpy(z[f"c{c}_{k}"]) for k in ("inputs", "masks", "tags", "trans"))
        dx, dtr = ops.crf_nll_grad(x.to(dev), tr.to(dev), m.to(dev, torch.uint8), None, t.to(dev, torch.int32),
                                   1.0 / x.shape[0])
        torch.testing.assert_close(dx.cpu(), torch.from_numpy(z[f"c{c}_d_inputs"]), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(dtr.cpu(), torch.from_numpy(z[f"c{c}_d_trans"]), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("B,L,T,ragged,relu", [(256, 127, 9, False, True), (64, 127, 12, True, True),
                                               (7, 300, 40, True, False), (3, 33, 64, True, False), (5, 1, 9, False, True)])
def test_crf_nll_grad_vs_oracle_autograd(dev, B, L, T, ragged, relu):
    from phonlp_amd import ops
    g = torch.Generator().manual_seed(B * 7 + L)
    x = torch.randn(B, L, T, generator=g)
    if relu:
        x = torch.relu(x)  # the reference's emissions are ReLU'd (jointmodel/model.py:158)
    tr = torch.randn(T, T, generator=g) * 0.3
    lens = sorted([int(v) for v in torch.randint(1, L + 1, (B,), generator=g)], reverse=True) if ragged else [L] * B
    masks = torch.arange(L).view(1, L) >= torch.tensor(lens).view(B, 1)
    x = x.masked_fill(masks.unsqueeze(2), 0.0)
    tags = torch.randint(0, T, (B, L), generator=g)
    xo, to = x.clone().requires_grad_(True), tr.clone().requires_grad_(True)
    O.crf_nll(xo, masks, tags, to).backward()
    lens_t = torch.tensor(lens, dtype=torch.int32, device=dev)
    for pm, sl in ((masks.to(dev, torch.uint8), None), (None, lens_t)):  # padding given either way
        dx, dtr = ops.crf_nll_grad(x.to(dev), tr.to(dev), pm, sl, tags.to(dev, torch.int32), 1.0 / B)
        torch.testing.assert_close(dx.cpu(), xo.grad, rtol=1e-4, atol=2e-6)
        torch.testing.assert_close(dtr.cpu(), to.grad, rtol=1e-4, atol=2e-5)
    # the shares are added in a fixed order: same bits on a second run; every probability row sums to one
    dx2, dtr2 = ops.crf_nll_grad(x.to(dev), tr.to(dev), None, lens_t, tags.to(dev, torch.int32), 1.0 / B)
    assert torch.equal(dx2, dx) and torch.equal(dtr2, dtr)
    rows = dx.cpu().sum(2) * B  # (sum_c P) - 1 = 0 on live steps, 0 on padding; alphas reach a few hundred, so an
    assert float(rows.abs().max()) <= 1e-3  # fp32 ulp of alpha + beta - logZ is ~3e-5 and exp() carries it over


def test_compat_crfloss_backward(dev, golden):
    """compat.CRFLoss is differentiable: loss.backward() fills inputs.grad and _transitions.grad like the reference's."""
    from phonlp_amd.compat import CRFLoss
    z = golden("crf.npz")
    for c in range(int(z["n"])):
        x, m, t, tr = (torch.from_numpy(z[f"c{c}_{k}"]) for k in ("inputs", "masks", "tags", "trans"))
        crit = CRFLoss(tr.shape[0])
        crit._transitions.data.copy_(tr)
        xin = x.clone().requires_grad_(True)
        loss, _ = crit(xin, m, t)
        assert abs(float(loss.detach()) - float(z[f"c{c}_loss"])) <= 1e-4 * max(1.0, abs(float(z[f"c{c}_loss"])))
        (2.0 * loss).backward()
        torch.testing.assert_close(xin.grad, 2.0 * torch.from_numpy(z[f"c{c}_d_inputs"]), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(crit._transitions.grad, 2.0 * torch.from_numpy(z[f"c{c}_d_trans"]), rtol=1e-4, atol=1e-6)
