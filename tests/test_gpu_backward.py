"""f4, backward THROUGH the layers: phonlp_linear_backward and phonlp_bilinear_backward against the gradients the
reference's own autograd produced behind DeepBiaffineScorer (tests/golden/biaffine_backward.npz, O = 1) and against torch
autograd of the same formulas on the CPU; all three matrix engines."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(params=[0, 1, 2], ids=["f32mfma", "bf16x6", "f16x3"])
def engine(request):
    from phonlp_amd import ops
    old = ops.get_gemm_engine()
    ops.set_gemm_engine(request.param)
    yield request.param
    ops.set_gemm_engine(old)


def _close(got, want, what, rtol=2e-4):
    want = torch.as_tensor(want)
    scale = float(want.abs().max()) + 1e-12
    err = float((got.cpu() - want).abs().max())
    assert err <= rtol * scale, f"{what}: max error {err:.3e} against scale {scale:.3e}"


@pytest.mark.parametrize("M,N,K,relu", [(37, 5, 10, True), (37, 5, 10, False), (1030, 24, 400, True), (2049, 401, 76, False)])
def test_linear_backward_vs_autograd(dev, engine, M, N, K, relu):
    from phonlp_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).requires_grad_(True)
    w = (torch.randn(N, K, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(N, generator=g).requires_grad_(True)
    dy = torch.randn(M, N, generator=g)
    y = x.double() @ w.double().t() + b.double()
    if relu:
        y = torch.relu(y)
    (y * dy.double()).sum().backward()
    yact = y.detach().float().to(dev)
    dX, dW, db = ops.linear_backward(x.detach().to(dev), w.detach().to(dev), yact if relu else None, dy.to(dev), relu)
    _close(dX, x.grad, "dX")
    _close(dW, w.grad, "dW")
    _close(db, b.grad, "db")
    only_dx, none_w, none_b = ops.linear_backward(x.detach().to(dev), w.detach().to(dev), yact if relu else None, dy.to(dev),
                                                  relu, need_dW=False, need_db=False)
    assert none_w is None and none_b is None and torch.equal(only_dx, dX)


def test_scorer_backward_vs_reference_autograd(dev, engine):
    """DeepBiaffineScorer (O = 1) end to end: first layers (W1 | W2 packed, ReLU, ones column) -> bilinear form; the
    gradients of the input and of every parameter equal what the reference's autograd recorded."""
    from phonlp_amd import ops
    z = np.load(os.path.join(GOLD, "biaffine_backward.npz"))
    for c in range(int(z["n_cases"])):
        t = lambda k: torch.from_numpy(z[f"c{c}_{k}"]).to(dev)
        h, G = t("h"), t("G")
        B, L, D = h.shape
        W1, b1, W2, b2, Wbil = t("p_W1.weight"), t("p_W1.bias"), t("p_W2.weight"), t("p_W2.bias"), t("p_scorer.W_bilin.weight")
        H = W1.shape[0]
        KP = ops.kp(H)
        W12, b12, Wbt = ops.pack_scorer(W1, b1, W2, b2, Wbil.contiguous())
        hm = h.reshape(B * L, D).contiguous()
        XF = ops.linear(hm, W12, b12, True, None, 0)
        # the forward these gradients belong to is the reference's
        S = torch.einsum("bid,ed,bje->bij", XF[:, :KP].view(B, L, KP), Wbt, XF[:, KP:].view(B, L, KP))
        _close(S, z[f"c{c}_out"], f"case {c}: scores")
        dXF, dWbt = ops.bilinear_backward(XF, Wbt, G.contiguous(), H)
        dh, dW12, db12 = ops.linear_backward(hm, W12, XF, dXF, True)
        _close(dh.view(B, L, D), z[f"c{c}_d_h"], f"case {c}: d hidden")
        _close(dW12[:H], z[f"c{c}_d_W1.weight"], f"case {c}: d W1.weight")
        _close(dW12[KP:KP + H], z[f"c{c}_d_W2.weight"], f"case {c}: d W2.weight")
        _close(db12[:H], z[f"c{c}_d_W1.bias"], f"case {c}: d W1.bias")
        _close(db12[KP:KP + H], z[f"c{c}_d_W2.bias"], f"case {c}: d W2.bias")
        # dweight[d1][d2][0] of the (H+1) x (H+1) x 1 parameter = dWbt[d2][d1]
        _close(dWbt[:H + 1, :H + 1].t(), z[f"c{c}_d_scorer.W_bilin.weight"][:, :, 0], f"case {c}: d W_bilin.weight")


def test_bilinear_backward_headline_shape(dev):
    """B=64, L=128, H=400 against torch on the device (fp32 matmuls): the sizes the training loop runs."""
    from phonlp_amd import ops
    B, L, H = 64, 128, 400
    KP = ops.kp(H)
    g = torch.Generator().manual_seed(9)
    XF = torch.relu(torch.randn(B * L, 2 * KP, generator=g)).to(dev)
    Wbt = (torch.randn(KP, KP, generator=g) * 0.05).to(dev)
    G = (torch.randn(B, L, L, generator=g) / (B * L)).to(dev)
    dXF, dWbt = ops.bilinear_backward(XF, Wbt, G, H)
    X1, X2 = XF[:, :KP].view(B, L, KP).double(), XF[:, KP:].view(B, L, KP).double()
    Gd, Wd = G.double(), Wbt.double()
    U = Gd @ X2
    T = X1 @ Wd.t()
    _close(dXF[:, :KP].view(B, L, KP), (U @ Wd).float().cpu(), "dX1")
    _close(dXF[:, KP:].view(B, L, KP), (Gd.transpose(1, 2) @ T).float().cpu(), "dX2")
    _close(dWbt, (U.reshape(B * L, KP).t() @ X1.reshape(B * L, KP)).float().cpu(), "dWbt")


def test_label_scorer_backward_vs_reference_autograd(dev, engine):
    """DeepBiaffineScorer with O relations, loss read at the gold heads (jointmodel/model.py:218-223): gradients of the
    input and of every parameter equal the reference's autograd."""
    from phonlp_amd import ops
    z = np.load(os.path.join(GOLD, "biaffine_backward.npz"))
    for c in range(int(z["n_label_cases"])):
        t = lambda k: torch.from_numpy(z[f"l{c}_{k}"]).to(dev)
        h, gl, head, lens = t("h"), t("g"), t("head"), t("lens")
        B, L, D = h.shape
        W1, b1, W2, b2, Wbil = t("p_W1.weight"), t("p_W1.bias"), t("p_W2.weight"), t("p_W2.bias"), t("p_scorer.W_bilin.weight")
        H, O = W1.shape[0], Wbil.shape[2]
        KP = ops.kp(H)
        W12, b12, Wbt = ops.pack_scorer(W1, b1, W2, b2, Wbil.contiguous())
        hm = h.reshape(B * L, D).contiguous()
        XF = ops.linear(hm, W12, b12, True, None, 0)
        # forward check: the scores at the gold heads are the reference's
        X1, X2 = XF[:, :KP].view(B, L, KP), XF[:, KP:].view(B, L, KP)
        X2h = torch.gather(X2, 1, head.long().view(B, L, 1).expand(-1, -1, KP))
        s = torch.einsum("bid,oed,bie->bio", X1, Wbt.view(O, KP, KP), X2h)
        _close(s, z[f"l{c}_at_head"], f"label case {c}: scores at the heads")
        dXF, dWbt = ops.label_backward(XF, Wbt, gl.contiguous(), head.contiguous(), lens.contiguous(), H, O)
        dh, dW12, db12 = ops.linear_backward(hm, W12, XF, dXF, True)
        _close(dh.view(B, L, D), z[f"l{c}_d_h"], f"label case {c}: d hidden")
        _close(dW12[:H], z[f"l{c}_d_W1.weight"], f"label case {c}: d W1.weight")
        _close(dW12[KP:KP + H], z[f"l{c}_d_W2.weight"], f"label case {c}: d W2.weight")
        _close(db12[:H], z[f"l{c}_d_W1.bias"], f"label case {c}: d W1.bias")
        _close(db12[KP:KP + H], z[f"l{c}_d_W2.bias"], f"label case {c}: d W2.bias")
        # dweight[d1][d2][o] of the (H+1) x (H+1) x O parameter -- stored flat as (d1, o, d2), the reference's reshape quirk
        # (biaffine.py:27-35) -- against dWbt[(o, d2)][d1]: pack the reference's gradient the way the weight is packed
        want = torch.from_numpy(z[f"l{c}_d_scorer.W_bilin.weight"]).to(dev).contiguous()
        _, _, want_packed = ops.pack_scorer(W1, b1, W2, b2, want)
        _close(dWbt, want_packed.cpu(), f"label case {c}: d W_bilin.weight")
