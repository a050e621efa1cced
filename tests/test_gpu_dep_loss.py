"""GPU parity of the training-side parser loss (phonlp_dep_loss) through the C-ABI:
against the loss the reference's own dep_forward(eval=False) returned (tests/golden/dep_loss_small.npz, made by
tests/golden/make_golden.py) and against the CPU oracle on synthetic batches at the real head sizes.
Tolerance: 1e-4 relative on every sum (north star: 1e-3 on scores); the terms are sums of fp32 log-probabilities."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import heads_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from phonlp_amd import _lib
    _lib.lib()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


@pytest.fixture(params=[0, 1, 2], ids=["f32mfma", "bf16x6", "f16x3"])
def engine(request, dev):
    from phonlp_amd import ops
    before = ops.get_gemm_engine()
    ops.set_gemm_engine(request.param)
    yield request.param
    ops.set_gemm_engine(before)


def _params(z, prefix="p_"):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def _rel(a, b):
    return abs(float(a) - float(b)) / max(1.0, abs(float(b)))


def test_dep_loss_reference_fixture(dev, engine, golden):
    from phonlp_amd import HeadWeights, TaskHeads
    z = golden("dep_loss_small.npz")
    heads = TaskHeads(HeadWeights.from_state_dict(_params(z), dev))
    for c in range(int(z["n_cases"])):
        lens = z[f"c{c}_lens"].tolist()
        loss = heads.parser_loss(torch.from_numpy(z[f"c{c}_enc"]).to(dev), lens, torch.from_numpy(z[f"c{c}_head"]),
                                 torch.from_numpy(z[f"c{c}_rel"]), int(z[f"c{c}_nwords"])).cpu().numpy()
        for k in range(3):  # arc, label, linearization sums as the reference's crit_dep returned them
            assert _rel(loss[k], z[f"c{c}_ce"][k]) <= 1e-4, (c, k, loss, z[f"c{c}_ce"])
        assert _rel(loss[4], z[f"c{c}_loss"]) <= 1e-4, (c, loss, z[f"c{c}_loss"])
        want_dist = float(z[f"c{c}_loss"]) * int(z[f"c{c}_nwords"]) - float(z[f"c{c}_ce"].astype(np.float64).sum())
        assert abs(loss[3] - want_dist) <= 1e-3 * max(1.0, float(z[f"c{c}_ce"].sum()))  # a difference of large sums


def _gold_trees(lens1, L, O_, seed):
    g = torch.Generator().manual_seed(seed)
    B = len(lens1)
    head = torch.zeros(B, L, dtype=torch.int64)
    rel = torch.zeros(B, L, dtype=torch.int64)
    for b, ln in enumerate(lens1):
        if ln > 1:
            h = torch.randint(0, ln - 1, (ln - 1,), generator=g)
            i = torch.arange(1, ln)
            head[b, 1:ln] = torch.where(h < i, h, h + 1)  # any head but the word itself
            rel[b, 1:ln] = torch.randint(0, O_, (ln - 1,), generator=g)
    return head, rel


def test_pos_backward_reference_fixture(dev, engine, golden):
    """TaskHeads.pos_backward against the reference's pos_forward(use_soft_pos=False) + loss.backward()
    (pos_backward_small.npz): the loss, the .grad of upos_hid / upos_clf, the gradient w.r.t. the encoder rows."""
    from phonlp_amd import HeadWeights, TaskHeads
    z = golden("pos_backward_small.npz")
    heads = TaskHeads(HeadWeights.from_state_dict(_params(z), dev))
    for c in range(int(z["n_cases"])):
        lens = z[f"c{c}_lens"].tolist()
        out = heads.pos_backward(torch.from_numpy(z[f"c{c}_enc"]).to(dev), lens, torch.from_numpy(z[f"c{c}_upos"]))
        assert abs(float(out["loss"][0]) - float(z[f"c{c}_loss"])) <= 1e-5 * abs(float(z[f"c{c}_loss"]))
        pairs = [(out["grads"][k], z[f"c{c}_dp_{k}"], k) for k in ("upos_hid.weight", "upos_hid.bias", "upos_clf.weight",
                                                                 "upos_clf.bias")]
        pairs.append((out["d_enc"], z[f"c{c}_d_enc"], "d_enc"))
        for got, want, name in pairs:
            want = torch.from_numpy(want)
            scale = float(want.abs().max()) + 1e-12
            err = float((got.cpu() - want).abs().max())
            assert err <= 2e-4 * scale, f"case {c} {name}: max error {err:.3e} against scale {scale:.3e}"


def test_parser_backward_reference_fixture(dev, engine, golden):
    """TaskHeads.parser_backward: the gradients the reference's loss.backward() left in the parameters of the four
    scorers and of dep_hid (dep_loss_grad_small.npz, the same backward call as above), through phonlp_bilinear_backward /
    phonlp_label_backward / phonlp_linear_backward.  2e-4 of each gradient's largest entry."""
    from phonlp_amd import HeadWeights, TaskHeads
    z = golden("dep_loss_grad_small.npz")
    heads = TaskHeads(HeadWeights.from_state_dict(_params(z), dev))
    names = [k[len("c0_dp_"):] for k in z.files if k.startswith("c0_dp_")]
    assert len(names) == 27  # 4 scorers x 5, dep_hid x 2, the tag embedding, upos_hid / upos_clf x 2
    for c in range(int(z["n_cases"])):
        lens = z[f"c{c}_lens"].tolist()
        out = heads.parser_backward(torch.from_numpy(z[f"c{c}_enc"]).to(dev), lens, torch.from_numpy(z[f"c{c}_head"]),
                                    torch.from_numpy(z[f"c{c}_rel"]), int(z[f"c{c}_nwords"]))
        for name in names:
            want = torch.from_numpy(z[f"c{c}_dp_{name}"])
            got = out["grads"][name].cpu()
            assert got.shape == want.shape, (name, got.shape, want.shape)
            scale = float(want.abs().max()) + 1e-12
            err = float((got - want).abs().max())
            assert err <= 2e-4 * scale, f"case {c} {name}: max error {err:.3e} against scale {scale:.3e}"
        L = max(lens)
        live = torch.arange(L)[None, :] < torch.tensor(lens)[:, None]
        assert not out["d_feat"].cpu()[~live].any()
        want = torch.from_numpy(z[f"c{c}_d_enc"])  # the encoder's last layer at the words' first sub-words
        err = float((out["d_enc"].cpu() - want).abs().max())
        assert err <= 2e-4 * float(want.abs().max()), f"case {c} d_enc: max error {err:.3e}"


@pytest.mark.parametrize("lin,dist", [(True, True), (False, True), (True, False), (False, False)])
def test_parser_backward_vs_oracle_autograd(dev, lin, dist):
    """The same at the real head sizes (E=768, H=400, O=33) and with the linearization / distance scorers switched off
    (args of the reference, annotate_model.py:77-94): torch autograd through the oracle's dep_heads + dep_loss (pinned
    to the reference by the fixtures above) gives the gradients of every parameter the loss reaches."""
    from phonlp_amd import HeadWeights, TaskHeads
    from phonlp_amd import synthetic as S
    B, L = 4, 24
    sd = S.make_state_dict()
    enc_dep, _ = S.make_encoder_outputs(B, L, seed=77)
    lens1 = [24, 17, 9, 3]
    head, rel = _gold_trees(lens1, L, 33, seed=5)
    nw = sum(lens1)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    enc_leaf = enc_dep.clone().requires_grad_(True)
    dep = O.dep_heads(enc_leaf, lens1, p, linearization=lin, distance=dist)
    total, _ = O.dep_loss(dep, lens1, head[:, 1:].numpy(), rel[:, 1:].numpy(), nw, linearization=lin, distance=dist)
    total.backward()
    heads = TaskHeads(HeadWeights.from_state_dict(sd, dev, linearization=lin, distance=dist))
    out = heads.parser_backward(enc_dep.to(dev), lens1, head, rel, nw)
    scorers = ["unlabeled", "deprel"] + (["linearization"] if lin else []) + (["distance"] if dist else [])
    names = [f"{s_}.{k}" for s_ in scorers for k in ("W1.weight", "W1.bias", "W2.weight", "W2.bias", "scorer.W_bilin.weight")]
    names += ["dep_hid.weight", "dep_hid.bias", "upos_emb_matrix_dep", "upos_hid.weight", "upos_hid.bias", "upos_clf.weight",
              "upos_clf.bias"]
    assert sorted(out["grads"]) == sorted(names)
    for name in names:
        want = p[name].grad
        scale = float(want.abs().max()) + 1e-12
        err = float((out["grads"][name].cpu() - want).abs().max())
        assert err <= 3e-4 * scale, f"{name}: max error {err:.3e} against scale {scale:.3e}"
    err = float((out["d_enc"].cpu() - enc_leaf.grad).abs().max())
    assert err <= 3e-4 * float(enc_leaf.grad.abs().max())


def test_ner_backward_reference_fixture(dev, engine, golden):
    """TaskHeads.ner_backward against the reference's ner_forward + loss.backward() (ner_backward_small.npz): CRF NLL ->
    ReLU -> ner_tag_clf -> [enc ; soft POS embedding] -> the POS layers; every parameter's .grad and the encoder rows'."""
    from phonlp_amd import HeadWeights, TaskHeads
    z = golden("ner_backward_small.npz")
    heads = TaskHeads(HeadWeights.from_state_dict(_params(z), dev))
    names = [k[len("c0_dp_"):] for k in z.files if k.startswith("c0_dp_")]
    assert len(names) == 8
    for c in range(int(z["n_cases"])):
        lens = z[f"c{c}_lens"].tolist()
        out = heads.ner_backward(torch.from_numpy(z[f"c{c}_enc"]).to(dev), lens, torch.from_numpy(z[f"c{c}_tags"]))
        pairs = [(out["grads"][k], z[f"c{c}_dp_{k}"], k) for k in names] + [(out["d_enc"], z[f"c{c}_d_enc"], "d_enc")]
        for got, want, name in pairs:
            want = torch.from_numpy(want)
            assert got.shape == want.shape, (name, got.shape, want.shape)
            scale = float(want.abs().max()) + 1e-12
            err = float((got.cpu() - want).abs().max())
            assert err <= 2e-4 * scale, f"case {c} {name}: max error {err:.3e} against scale {scale:.3e}"


@pytest.mark.parametrize("B,L,ragged,lin,dist", [(8, 64, True, True, True), (6, 128, False, True, True),
                                                 (5, 40, True, False, True), (5, 40, True, True, False),
                                                 (4, 33, True, False, False), (2, 200, True, True, True)])
def test_dep_loss_vs_oracle(dev, engine, B, L, ragged, lin, dist):
    from phonlp_amd import HeadWeights, TaskHeads
    from phonlp_amd import synthetic as S
    sd = S.make_state_dict()
    enc_dep, _ = S.make_encoder_outputs(B, L)
    lens1 = S.make_sentlens(B, L, ragged)
    head, rel = _gold_trees(lens1, L, sd["deprel.scorer.W_bilin.weight"].shape[2], seed=B * 1000 + L)
    rel[0, 1] = -1  # an ignored word (ignore_index)
    heads = TaskHeads(HeadWeights.from_state_dict(sd, dev, linearization=lin, distance=dist))
    got = heads.parser_loss(enc_dep.to(dev), lens1, head, rel).cpu().numpy()
    od = O.dep_heads(enc_dep, lens1, sd, linearization=lin, distance=dist)
    total, terms = O.dep_loss(od, lens1, head[:, 1:], rel[:, 1:], sum(lens1), linearization=lin, distance=dist)
    for k in range(4):
        assert _rel(got[k], terms[k]) <= 1e-4, (k, got, [float(t) for t in terms])
    assert _rel(got[4], total) <= 1e-4
    if not lin:
        assert got[2] == 0.0
    if not dist:
        assert got[3] == 0.0


def test_dep_loss_same_value_twice_and_argument_errors(dev):
    """The sums are formed in a fixed order: two runs give the same bits.  Bad arguments come back as error codes."""
    from phonlp_amd import HeadWeights, TaskHeads, ops
    from phonlp_amd import synthetic as S
    from phonlp_amd._lib import HeadsLibraryError
    sd = S.make_state_dict()
    B, L = 4, 48
    enc_dep, _ = S.make_encoder_outputs(B, L)
    lens1 = S.make_sentlens(B, L, True)
    head, rel = _gold_trees(lens1, L, 33, seed=5)
    heads = TaskHeads(HeadWeights.from_state_dict(sd, dev))
    a = heads.parser_loss(enc_dep.to(dev), lens1, head, rel).cpu()
    b = heads.parser_loss(enc_dep.to(dev), lens1, head[:, 1:], rel[:, 1:]).cpu()  # the reference's B x (L-1) form
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    w = heads.w
    dep = heads.parser(enc_dep.to(dev), lens1, want_raw=True, want_label_argmax=False)
    XF, T = ops.label_project(dep["hidden_out"], w.lab_W12, w.lab_b12, w.lab_Wbt, w.H, w.O)
    h32, r32 = head.to(dev, torch.int32), rel.to(dev, torch.int32)
    with pytest.raises(HeadsLibraryError):  # linearization asked for without the raw scorer outputs
        ops.dep_loss(dep["arc_logp"], None, T, XF, h32, r32, dep["sentlens"], w.H, w.O, True, True, 10.0)
    with pytest.raises(HeadsLibraryError):
        ops.dep_loss(dep["arc_logp"], dep["raw"], T, XF, h32, r32, dep["sentlens"], w.H, w.O, True, True, 0.0)


def test_pos_loss_reference_fixture(dev, engine, golden):
    """POS loss against the value the reference's pos_forward(use_soft_pos=False) returned (crit_pos: mean,
    ignore_index 1)."""
    from phonlp_amd import HeadWeights, TaskHeads
    z = golden("dep_loss_small.npz")
    heads = TaskHeads(HeadWeights.from_state_dict(_params(z), dev))
    for c in range(int(z["n_cases"])):
        lens = z[f"c{c}_tag_lens"].tolist()
        upos = z[f"c{c}_upos"]
        got = heads.pos_loss(torch.from_numpy(z[f"c{c}_tag_enc"]).to(dev), lens, torch.from_numpy(upos)).cpu().numpy()
        assert _rel(got[0], z[f"c{c}_pos_loss"]) <= 1e-4, (c, got, z[f"c{c}_pos_loss"])
        counted = sum(int((upos[b, :l] != 1).sum()) for b, l in enumerate(lens))
        assert got[2] == counted and _rel(got[1], got[0] * counted) <= 1e-5


@pytest.mark.parametrize("B,L,ragged", [(16, 64, True), (256, 127, False), (3, 300, True)])
def test_pos_loss_vs_oracle(dev, engine, B, L, ragged):
    from phonlp_amd import HeadWeights, TaskHeads
    from phonlp_amd import synthetic as S
    sd = S.make_state_dict()
    _, enc_tag = S.make_encoder_outputs(B, L + 1)
    lens = [l - 1 for l in S.make_sentlens(B, L + 1, ragged)]
    U = sd["upos_clf.weight"].shape[0]
    gold = torch.randint(0, U, (B, L), generator=torch.Generator().manual_seed(B + L))
    heads = TaskHeads(HeadWeights.from_state_dict(sd, dev))
    got = heads.pos_loss(enc_tag.to(dev), lens, gold).cpu().numpy()
    logits, _ = O.pos_head(enc_tag, lens, sd, "upos_emb_matrix_ner")
    assert _rel(got[0], O.pos_loss(logits, lens, gold)) <= 1e-4
    # nothing counted -> NaN, like torch's mean over an empty selection
    none = heads.pos_loss(enc_tag.to(dev), lens, torch.ones(B, L, dtype=torch.int64)).cpu().numpy()
    assert np.isnan(none[0]) and none[2] == 0


@pytest.mark.parametrize("N,U", [(1000, 24), (37, 6), (5, 70)])
def test_pos_cross_entropy_module_backward(dev, N, U):
    """compat.PosCrossEntropy against the reference's criterion itself -- crit_pos IS nn.CrossEntropyLoss(ignore_index=1)
    (jointmodel/model.py:94) -- value and gradient w.r.t. the logits, on packed rows as at :134."""
    from phonlp_amd.compat import PosCrossEntropy
    g = torch.Generator().manual_seed(N + U)
    x = torch.randn(N, U, generator=g) * 2
    t = torch.randint(0, U, (N,), generator=g)
    xr = x.clone().requires_grad_(True)
    want = torch.nn.CrossEntropyLoss(ignore_index=1)(xr, t)
    (3.0 * want).backward()
    xg = x.clone().requires_grad_(True)
    got = PosCrossEntropy()(xg, t)
    (3.0 * got).backward()
    assert _rel(got.detach().cpu(), want.detach()) <= 1e-5
    torch.testing.assert_close(xg.grad, xr.grad, rtol=1e-4, atol=1e-7)


# ------------------------------------------------------------------------------------------ backward w.r.t. the scores
def test_dep_loss_grad_reference_fixture(dev, engine, golden):
    """phonlp_dep_loss_grad against the gradients the reference's own graph sent into its four scorers
    (dep_loss_grad_small.npz: autograd of jointmodel/model.py:209-236 under loss.backward(), model in .eval()).
    Tolerance 1e-4 relative + 1e-6 absolute (gradients are soft-max probabilities / number_of_words)."""
    from phonlp_amd import HeadWeights, TaskHeads
    z = golden("dep_loss_grad_small.npz")
    heads = TaskHeads(HeadWeights.from_state_dict(_params(z), dev))
    for c in range(int(z["n_cases"])):
        lens = z[f"c{c}_lens"].tolist()
        g = heads.parser_loss_grad(torch.from_numpy(z[f"c{c}_enc"]).to(dev), lens, torch.from_numpy(z[f"c{c}_head"]),
                                   torch.from_numpy(z[f"c{c}_rel"]), int(z[f"c{c}_nwords"]))
        tt = lambda k: torch.from_numpy(z[f"c{c}_{k}"])
        torch.testing.assert_close(g["d_unlabeled"].cpu(), tt("d_unlabeled"), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(g["d_label"].cpu()[:, 1:], tt("d_label"), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(g["d_lin"].cpu()[:, 1:], tt("d_lin"), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(g["d_dist"].cpu()[:, 1:], tt("d_dist"), rtol=1e-4, atol=1e-6)
        assert not g["d_label"][:, 0].any() and not g["d_lin"][:, 0].any() and not g["d_dist"][:, 0].any()


@pytest.mark.parametrize("B,L,ragged,lin,dist", [(8, 64, True, True, True), (6, 128, False, True, True),
                                                 (5, 40, True, False, True), (4, 33, True, False, False),
                                                 (2, 200, True, True, True)])
def test_dep_loss_grad_vs_oracle(dev, engine, B, L, ragged, lin, dist):
    """The same at the real head sizes against the oracle's closed form (itself pinned to the reference's autograd by
    tests/test_oracle_golden.py::test_dep_loss_gradients)."""
    from phonlp_amd import HeadWeights, TaskHeads
    from phonlp_amd import synthetic as S
    sd = S.make_state_dict()
    enc_dep, _ = S.make_encoder_outputs(B, L)
    lens1 = S.make_sentlens(B, L, ragged)
    head, rel = _gold_trees(lens1, L, 33, seed=B * L)
    nw = sum(lens1)
    heads = TaskHeads(HeadWeights.from_state_dict(sd, dev, linearization=lin, distance=dist))
    g = heads.parser_loss_grad(enc_dep.to(dev), lens1, head, rel, nw)
    dep = O.dep_heads(enc_dep, lens1, sd, linearization=lin, distance=dist)
    want = O.dep_loss_grad(dep, lens1, head[:, 1:], rel[:, 1:], nw, linearization=lin, distance=dist)
    for k in ("d_unlabeled", "d_label", "d_lin", "d_dist"):
        torch.testing.assert_close(g[k].cpu(), want[k], rtol=2e-4, atol=1e-6, msg=lambda m: f"{k}: {m}")
