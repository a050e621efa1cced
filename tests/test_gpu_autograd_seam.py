"""The three training losses as autograd nodes on the HIP heads (phonlp_amd.compat.parser_loss / pos_loss / ner_loss and the
`update` seam of patch_trainer, ref jointmodel/trainer.py:62-99 over jointmodel/model.py:107-281): after loss.backward()
the parameters' .grad and the encoder rows' hold what the reference's own loss.backward() left there -- the fixtures
dep_loss_grad_small.npz / pos_backward_small.npz / ner_backward_small.npz (recorded from the reference, model in .eval()).
Tolerance 2e-4 of each gradient's largest entry, loss values 1e-4 relative."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from phonlp_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _leaf_params(z, dev):
    return {k[2:]: torch.from_numpy(z[k]).to(dev).requires_grad_(True) for k in z.files if k.startswith("p_")}


def _check(got, want, name):
    want = torch.from_numpy(want)
    assert got is not None, name
    assert tuple(got.shape) == tuple(want.shape), (name, got.shape, want.shape)
    scale = float(want.abs().max()) + 1e-12
    err = float((got.cpu() - want).abs().max())
    assert err <= 2e-4 * scale, f"{name}: max error {err:.3e} against scale {scale:.3e}"


@pytest.mark.parametrize("task", ["parser", "pos", "ner"])
def test_loss_nodes_leave_the_references_gradients(dev, golden, task):
    from phonlp_amd import HeadWeights, TaskHeads, compat
    z = golden({"parser": "dep_loss_grad_small.npz", "pos": "pos_backward_small.npz", "ner": "ner_backward_small.npz"}[task])
    names = [k[len("c0_dp_"):] for k in z.files if k.startswith("c0_dp_")]
    for c in range(int(z["n_cases"])):
        params = _leaf_params(z, dev)
        heads = TaskHeads(HeadWeights.from_state_dict({k: v.detach() for k, v in params.items()}, dev))
        enc = torch.from_numpy(z[f"c{c}_enc"]).to(dev).requires_grad_(True)
        lens = z[f"c{c}_lens"].tolist()
        if task == "parser":
            loss = compat.parser_loss(heads, params, enc, lens, torch.from_numpy(z[f"c{c}_head"]), torch.from_numpy(z[f"c{c}_rel"]),
                                      int(z[f"c{c}_nwords"]))
        elif task == "pos":
            loss = compat.pos_loss(heads, params, enc, lens, torch.from_numpy(z[f"c{c}_upos"]))
            assert abs(float(loss.detach()) - float(z[f"c{c}_loss"])) <= 1e-4 * abs(float(z[f"c{c}_loss"]))
        else:
            loss = compat.ner_loss(heads, params, enc, lens, torch.from_numpy(z[f"c{c}_tags"]))
        assert loss.requires_grad and loss.dim() == 0
        (3.0 * loss).backward()  # the incoming gradient scales everything
        for name in names:
            _check(params[name].grad / 3.0, z[f"c{c}_dp_{name}"], f"case {c} {name}")
        _check(enc.grad / 3.0, z[f"c{c}_d_enc"], f"case {c} d_enc")
        untouched = [k for k in params if k not in names]
        assert all(params[k].grad is None for k in untouched), "a loss must not reach the other heads' parameters"


class _StubTrainer:
    def __init__(self, model):
        self.model, self.vocab, self.use_cuda = model, model.vocab, True


def test_update_seam_matches_the_references_losses_and_reaches_the_encoder(golden):
    """patch_trainer(train_seam=True).update on the three batches of trainer_small.npz: the four values the reference's
    forward returned (joint_losses; dropouts off), every head parameter and the encoder with a gradient, and a second call
    after an optimizer-like step sees the new weights (the packs follow the parameters)."""
    from phonlp_amd.compat import patch_trainer
    from tests.stub_model import StubJointModel
    z = golden("trainer_small.npz")
    model = StubJointModel(z)
    model.args.update({"dropout": 0.0, "word_dropout": 0.0})
    for m in model.phobert.modules():  # update() trains: the stub encoder's own dropout off, so the rows are the fixture's
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    for v in model._heads.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    trainer = patch_trainer(_StubTrainer(model), move_encoder=False, train_seam=True)
    T = lambda k: torch.from_numpy(z[k])
    dep_batch = (T("dep_tokens"), T("dep_first"), T("dep_mask"), T("loss_gold_head"), T("dep_gold_rel"),
                 int(z["dep_nwords"]), z["dep_oidx"].tolist(), z["dep_lens"].tolist())
    pos_batch = (T("tag_tokens"), T("tag_first"), T("gold_upos"), z["tag_oidx"].tolist(), z["tag_lens"].tolist())
    ner_batch = (T("tag_tokens"), T("tag_first"), T("tag_mask"), T("gold_ner"), z["tag_oidx"].tolist(), z["tag_lens"].tolist())
    got = trainer.update(dep_batch, pos_batch, ner_batch, 0.4, 0.2, 0.4)
    np.testing.assert_allclose(got, z["joint_losses"], rtol=1e-4)
    assert model.training
    # every parameter the reference's backward reaches (the names of the three backward fixtures; its W_bilin.bias is never
    # added, biaffine.py:27-35, and the drop_replacement_* vectors only act under word dropout) has a gradient now
    reached = set()
    for f in ("dep_loss_grad_small.npz", "pos_backward_small.npz", "ner_backward_small.npz"):
        reached |= {k[len("c0_dp_"):] for k in golden(f).files if k.startswith("c0_dp_")}
    assert len(reached) == 31  # 27 (parser) + ner_tag_clf x 2, the transitions, the NER tag embedding; the POS layers are shared
    missing = [k for k in sorted(reached) if model._heads[k].grad is None or not bool(model._heads[k].grad.abs().sum() > 0)]
    assert not missing, missing
    enc_grads = [p.grad for p in model.phobert.parameters() if p.requires_grad]
    assert any(g is not None and bool(g.abs().sum() > 0) for g in enc_grads), "nothing reached the encoder"
    # a step on the head weights, then update again: the loss moves (the packs are refreshed from the live tensors)
    with torch.no_grad():
        for k, v in model._heads.items():
            if v.is_floating_point() and v.grad is not None:
                v.sub_(0.05 * v.grad)
                v.grad = None
    got2 = trainer.update(dep_batch, pos_batch, ner_batch, 0.4, 0.2, 0.4)
    assert got2[0] < got[0], (got, got2)


def test_update_seam_with_head_dropout(golden):
    """update() with the heads' dropouts on: masks are drawn per step (compat.draw_masks) -- two calls give different,
    finite losses, the replacement vectors get gradients, and the share of dropped words / kept units is what p says."""
    from phonlp_amd.compat import patch_trainer, draw_masks, refresh_heads
    from tests.stub_model import StubJointModel
    z = golden("trainer_small.npz")
    model = StubJointModel(z)
    model.args.update({"dropout": 0.33, "word_dropout": 0.2})
    g = torch.Generator().manual_seed(4)
    E, D = model._heads["upos_hid.weight"].shape[1], model._heads["dep_hid.weight"].shape[1]
    model._heads.update({"drop_replacement_pos": torch.randn(E, generator=g) / E ** 0.5,
                         "drop_replacement_ner": torch.randn(D, generator=g) / D ** 0.5,
                         "drop_replacement_dep": torch.randn(D, generator=g) / D ** 0.5})
    for m in model.phobert.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    for v in model._heads.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    trainer = patch_trainer(_StubTrainer(model), move_encoder=False, train_seam=True)
    T = lambda k: torch.from_numpy(z[k])
    dep_batch = (T("dep_tokens"), T("dep_first"), T("dep_mask"), T("loss_gold_head"), T("dep_gold_rel"),
                 int(z["dep_nwords"]), z["dep_oidx"].tolist(), z["dep_lens"].tolist())
    pos_batch = (T("tag_tokens"), T("tag_first"), T("gold_upos"), z["tag_oidx"].tolist(), z["tag_lens"].tolist())
    ner_batch = (T("tag_tokens"), T("tag_first"), T("tag_mask"), T("gold_ner"), z["tag_oidx"].tolist(), z["tag_lens"].tolist())
    a = trainer.update(dep_batch, pos_batch, ner_batch, 0.4, 0.2, 0.4)
    b = trainer.update(dep_batch, pos_batch, ner_batch, 0.4, 0.2, 0.4)
    assert all(np.isfinite(a)) and all(np.isfinite(b)) and a != b
    for k in ("drop_replacement_pos", "drop_replacement_ner", "drop_replacement_dep", "dep_hid.weight"):
        assert model._heads[k].grad is not None and bool(torch.isfinite(model._heads[k].grad).all())
    h = refresh_heads(model)
    lens = [50] * 64
    mk = draw_masks(h, "parser", 64, 50, lens, 0.33, 0.2, h.w.trans.device, torch.Generator(device=h.w.trans.device).manual_seed(1))
    assert abs(float(mk["wd"].float().mean()) - 0.2) < 0.03
    kept = (mk["hidden"]["deprel"][1] > 0).float().mean()
    assert abs(float(kept) - 0.67) < 0.02 and abs(float(mk["hidden"]["deprel"][1].max()) - 1 / 0.67) < 1e-5


@pytest.mark.parametrize("task", ["parser", "ner", "pos"])
def test_train_mode_with_the_references_dropout_masks(dev, golden, task):
    """The three losses with the model in .train() (ref jointmodel/model.py:107-236: word dropout + replacement vectors,
    nn.Dropout on the hidden states and inside the scorers), the masks the reference's own modules drew given as inputs
    (train_dropout_small.npz: read off those modules by forward hooks): loss, every parameter's .grad -- the replacement
    vectors' included -- and the encoder rows' gradient equal the reference's."""
    from phonlp_amd import HeadWeights, TaskHeads, compat
    z = golden("train_dropout_small.npz")
    tag = {"parser": "dep", "ner": "ner", "pos": "pos"}[task]
    order = ("unlabeled", "deprel", "linearization", "distance")  # the order dep_forward calls them in (:185-200)
    for c in range(int(z["n_cases"])):
        params = {k[2:]: torch.from_numpy(z[k]).to(dev).requires_grad_(True) for k in z.files
                  if k.startswith("p_") and z[k].ndim > 0}
        heads = TaskHeads(HeadWeights.from_state_dict({k: v.detach() for k, v in params.items()}, dev))
        enc = torch.from_numpy(z[f"{tag}{c}_enc"]).to(dev).requires_grad_(True)
        lens = z[f"{tag}{c}_lens"].tolist()
        m = lambda k: torch.from_numpy(z[f"{tag}{c}_m_{k}"])
        masks = {"wd_pos": m("wd_pos"), "drop_pos": m("drop_pos")}
        if task == "parser":
            mh, ms = m("hidden"), m("scorer")
            masks.update({"wd": m("wd"), "hidden": {n: (mh[i, 0], mh[i, 1]) for i, n in enumerate(order)},
                          "scorer": {n: (ms[i, 0], ms[i, 1]) for i, n in enumerate(order)}})
            loss = compat.parser_loss(heads, params, enc, lens, torch.from_numpy(z[f"dep{c}_head"]), torch.from_numpy(z[f"dep{c}_rel"]),
                                      int(z[f"dep{c}_nwords"]), masks=masks)
        elif task == "ner":
            masks["wd"] = m("wd")
            loss = compat.ner_loss(heads, params, enc, lens, torch.from_numpy(z[f"ner{c}_tags"]), masks=masks)
        else:
            loss = compat.pos_loss(heads, params, enc, lens, torch.from_numpy(z[f"pos{c}_upos"]), masks=masks)
        want = float(z[f"{tag}{c}_loss"])
        assert abs(float(loss.detach()) - want) <= 1e-4 * abs(want), (float(loss.detach()), want)
        loss.backward()
        names = [k[len(f"{tag}{c}_dp_"):] for k in z.files if k.startswith(f"{tag}{c}_dp_")]
        assert f"drop_replacement_{tag}" in names or task == "pos"
        assert "drop_replacement_pos" in names
        for name in names:
            _check(params[name].grad, z[f"{tag}{c}_dp_{name}"], f"{task} case {c} {name}")
        _check(enc.grad, z[f"{tag}{c}_d_enc"], f"{task} case {c} d_enc")
