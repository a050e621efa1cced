"""`phonlp_amd.compat`: the reference-shaped seams (patch(model).annotate, tagger_forward, dep_forward,
viterbi_decode, chuliu_edmonds_one_root, CRFLoss, DeepBiaffineScorer) against outputs recorded from the
reference's own JointModel (tests/golden/joint_small.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def patched(golden):
    assert torch.cuda.is_available()
    from phonlp_amd.compat import patch
    from tests.stub_model import StubJointModel
    z = golden("joint_small.npz")
    # the encoder stays on the CPU so that the heads see the encoder rows the fixture was recorded with; a GPU
    # encoder differs in the last bits, which can flip an arg-max of this tiny random model at a near-tie
    return patch(StubJointModel(z), move_encoder=False), z


def test_gpu_encoder_path_runs(golden):
    from phonlp_amd.compat import patch
    from tests.stub_model import StubJointModel
    z = golden("joint_small.npz")
    model = patch(StubJointModel(z))  # encoder on the GPU: same output up to near-ties
    data, pos, ner, dep = model.annotate(text=z["sent_text"].tolist()[0])
    want = z["annotate_text_dep_head"].tolist()
    agree = sum(a[0] == b for a, b in zip(dep[0], want))
    assert len(dep[0]) == len(want) and agree >= len(want) - 2


def test_annotate_file_matches_reference_output(patched, tmp_path):
    model, z = patched
    fin, fout = tmp_path / "in.txt", tmp_path / "out.txt"
    fin.write_text("\n".join(z["sent_text"].tolist()) + "\n")
    model.annotate(input_file=str(fin), output_file=str(fout), batch_size=5)
    assert fout.read_text() == str(z["annotate_file"])
    model.annotate(input_file=str(fin), output_file=str(fout), batch_size=5, output_type="conll")
    assert fout.read_text() == str(z["annotate_file_conll"])
    # One batch holding all 14 sentences.  The reference calls the encoder without an attention mask, so its
    # outputs (and a few decisions) depend on how much padding a batch carries; the single-batch run is
    # therefore compared with the fixture's single-batch decode (dec_*), not with the batch_size=5 file.
    model.annotate(input_file=str(fin), output_file=str(fout), batch_size=64)
    sents = [blk.split("\n") for blk in fout.read_text().strip().split("\n\n")]
    # the batch's row order: chunk_batches sorts by (length, index) descending, get_batch sorts the chunk again
    # by (length, position) descending, which flips equal-length neighbours back
    first = sorted(range(len(sents)), key=lambda i: (len(sents[i]), i), reverse=True)
    order = [first[k] for k in sorted(range(len(first)), key=lambda k: (len(sents[first[k]]), k), reverse=True)]
    heads = [int(ln.split("\t")[4]) for i in order for ln in sents[i]]
    rels = [ln.split("\t")[5] for i in order for ln in sents[i]]
    ners = [ln.split("\t")[3] for i in order for ln in sents[i]]
    assert heads == z["dec_heads"].tolist()
    assert rels == model.vocab["deprel"].unmap(z["dec_rels"])
    assert ners == model.vocab["ner_tag"].unmap(z["dec_ner"])


def test_annotate_text_matches_reference_output(patched):
    model, z = patched
    data, pos, ner, dep = model.annotate(text=z["sent_text"].tolist()[0])
    assert [p[0] for p in pos[0]] == z["annotate_text_pos"].tolist()
    assert ner[0] == z["annotate_text_ner"].tolist()
    assert [d[0] for d in dep[0]] == z["annotate_text_dep_head"].tolist()
    assert [d[1] for d in dep[0]] == z["annotate_text_dep_rel"].tolist()


def test_forward_seams_types_and_values(patched):
    model, z = patched
    sents = [s.split(" ") for s in z["sent_text"].tolist()]
    chunks = model.chunk_batches(model.process_data_parser(batch_text=sents), len(sents))
    tok1, fsw1, _, _, _, lens1 = model.get_batch(0, chunks)
    chunks = model.chunk_batches(model.process_data_tagger(batch_text=sents), len(sents))
    tok, fsw, _, _, _, lens = model.get_batch(0, chunks)
    assert lens1 == z["dep_sentlens"].tolist() and lens == z["tag_sentlens"].tolist()
    with torch.no_grad():
        arc, lab = model.dep_forward(tok1, fsw1, lens1)
        preds_pos, logits = model.tagger_forward(tok, fsw, lens)
    assert isinstance(arc, np.ndarray) and arc.dtype == np.float32 and lab.dtype == np.int64
    fin = np.isfinite(z["dep_arc_logp"])
    np.testing.assert_allclose(arc[fin], z["dep_arc_logp"][fin], rtol=1e-3, atol=1e-4)
    assert np.array_equal(lab, z["dep_label_argmax"])
    assert preds_pos[0].dtype == torch.int64 and np.array_equal(preds_pos[0].cpu().numpy(), z["tag_pos_pred"])
    np.testing.assert_allclose(logits.cpu().numpy(), z["tag_ner_logits"], rtol=1e-3, atol=1e-5)


def test_free_functions(golden):
    from phonlp_amd.compat import CRFLoss, DeepBiaffineScorer, chuliu_edmonds_one_root, viterbi_decode
    z = golden("viterbi.npz")
    for c in (2, 5, 7):
        tags, best = viterbi_decode(z[f"v{c}_scores"], z[f"v{c}_trans"])
        assert tags == z[f"v{c}_tags"].tolist() and best == z[f"v{c}_best"]
    m = golden("mst.npz")
    l = int(m["sizes"][10])
    off = int((m["sizes"][:10] ** 2).sum())
    hoff = int(m["sizes"][:10].sum())
    got = chuliu_edmonds_one_root(m["flat"][off:off + l * l].reshape(l, l))
    assert got.dtype == np.int64 and got.tolist() == m["heads"][hoff:hoff + l].tolist()
    with pytest.raises(AssertionError):
        chuliu_edmonds_one_root(np.zeros((1, 1), np.float32))
    c = golden("crf.npz")
    crit = CRFLoss(c["c0_trans"].shape[0])
    crit._transitions.data.copy_(torch.from_numpy(c["c0_trans"]))
    loss, trans = crit(torch.from_numpy(c["c0_inputs"]), torch.from_numpy(c["c0_masks"]), torch.from_numpy(c["c0_tags"]))
    assert abs(float(loss.detach()) - float(c["c0_loss"])) < 1e-4 * max(1.0, abs(float(c["c0_loss"])))
    b = golden("biaffine.npz")
    p = {k[len("db1_p_"):]: torch.from_numpy(b[k]) for k in b.files if k.startswith("db1_p_")}
    h = torch.nn.functional.pad(torch.from_numpy(b["db1_h"]), (0, 2))  # K % 4 == 0 for the kernels
    sc = DeepBiaffineScorer(12, 12, 7, 4)
    with torch.no_grad():
        sc.W1.weight.copy_(torch.nn.functional.pad(p["W1.weight"], (0, 2))); sc.W1.bias.copy_(p["W1.bias"])
        sc.W2.weight.copy_(torch.nn.functional.pad(p["W2.weight"], (0, 2))); sc.W2.bias.copy_(p["W2.bias"])
        sc.scorer.W_bilin.weight.copy_(p["scorer.W_bilin.weight"])
    hd = h.cuda()
    out = sc(hd, hd)
    assert tuple(out.shape) == tuple(b["db1_out"].shape)
    np.testing.assert_allclose(out.cpu().numpy(), b["db1_out"], rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------------------------ trainer seams
class _StubTrainer:
    """Stand-in for the reference's JointTrainer (models/jointmodel/trainer.py): what patch_trainer() touches."""

    def __init__(self, model):
        self.model, self.vocab, self.use_cuda = model, model.vocab, True


def test_patch_trainer_predict_matches_reference(golden):
    """predict_dep / predict_pos / predict_ner of a patched trainer against the strings the reference's own
    JointTrainer produced on the same batches (tests/golden/trainer_small.npz), sorted and unsorted."""
    from phonlp_amd.compat import patch_trainer
    from tests.stub_model import StubJointModel
    z = golden("trainer_small.npz")
    trainer = patch_trainer(_StubTrainer(StubJointModel(z)), move_encoder=False)
    T = lambda k: torch.from_numpy(z[k])
    dep_batch = (T("dep_tokens"), T("dep_first"), T("dep_mask"), T("dep_gold_head"), T("dep_gold_rel"),
                 int(z["dep_nwords"]), z["dep_oidx"].tolist(), z["dep_lens"].tolist())
    pos_batch = (T("tag_tokens"), T("tag_first"), T("gold_upos"), z["tag_oidx"].tolist(), z["tag_lens"].tolist())
    ner_batch = (T("tag_tokens"), T("tag_first"), T("tag_mask"), T("gold_ner"), z["tag_oidx"].tolist(),
                 z["tag_lens"].tolist())
    for unsort, tag in ((True, "u"), (False, "s")):
        dep = trainer.predict_dep(dep_batch, unsort=unsort)
        pos = trainer.predict_pos(pos_batch, unsort=unsort)
        ner = trainer.predict_ner(ner_batch, unsort=unsort)
        assert ["|".join(h + ":" + r for h, r in s) for s in dep] == z[f"dep_{tag}"].tolist()
        assert ["|".join(w[0] for w in s) for s in pos] == z[f"pos_{tag}"].tolist()
        got_ner, want_ner = ["|".join(s) for s in ner], z[f"ner_{tag}"].tolist()
        if got_ner != want_ner:
            # ReLU'd emissions of this tiny random model hold many exact zeros, so distinct tag paths can score
            # within fp32 rounding of each other; the decoder itself is bit-exact on identical logits
            # (test_gpu_parity.py).  A differing sentence must be such a near-tie under the CPU oracle's logits.
            from oracle import heads_oracle as O
            model = trainer.model
            sd = {k: v for k, v in model.state_dict().items() if not k.startswith("phobert.")}
            with torch.no_grad():
                emb = model.phobert(T("tag_tokens"))[2][-1]
            enc = torch.stack([emb[i].index_select(0, T("tag_first")[i]) for i in range(emb.size(0))])
            lens = z["tag_lens"].tolist()
            logits = O.tagger_heads(enc, lens, sd)["ner_logits"].numpy()
            trans = sd["crit_ner._transitions"].numpy()
            order = z["tag_oidx"].tolist()
            rows = {k: b for b, k in enumerate(order)} if unsort else {b: b for b in range(len(lens))}
            ids = model.vocab["ner_tag"].index

            def path_score(b, tags):
                t = [ids[x] for x in tags.split("|")]
                return sum(float(logits[b, i, t[i]]) for i in range(len(t))) + sum(float(trans[t[i - 1], t[i]]) for i in range(1, len(t)))
            n_diff = 0
            for k, (g_, w_) in enumerate(zip(got_ner, want_ner)):
                if g_ != w_:
                    n_diff += 1
                    b = rows[k]
                    assert abs(path_score(b, g_) - path_score(b, w_)) <= 1e-4 * max(1.0, abs(path_score(b, w_))), (k, g_, w_)
            assert n_diff <= 2


def test_patch_trainer_dev_loss_matches_reference(golden):
    """trainer.dev_loss against the four values the reference's JointModel.forward (jointmodel/model.py:243-281, the
    forward half of JointTrainer.update) returned in .eval() mode on the same three batches, lambdas 0.4 / 0.2 / 0.4.
    The encoder runs on the CPU (move_encoder=False) so the heads see the reference's encoder rows bit for bit."""
    from phonlp_amd.compat import patch_trainer
    from tests.stub_model import StubJointModel
    z = golden("trainer_small.npz")
    trainer = patch_trainer(_StubTrainer(StubJointModel(z)), move_encoder=False)
    T = lambda k: torch.from_numpy(z[k])
    dep_batch = (T("dep_tokens"), T("dep_first"), T("dep_mask"), T("loss_gold_head"), T("dep_gold_rel"),
                 int(z["dep_nwords"]), z["dep_oidx"].tolist(), z["dep_lens"].tolist())
    pos_batch = (T("tag_tokens"), T("tag_first"), T("gold_upos"), z["tag_oidx"].tolist(), z["tag_lens"].tolist())
    ner_batch = (T("tag_tokens"), T("tag_first"), T("tag_mask"), T("gold_ner"), z["tag_oidx"].tolist(),
                 z["tag_lens"].tolist())
    got = trainer.dev_loss(dep_batch, pos_batch, ner_batch, 0.4, 0.2, 0.4)
    np.testing.assert_allclose(got, z["joint_losses"], rtol=1e-4)


def test_patched_trainer_follows_weight_updates_and_eval_mode(golden):
    """the packs are copies, so the seams must re-pack after a training step; and they must put the model
    in eval mode like the reference (trainer.py:104,134,150), or the encoder's dropout stays on after update()."""
    from phonlp_amd.compat import patch_trainer
    from tests.stub_model import StubJointModel
    z = golden("trainer_small.npz")
    T = lambda k: torch.from_numpy(z[k])
    dep_batch = (T("dep_tokens"), T("dep_first"), T("dep_mask"), T("dep_gold_head"), T("dep_gold_rel"),
                 int(z["dep_nwords"]), z["dep_oidx"].tolist(), z["dep_lens"].tolist())
    pos_batch = (T("tag_tokens"), T("tag_first"), T("gold_upos"), z["tag_oidx"].tolist(), z["tag_lens"].tolist())
    model = StubJointModel(z)
    trainer = patch_trainer(_StubTrainer(model), move_encoder=False)
    before_dep, before_pos = trainer.predict_dep(dep_batch), trainer.predict_pos(pos_batch)
    # (1) train mode: RoBERTa's dropout (p = 0.1) would make the encoder rows random; the seams switch it off
    model.train()
    assert trainer.predict_dep(dep_batch) == before_dep and not model.training
    model.train()
    assert trainer.predict_pos(pos_batch) == before_pos and not model.training
    # (2) an in-place "optimizer step" on head weights after patching: predictions must be those of a fresh pack
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for k in ("unlabeled.scorer.W_bilin.weight", "deprel.scorer.W_bilin.weight", "upos_clf.weight"):
            model._heads[k].add_(torch.randn(model._heads[k].shape, generator=g) * 0.5)
    after_dep, after_pos = trainer.predict_dep(dep_batch), trainer.predict_pos(pos_batch)
    fresh = patch_trainer(_StubTrainer(model), move_encoder=False)
    assert after_dep == fresh.predict_dep(dep_batch) and after_pos == fresh.predict_pos(pos_batch)
    assert after_dep != before_dep and after_pos != before_pos
