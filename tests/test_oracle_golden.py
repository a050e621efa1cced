"""Pins oracle/heads_oracle.py (the CPU restatement) to outputs of the reference's own code.

The fixtures were produced by tests/golden/make_golden.py, which runs the reference modules
from /root/reference; nothing here needs the reference at run time.
"""
import numpy as np
import pytest
import torch

from oracle import heads_oracle as O


def _params(z, prefix="p_"):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def test_pairwise_bilinear_quirks(golden):
    z = golden("biaffine.npz")
    for c in range(int(z["n_pb"])):
        x1, x2, w = (torch.from_numpy(z[f"pb{c}_{k}"]) for k in ("x1", "x2", "weight"))
        out = O.pairwise_bilinear(x1, x2, w)
        ref = torch.from_numpy(z[f"pb{c}_out"])
        assert out.shape == ref.shape
        torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
        if w.shape[2] > 1 and w.shape[0] == w.shape[1]:
            # the documented (D1, D2, O) einsum must NOT match: the flat-reshape quirk is real
            doc = torch.einsum("bid,deo,bje->bijo", x1, w, x2)
            assert not torch.allclose(doc, ref, rtol=1e-3, atol=1e-3)


def test_deep_biaffine(golden):
    z = golden("biaffine.npz")
    for c in range(int(z["n_db"])):
        p = _params(z, f"db{c}_p_")
        h = torch.from_numpy(z[f"db{c}_h"])
        out = O.deep_biaffine(h, h, p["W1.weight"], p["W1.bias"], p["W2.weight"], p["W2.bias"],
                              p["scorer.W_bilin.weight"])
        torch.testing.assert_close(out, torch.from_numpy(z[f"db{c}_out"]), rtol=1e-5, atol=1e-5)


def test_viterbi(golden):
    z = golden("viterbi.npz")
    for c in range(int(z["n"])):
        tags, best = O.viterbi_decode(z[f"v{c}_scores"], z[f"v{c}_trans"])
        assert tags == z[f"v{c}_tags"].tolist(), f"case {c}"
        assert np.float32(best) == z[f"v{c}_best"], f"case {c}"  # single fp32 adds: bit-exact


def test_crf_log_norm_and_loss(golden):
    z = golden("crf.npz")
    for c in range(int(z["n"])):
        x, m, t, tr = (torch.from_numpy(z[f"c{c}_{k}"]) for k in ("inputs", "masks", "tags", "trans"))
        torch.testing.assert_close(O.crf_log_norm(x, m, tr), torch.from_numpy(z[f"c{c}_log_norm"]),
                                   rtol=1e-6, atol=1e-6)
        assert abs(float(O.crf_nll(x, m, t, tr)) - float(z[f"c{c}_loss"])) <= 1e-5 * max(1, abs(float(z[f"c{c}_loss"])))


def test_mst_one_root(golden):
    z = golden("mst.npz")
    sizes, flat, heads = z["sizes"], z["flat"], z["heads"]
    po = ph = 0
    for k, l in enumerate(sizes):
        m = flat[po:po + l * l].reshape(l, l)
        want = heads[ph:ph + l]
        got = O.chuliu_edmonds_one_root(m)
        assert got.tolist() == want.tolist(), f"case {k} kind {z['kinds'][k]} l={l}"
        po += l * l
        ph += l


@pytest.mark.parametrize("name", ["joint_small.npz", "joint_zero_init.npz"])
def test_joint_heads(golden, name):
    z = golden(name)
    p = _params(z)
    dep = O.dep_heads(torch.from_numpy(z["dep_enc"]), z["dep_sentlens"].tolist(), p)
    ref_arc = torch.from_numpy(z["dep_arc_logp"])
    fin = torch.isfinite(ref_arc)
    assert torch.equal(fin, torch.isfinite(dep["arc_logp"]))
    torch.testing.assert_close(dep["arc_logp"][fin], ref_arc[fin], rtol=1e-4, atol=1e-5)
    assert np.array_equal(dep["label_argmax"].numpy(), z["dep_label_argmax"])
    tag = O.tagger_heads(torch.from_numpy(z["tag_enc"]), z["tag_sentlens"].tolist(), p)
    assert np.array_equal(tag["pos_pred"].numpy(), z["tag_pos_pred"])
    torch.testing.assert_close(tag["ner_logits"], torch.from_numpy(z["tag_ner_logits"]), rtol=1e-5, atol=1e-6)
    # decoders driven the way annotate() drives them
    lens1, lens = z["dep_sentlens"].tolist(), z["tag_sentlens"].tolist()
    arc = z["dep_arc_logp"]
    heads = [O.chuliu_edmonds_one_root(arc[i][:l, :l])[1:] for i, l in enumerate(lens1)]
    assert np.concatenate(heads).tolist() == z["dec_heads"].tolist()
    rels = [z["dep_label_argmax"][i][j + 1][h] for i, hs in enumerate(heads) for j, h in enumerate(hs)]
    assert [int(r) for r in rels] == z["dec_rels"].tolist()
    trans = z["p_crit_ner._transitions"]
    ner = [O.viterbi_decode(z["tag_ner_logits"][i, :l], trans)[0] for i, l in enumerate(lens)]
    assert [t for s in ner for t in s] == z["dec_ner"].tolist()
    loss = O.crf_nll(torch.from_numpy(z["tag_ner_logits"]), torch.from_numpy(z["tag_words_mask"]),
                     torch.from_numpy(z["crf_gold"]), torch.from_numpy(trans))
    assert abs(float(loss) - float(z["crf_loss"])) <= 1e-5 * max(1.0, abs(float(z["crf_loss"])))


def test_ner_entity_scoring(golden):
    """decode_from_bioes / score_by_entity restatement against the reference's own counts and P/R/F1, incl. malformed
    sequences (I- without B-, type switches, specials inside entities); plus the host-side tag tables."""
    from phonlp_amd.sharding import prf, tag_tables
    z = golden("ner_entities.npz")
    vocab = [str(v) for v in z["vocab"]]
    kind, typ = tag_tables(vocab)
    assert kind.tolist()[:5] == [5, 5, 5, 5, 0] and kind.tolist()[5:9] == [1, 2, 3, 4]
    assert typ.tolist()[5:9] == [0, 0, 0, 0] and typ.tolist()[9] == 1 and typ.tolist()[4] == -1
    for c in range(int(z["n_cases"])):
        pred, gold, lens = z[f"c{c}_pred"], z[f"c{c}_gold"], z[f"c{c}_lens"]
        ps = [[vocab[t] for t in pred[b, :lens[b]]] for b in range(len(lens))]
        gs = [[vocab[t] for t in gold[b, :lens[b]]] for b in range(len(lens))]
        counts = O.ner_entity_counts(ps, gs)
        assert list(counts) == z[f"c{c}_counts"].tolist()
        np.testing.assert_allclose(prf(*counts), z[f"c{c}_prf"], rtol=0, atol=1e-15)
        types_ = [str(t) for t in z[f"c{c}_types"]]
        ents = [[i, st, en, types_.index(ty)] for i, p in enumerate(ps) for st, en, ty in O.decode_from_bioes(p)]
        assert ents == z[f"c{c}_pred_ents"].tolist()


def test_dep_loss(golden):
    """Training-side parser loss (ref models/jointmodel/model.py:209-236, dropout off) against the loss the reference's
    dep_forward(eval=False) returned and the three sums its own crit_dep produced on the way."""
    z = golden("dep_loss_small.npz")
    p = _params(z)
    for c in range(int(z["n_cases"])):
        lens = z[f"c{c}_lens"].tolist()
        dep = O.dep_heads(torch.from_numpy(z[f"c{c}_enc"]), lens, p)
        total, terms = O.dep_loss(dep, lens, z[f"c{c}_head"], z[f"c{c}_rel"], int(z[f"c{c}_nwords"]))
        np.testing.assert_allclose([float(t) for t in terms[:3]], z[f"c{c}_ce"], rtol=2e-5)
        assert abs(float(total) - float(z[f"c{c}_loss"])) <= 2e-5 * abs(float(z[f"c{c}_loss"]))
        tl = z[f"c{c}_tag_lens"].tolist()
        logits, _ = O.pos_head(torch.from_numpy(z[f"c{c}_tag_enc"]), tl, p, "upos_emb_matrix_ner")
        got = float(O.pos_loss(logits, tl, z[f"c{c}_upos"]))
        assert abs(got - float(z[f"c{c}_pos_loss"])) <= 2e-5 * abs(float(z[f"c{c}_pos_loss"]))


def test_dep_loss_gradients(golden):
    """The closed-form backward of the parser loss (oracle.dep_loss_grad) against the gradients the reference's own
    graph produced at the outputs of its four scorers (dep_loss_grad_small.npz), and against autograd through the
    oracle's dep_loss."""
    z = golden("dep_loss_grad_small.npz")
    p = _params(z)
    for c in range(int(z["n_cases"])):
        lens = z[f"c{c}_lens"].tolist()
        nw = int(z[f"c{c}_nwords"])
        dep = O.dep_heads(torch.from_numpy(z[f"c{c}_enc"]), lens, p)
        total, _ = O.dep_loss(dep, lens, z[f"c{c}_head"], z[f"c{c}_rel"], nw)
        assert abs(float(total) - float(z[f"c{c}_loss"])) <= 2e-5 * abs(float(z[f"c{c}_loss"]))
        g = O.dep_loss_grad(dep, lens, z[f"c{c}_head"], z[f"c{c}_rel"], nw)
        tt = lambda k: torch.from_numpy(z[f"c{c}_{k}"])
        torch.testing.assert_close(g["d_unlabeled"], tt("d_unlabeled"), rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(g["d_label"][:, 1:], tt("d_label"), rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(g["d_lin"][:, 1:], tt("d_lin"), rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(g["d_dist"][:, 1:], tt("d_dist"), rtol=1e-4, atol=1e-7)
        assert not g["d_label"][:, 0].any() and not g["d_lin"][:, 0].any() and not g["d_dist"][:, 0].any()
        # autograd through the oracle's own loss says the same
        leaves = {k: dep[k].clone().requires_grad_(True) for k in ("unlabeled_raw", "lin_raw", "dist_raw", "label_scores")}
        t2, _ = O.dep_loss(dict(dep, **leaves), lens, z[f"c{c}_head"], z[f"c{c}_rel"], nw)
        t2.backward()
        torch.testing.assert_close(leaves["unlabeled_raw"].grad, g["d_unlabeled"], rtol=1e-4, atol=1e-7)
        head = torch.zeros(len(lens), max(lens), dtype=torch.long)
        head[:, 1:] = tt("head")
        at = lambda t: torch.gather(t, 2, head.unsqueeze(2)).squeeze(2)
        torch.testing.assert_close(at(leaves["lin_raw"].grad)[:, 1:], g["d_lin"][:, 1:], rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(at(leaves["dist_raw"].grad)[:, 1:], g["d_dist"][:, 1:], rtol=1e-4, atol=1e-7)
        B, L, _, O_ = dep["label_scores"].shape
        lab_h = torch.gather(leaves["label_scores"].grad, 2, head.view(B, L, 1, 1).expand(-1, -1, -1, O_)).squeeze(2)
        torch.testing.assert_close(lab_h[:, 1:], g["d_label"][:, 1:], rtol=1e-4, atol=1e-7)


def test_crf_nll_gradients(golden):
    """Autograd through the oracle's crf_nll against the gradients the reference's own CRFLoss produced."""
    z = golden("crf.npz")
    for c in range(int(z["n"])):
        x, m, t, tr = (torch.from_numpy(z[f"c{c}_{k}"]) for k in ("inputs", "masks", "tags", "trans"))
        x, tr = x.clone().requires_grad_(True), tr.clone().requires_grad_(True)
        O.crf_nll(x, m, t, tr).backward()
        torch.testing.assert_close(x.grad, torch.from_numpy(z[f"c{c}_d_inputs"]), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(tr.grad, torch.from_numpy(z[f"c{c}_d_trans"]), rtol=1e-4, atol=1e-6)
