"""Ragged batches with their live words packed (include/phonlp_heads.h, "packed rows"): the reference runs its dense layers
on PackedSequence data (annotate_model.py:114-130, 155-187); the packed path here must give what the padded path gives.

Bar: the decoded outputs (pos, ner, heads, rels, mst_status) identical on every batch; the arc log-probabilities of the
live dependents identical bit for bit (a row's arithmetic does not depend on where the row sits), and the whole path
against the CPU oracle on a sample of sentences."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _heads(dev, dims, pack):
    from phonlp_amd import HeadWeights, TaskHeads
    from phonlp_amd import synthetic as S
    h = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(*dims, seed=5), dev))
    h.cfg["pack_pct"] = 90 if pack else 0
    h.cfg["fused_max_len"] = 256  # the fused (and packed) label path also for the L = 200 case (default: up to 128)
    return h


def test_pack_descriptor_and_rows_roundtrip():
    from phonlp_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    for B, L, W in ((1, 7, 4), (5, 9, 12), (300, 33, 8), (1500, 5, 4)):
        lens = torch.randint(0, L + 1, (B,), generator=g).tolist()
        lens[0] = max(lens[0], 1)
        m = sum(lens)
        off, rowmap = ops.pack_descriptor(torch.tensor(lens, dtype=torch.int32, device=dev), L, m)
        want_off = np.concatenate([[0], np.cumsum(lens)])
        want_map = np.concatenate([b * L + np.arange(l) for b, l in enumerate(lens)])
        assert off.cpu().numpy().tolist() == want_off.tolist()
        assert rowmap.cpu().numpy().tolist() == want_map.tolist()
        x = torch.randn(B * L, W, generator=g).to(dev)
        xp = ops.pack_rows(x, rowmap)
        assert torch.equal(xp.cpu(), x.cpu()[want_map])
        back = ops.unpack_rows(xp, rowmap, B * L)
        want = torch.zeros(B * L, W)
        want[want_map] = x.cpu()[want_map]
        assert torch.equal(back.cpu(), want)
        ids = torch.arange(1, m + 1, dtype=torch.int32, device=dev)
        bi = ops.unpack_rows(ids, rowmap, B * L).cpu()
        wi = torch.zeros(B * L, dtype=torch.int32)
        wi[want_map] = torch.arange(1, m + 1, dtype=torch.int32)
        assert torch.equal(bi, wi)


@pytest.mark.parametrize("dims,B,L,scorers", [
    ((128, 20, 48, 11, 13, 9), 24, 40, 3),
    ((128, 20, 48, 11, 13, 9), 7, 97, 1),
    ((768, 100, 400, 24, 33, 9), 64, 128, 3),
    ((96, 12, 40, 9, 7, 5), 5, 200, 3),
])
def test_packed_path_matches_padded_path(dims, B, L, scorers):
    from phonlp_amd import ops
    from phonlp_amd import synthetic as S
    dev = torch.device("cuda:0")
    ops.set_gemm_engine(2)
    hp, hd = _heads(dev, dims, True), _heads(dev, dims, False)
    if scorers == 1:  # checkpoints without the linearization / distance scorers
        from phonlp_amd import HeadWeights, TaskHeads
        sd = S.make_state_dict(*dims, seed=5)
        for h in (hp, hd):
            cfg = h.cfg
            h.__init__(HeadWeights.from_state_dict(sd, dev, linearization=False, distance=False))
            h.cfg = cfg
    enc_dep, enc_tag = S.make_encoder_outputs(B, L, dims[0], seed=11)
    lens1 = S.make_sentlens(B, L, ragged=True, seed=B + L)
    lens1[-1] = 2  # the shortest sentence a parser sees: ROOT + one word
    lens = [l - 1 for l in lens1]
    ed, et = enc_dep.to(dev), enc_tag.to(dev)
    assert hp._pack_plan(ed, lens1, et, lens) == (sum(lens1), sum(lens))
    assert hd._pack_plan(ed, lens1, et, lens) is None
    a = hp.annotate_batch(ed, lens1, et, lens)
    b = hd.annotate_batch(ed, lens1, et, lens)
    torch.cuda.synchronize()
    assert int((a["mst_status"] != 0).sum()) == 0 and int((b["mst_status"] != 0).sum()) == 0
    for k in ("pos", "ner", "heads", "rels", "mst_status"):
        assert torch.equal(a[k], b[k]), k
    # device lengths (nothing the host can size launches with): the padded path, same results
    c = hp.annotate_batch(ed, torch.tensor(lens1, dtype=torch.int32, device=dev), et,
                          torch.tensor(lens, dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    for k in ("pos", "ner", "heads", "rels"):
        assert torch.equal(a[k], c[k]), k


def test_packed_arc_scores_bitwise_and_untouched_padding():
    """Live dependents: the packed arc kernel's log-probs equal the padded kernel's bit for bit (padded head columns take
    pad_row); padded dependents' rows are left as they were."""
    from phonlp_amd import ops
    from phonlp_amd import synthetic as S
    from phonlp_amd.heads import as_sentlens
    dev = torch.device("cuda:0")
    ops.set_gemm_engine(2)
    dims = (128, 20, 48, 11, 13, 9)
    B, L = 9, 70
    h = _heads(dev, dims, True)
    w = h.w
    enc_dep, _ = S.make_encoder_outputs(B, L, dims[0], seed=2)
    lens1 = S.make_sentlens(B, L, ragged=True, seed=77)
    lens_d = as_sentlens(lens1, dev)
    ed = enc_dep.to(dev)
    D = dims[0] + w.emb_eye.shape[1]
    featp, feati, _, _, _ = ops.pos_head_split(ed, lens_d, w.upos_hid_w, w.upos_hid_b, w.upos_clf_w, w.upos_clf_b,
                                               w.emb_eye, False)
    XFa = ops.linear_split(featp, feati, w.arc_W12p_s[0], w.arc_W12p_s[1], w.arc_b12f, D, True, lens_d, L, w.arc_b12, 3)
    dense = ops.biaffine_arc_scores(XFa, B, L, w.arc_Wbt, w.H, w.n_arc_scorers, w.linearization, w.distance)
    off, rowmap = ops.pack_descriptor(lens_d, L, sum(lens1))
    XFp = ops.pack_rows(XFa, rowmap)
    packed = ops.biaffine_arc_scores_packed(XFp, off, lens_d, h._arc_pad_row, B, L, w.arc_Wbt, w.H, w.n_arc_scorers,
                                            w.linearization, w.distance)
    torch.cuda.synchronize()
    # the padded rows of XFa ARE pad_row
    pad_rows = XFa.view(B, L, -1)[1, lens1[1]:]
    assert pad_rows.numel() and torch.equal(pad_rows, h._arc_pad_row.expand_as(pad_rows))
    for bb in range(B):
        assert torch.equal(packed[bb, :lens1[bb]], dense[bb, :lens1[bb]]), bb


def test_packed_path_against_oracle():
    """The packed path end to end against the CPU oracle chain on a ragged batch: every decoded id."""
    from oracle import c_oracle as C
    from oracle import heads_oracle as O
    from phonlp_amd import ops
    from phonlp_amd import synthetic as S
    dev = torch.device("cuda:0")
    ops.set_gemm_engine(2)
    dims = (128, 20, 48, 11, 13, 9)
    B, L = 12, 50
    h = _heads(dev, dims, True)
    sd = S.make_state_dict(*dims, seed=5)
    enc_dep, enc_tag = S.make_encoder_outputs(B, L, dims[0], seed=4)
    lens1 = S.make_sentlens(B, L, ragged=True, seed=9)
    lens = [l - 1 for l in lens1]
    out = h.annotate_batch(enc_dep.to(dev), lens1, enc_tag.to(dev), lens)
    torch.cuda.synchronize()
    od = O.dep_heads(enc_dep, lens1, sd, keep_label_scores=True)
    ot = O.tagger_heads(enc_tag, lens, sd)
    want_h, want_st = C.mst_batch(od["arc_logp"].numpy(), lens1)
    want_t, _ = C.viterbi_batch(ot["ner_logits"].numpy(), sd["crit_ner._transitions"].numpy(), lens)
    assert not want_st.any()
    hg, rg = out["heads"].cpu().numpy(), out["rels"].cpu().numpy()
    ng, pg = out["ner"].cpu().numpy(), out["pos"].cpu().numpy()
    lab = od["label_scores"]  # [b, i, j, o]
    for b in range(B):
        assert hg[b, :lens1[b]].tolist() == want_h[b, :lens1[b]].tolist(), b
        assert not hg[b, lens1[b]:].any() and not rg[b, lens1[b]:].any() and rg[b, 0] == 0
        for i in range(1, lens1[b]):
            assert rg[b, i] == int(lab[b, i, hg[b, i]].argmax()), (b, i)
        assert ng[b, :lens[b]].tolist() == want_t[b, :lens[b]].tolist(), b
        assert pg[b, :lens[b]].tolist() == ot["pos_pred"][b, :lens[b]].tolist(), b
        assert not pg[b, lens[b]:].any()
