"""The custom ops' shape functions (`register_fake`, phonlp_amd/ops.py): every op of the head path must trace under
FakeTensorMode (what torch.compile / torch.export do) without running a kernel, and what the shape function promises
must be what the op returns.  CPU part: a head chain on fake ROCm tensors (no GPU needed); GPU part: fake against
real metadata, op by op, at the reference's head sizes."""
import pytest
import torch
from torch._subclasses.fake_tensor import FakeTensorMode


def _meta(out):
    outs = out if isinstance(out, (tuple, list)) else (out,)
    return [(tuple(o.shape), o.dtype, o.device.type) for o in outs if o is not None]


def test_head_chain_traces_on_fake_tensors():
    """No GPU: shapes flow through the ops of one annotate step on fake device tensors."""
    from phonlp_amd import ops
    B, L, E, Hh, U, TE, T, H, O = 3, 10, 32, 16, 5, 8, 7, 12, 6
    KP = ops.kp(H)
    with FakeTensorMode():
        f = lambda *s: torch.empty(*s, device="cuda")
        i32 = lambda *s: torch.empty(*s, dtype=torch.int32, device="cuda")
        enc, lens = f(B, L, E), i32(B)
        feat, pos, logits = ops.pos_head(enc, lens, f(Hh, E), f(Hh), f(U, Hh), f(U), f(U, TE))
        assert feat.shape == (B, L, E + TE) and pos.dtype == torch.int32 and logits.shape == (B, L, U)
        pos2, ner, _ = ops.tagger_head(enc, lens, f(Hh + T, E), f(Hh + T), f(U, Hh), f(U), f(U, TE), f(T, TE), f(T))
        assert ner.shape == (B, L, T) and pos2.shape == (B, L)
        tags, best = ops.crf_viterbi(ner, f(T, T), lens)
        assert tags.shape == (B, L) and tags.dtype == torch.int32 and best.shape == (B,)
        W12, b12, Wbt = ops.pack_scorer(f(H, E + TE), f(H), f(H, E + TE), f(H), f(H + 1, H + 1, O))
        assert W12.shape == (2 * KP, E + TE) and Wbt.shape == (O * KP, KP)
        hid = ops.dep_hidden(feat, lens, f(E + TE, E + TE), f(E + TE))
        arc, raw = ops.deep_biaffine_arc(hid, W12, b12, f(KP, KP), H, 1, False, False, False)
        assert arc.shape == (B, L, L) and raw.numel() == 0
        heads, status = ops.mst_one_root(arc, lens)
        assert heads.shape == (B, L) and status.shape == (B,) and heads.dtype == torch.int32
        XF, Tm = ops.label_project(hid, W12, b12, Wbt, H, O)
        assert XF.shape == (B * L, 2 * KP) and Tm.shape == (B * L, O * KP)
        rels, score = ops.label_at_heads(Tm, XF, heads, lens, H, O)
        assert rels.shape == (B, L) and rels.dtype == torch.int32 and score.dtype == torch.float32
        xp, xi = ops.split_rows(XF[:, :KP])
        wp, wi = ops.split_rows(Wbt)
        assert xp.dtype == torch.uint8 and xp.shape[1] == wp.shape[1] and xi.shape == (B * L,)
        y = ops.linear_split(xp, xi, wp, wi, None, KP, False, None, 0, None, 1)
        assert y.shape == (B * L, O * KP)
        counts = ops.eval_counts(pos, tags, heads, rels, pos, tags, heads, rels, lens, lens, status)
        assert counts.shape == (8,) and counts.dtype == torch.int64


@pytest.mark.gpu
def test_fake_metadata_matches_real_outputs():
    from phonlp_amd import HeadWeights, ops, synthetic as S
    dev = torch.device("cuda:0")
    w = HeadWeights.from_state_dict(S.make_state_dict(), dev)
    B, L = 3, 12
    g = torch.Generator().manual_seed(5)
    enc = torch.randn(B, L, w.upos_hid_w.shape[1], generator=g).to(dev)
    lens = torch.tensor([12, 7, 3], dtype=torch.int32, device=dev)
    H, O, KP = w.H, w.O, ops.kp(w.H)
    T = w.ner_w.shape[0]

    checked = []

    def both(op, *args):
        real = op(*args)
        with FakeTensorMode(allow_non_fake_inputs=False) as mode:
            fargs = [mode.from_tensor(a) if isinstance(a, torch.Tensor) else a for a in args]
            fake = op(*fargs)
        assert _meta(fake) == _meta(real), op
        checked.append(op)
        return real

    feat, pos, logits = both(ops.pos_head, enc, lens, w.upos_hid_w, w.upos_hid_b, w.upos_clf_w, w.upos_clf_b, w.emb_dep)
    both(ops.pos_head_split, enc, lens, w.upos_hid_w, w.upos_hid_b, w.upos_clf_w, w.upos_clf_b, w.emb_dep, True)
    planes, inv, *_ = both(ops.pos_head_split, enc, lens, w.upos_hid_w, w.upos_hid_b, w.upos_clf_w, w.upos_clf_b,
                           w.emb_dep, False)
    _, ner, _ = both(ops.tagger_head, enc, lens, w.tag_Whn, w.tag_bhn, w.upos_clf_w, w.upos_clf_b, w.emb_ner,
                     w.ner_w_emb, w.ner_b)
    feat_n, _, _ = ops.pos_head(enc, lens, w.upos_hid_w, w.upos_hid_b, w.upos_clf_w, w.upos_clf_b, w.emb_ner)
    both(ops.ner_emission, feat_n, lens, w.ner_w, w.ner_b)
    tags, _ = both(ops.crf_viterbi, ner, w.trans, lens)
    both(ops.crf_log_partition, ner, w.trans, None, lens, tags)
    hid = both(ops.dep_hidden, feat, lens, w.dep_hid_w, w.dep_hid_b)
    both(ops.linear, hid.view(B * L, -1), w.lab_W12, w.lab_b12, True, lens, L)
    arc, _ = both(ops.deep_biaffine_arc, hid, w.arc_W12, w.arc_b12, w.arc_Wbt, H, w.n_arc_scorers, True, True, True)
    both(ops.deep_biaffine_arc, hid, w.arc_W12, w.arc_b12, w.arc_Wbt, H, w.n_arc_scorers, True, True, False)
    lab, _ = both(ops.deep_biaffine_label, hid, w.lab_W12, w.lab_b12, w.lab_Wbt, H, O, True)
    both(ops.deep_biaffine_label, hid, w.lab_W12, w.lab_b12, w.lab_Wbt, H, O, False)
    heads, status = both(ops.mst_one_root, arc, lens)
    XF, Tm = both(ops.label_project, hid, w.lab_W12, w.lab_b12, w.lab_Wbt, H, O)
    rels, _ = both(ops.label_at_heads, Tm, XF, heads, lens, H, O)
    both(ops.gather_labels, lab, heads, lens)
    x1p, x1i = both(ops.split_rows, XF[:, :KP])
    both(ops.linear_split, x1p, x1i, w.lab_Wbt_s[0], w.lab_Wbt_s[1], None, KP, False, None, 0, None, 1)
    XFa = ops.linear(hid.view(B * L, -1), w.arc_W12, w.arc_b12, True, lens, L)
    both(ops.biaffine_arc_scores, XFa, B, L, w.arc_Wbt, H, w.n_arc_scorers, True, True)
    T4 = both(ops.label_stage_a_split, x1p, x1i, w.lab_Wbt_s[0], w.lab_Wbt_s[1], B, L, H, O, w.lab_ones_col)
    both(ops.label_at_heads_rows4, T4, XF, heads, lens, H, O)
    both(ops.label_at_heads_fused, x1p, x1i, w.lab_Wbtc_s[0], w.lab_Wbtc_s[1], XF, heads, lens, H, O)
    sub = torch.arange(L, device=dev).repeat(B, 1)
    both(ops.gather_first_subword, enc, sub, lens)
    both(ops.eval_counts, pos, tags, heads, rels, pos, tags, heads, rels, lens, lens, status)
    both(ops.pack_scorer, torch.randn(H, 40, device=dev), torch.randn(H, device=dev), torch.randn(H, 40, device=dev),
         torch.randn(H, device=dev), torch.randn(H + 1, H + 1, 3, device=dev))
    torch.cuda.synchronize()
    assert len(checked) >= 25


@pytest.mark.gpu
def test_ops_run_under_torch_compile_tracing():
    """dynamo + AOT tracing (backend aot_eager: no code generation) of a function made of the custom ops gives the
    eager results -- the ops are opaque nodes whose shapes come from the functions above."""
    from phonlp_amd import HeadWeights, ops, synthetic as S
    dev = torch.device("cuda:0")
    w = HeadWeights.from_state_dict(S.make_state_dict(), dev)
    g = torch.Generator().manual_seed(6)
    enc = torch.randn(2, 9, w.upos_hid_w.shape[1], generator=g).to(dev)
    lens = torch.tensor([9, 4], dtype=torch.int32, device=dev)

    def step(enc, lens):
        feat, pos, _ = ops.pos_head(enc, lens, w.upos_hid_w, w.upos_hid_b, w.upos_clf_w, w.upos_clf_b, w.emb_dep)
        hid = ops.dep_hidden(feat, lens, w.dep_hid_w, w.dep_hid_b)
        arc, _ = ops.deep_biaffine_arc(hid, w.arc_W12, w.arc_b12, w.arc_Wbt, w.H, w.n_arc_scorers, True, True, False)
        heads, status = ops.mst_one_root(arc, lens)
        XF, Tm = ops.label_project(hid, w.lab_W12, w.lab_b12, w.lab_Wbt, w.H, w.O)
        rels, _ = ops.label_at_heads(Tm, XF, heads, lens, w.H, w.O)
        return pos, heads, rels, status

    want = step(enc, lens)
    got = torch.compile(step, backend="aot_eager", fullgraph=True)(enc, lens)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
