#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by RUNNING THE REFERENCE'S OWN CODE.

Runs only in the authoring container (needs /root/reference); the GPU box only sees the
committed .npz files.  The reference modules are imported from where they lie (by file path
for the three dependency-free files, as a package -- with an empty stand-in for the missing
``gdown`` download helper, which the hot path never calls -- for ``annotate_model``).
Nothing from the reference is copied into this repository: the fixtures hold only inputs,
weights drawn here, and the outputs the reference produced for them.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF_ROOT = "/root/reference"
REF = os.path.join(REF_ROOT, "phonlp")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.setrecursionlimit(20000)


def load_by_path(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


bia = load_by_path("models/common/biaffine.py", "ref_biaffine")
crf = load_by_path("models/common/crf.py", "ref_crf")
cle = load_by_path("models/common/chuliu_edmonds.py", "ref_cle")


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays")


# ---------------------------------------------------------------------------- F1 biaffine
def gen_biaffine():
    g = torch.Generator().manual_seed(101)
    arrs = {}
    cases = [(2, 5, 7, 6, 5, 1), (2, 5, 7, 6, 5, 5), (3, 9, 9, 8, 8, 33), (1, 4, 4, 3, 3, 2)]
    for c, (B, L1, L2, D1, D2, O) in enumerate(cases):
        m = bia.PairwiseBilinear(D1, D2, O)
        with torch.no_grad():
            m.weight.copy_(torch.randn(D1, D2, O, generator=g))
            m.bias.copy_(torch.randn(O, generator=g) + 3.0)  # non-zero: proves it is ignored
        x1 = torch.randn(B, L1, D1, generator=g)
        x2 = torch.randn(B, L2, D2, generator=g)
        with torch.no_grad():
            out = m(x1, x2)
        arrs[f"pb{c}_x1"] = x1.numpy()
        arrs[f"pb{c}_x2"] = x2.numpy()
        arrs[f"pb{c}_weight"] = m.weight.detach().numpy()
        arrs[f"pb{c}_bias"] = m.bias.detach().numpy()
        arrs[f"pb{c}_out"] = out.contiguous().numpy()
    # full DeepBiaffineScorer (W1/W2 + relu + ones column)
    for c, (B, L, D, H, O) in enumerate([(2, 6, 10, 7, 1), (2, 6, 10, 7, 4)]):
        m = bia.DeepBiaffineScorer(D, D, H, O, pairwise=True, dropout=0.5).eval()
        with torch.no_grad():
            m.scorer.W_bilin.weight.copy_(torch.randn(H + 1, H + 1, O, generator=g) * 0.3)
            m.scorer.W_bilin.bias.copy_(torch.randn(O, generator=g))
        h = torch.randn(B, L, D, generator=g)
        with torch.no_grad():
            out = m(h, h)
        arrs[f"db{c}_h"] = h.numpy()
        for k, v in m.state_dict().items():
            arrs[f"db{c}_p_{k}"] = v.numpy()
        arrs[f"db{c}_out"] = out.contiguous().numpy()
    arrs["n_pb"] = np.int64(len(cases))
    arrs["n_db"] = np.int64(2)
    save("biaffine.npz", **arrs)


# ---------------------------------------------------------------------------- f4: backward through a scorer
def gen_biaffine_backward():
    """The reference's own autograd through DeepBiaffineScorer (O = 1: the unlabeled / linearization / distance scorers,
    biaffine.py:69-85) under a random upstream gradient G: loss = sum(G * scores).  Records the inputs and the gradients
    of the input and of every parameter the forward touches (W_bilin.bias is never added: no gradient)."""
    g = torch.Generator().manual_seed(4242)
    arrs = {}
    cases = [(3, 10, 20, 12), (2, 7, 24, 16), (4, 33, 36, 24)]  # L not a multiple of 4: the padded reductions
    for c, (B, L, D, H) in enumerate(cases):
        m = bia.DeepBiaffineScorer(D, D, H, 1, pairwise=True, dropout=0.5).eval()
        with torch.no_grad():
            m.scorer.W_bilin.weight.copy_(torch.randn(H + 1, H + 1, 1, generator=g) * 0.3)
        h = torch.randn(B, L, D, generator=g).requires_grad_(True)
        G = torch.randn(B, L, L, generator=g)
        out = m(h, h)  # [B, L, L, 1]
        (out.squeeze(3) * G).sum().backward()
        arrs[f"c{c}_h"] = h.detach().numpy()
        arrs[f"c{c}_G"] = G.numpy()
        arrs[f"c{c}_out"] = out.detach().squeeze(3).contiguous().numpy()
        for k, v in m.state_dict().items():
            arrs[f"c{c}_p_{k}"] = v.numpy()
        arrs[f"c{c}_d_h"] = h.grad.numpy()
        for k, v in m.named_parameters():
            if v.grad is not None:
                arrs[f"c{c}_d_{k}"] = v.grad.numpy()
        assert m.scorer.W_bilin.bias.grad is None
    arrs["n_cases"] = np.int64(len(cases))
    # the label scorer: O relations, the loss reads the scores at the gold heads only (jointmodel/model.py:218-223);
    # g = 0 on ROOT and padding rows, as the parser loss leaves them
    lcases = [(3, 10, 20, 12, 5), (2, 13, 24, 16, 7)]
    for c, (B, L, D, H, O) in enumerate(lcases):
        m = bia.DeepBiaffineScorer(D, D, H, O, pairwise=True, dropout=0.5).eval()
        with torch.no_grad():
            m.scorer.W_bilin.weight.copy_(torch.randn(H + 1, H + 1, O, generator=g) * 0.3)
        h = torch.randn(B, L, D, generator=g).requires_grad_(True)
        lens = torch.randint(3, L + 1, (B,), generator=g)
        lens[0] = L
        head = torch.stack([torch.randint(0, int(n), (L,), generator=g) for n in lens])
        gl = torch.randn(B, L, O, generator=g)
        live = (torch.arange(L)[None, :] < lens[:, None]) & (torch.arange(L)[None, :] >= 1)
        gl = gl * live[:, :, None]
        out = m(h, h)  # [B, L, L, O]
        at_head = torch.gather(out, 2, head.view(B, L, 1, 1).expand(-1, -1, -1, O)).squeeze(2)  # [B, L, O]
        (at_head * gl).sum().backward()
        arrs[f"l{c}_h"] = h.detach().numpy()
        arrs[f"l{c}_g"] = gl.numpy()
        arrs[f"l{c}_head"] = head.numpy().astype(np.int32)
        arrs[f"l{c}_lens"] = lens.numpy().astype(np.int32)
        arrs[f"l{c}_at_head"] = at_head.detach().numpy()
        for k, v in m.state_dict().items():
            arrs[f"l{c}_p_{k}"] = v.numpy()
        arrs[f"l{c}_d_h"] = h.grad.numpy()
        for k, v in m.named_parameters():
            if v.grad is not None:
                arrs[f"l{c}_d_{k}"] = v.grad.numpy()
    arrs["n_label_cases"] = np.int64(len(lcases))
    save("biaffine_backward.npz", **arrs)


# ---------------------------------------------------------------------------- F3 viterbi
def gen_viterbi():
    rng = np.random.default_rng(303)
    arrs = {}
    n = 0
    for (l, T, mode) in [
        (1, 9, "rand"), (2, 9, "rand"), (17, 9, "rand"), (63, 12, "rand"), (128, 9, "rand"),
        (31, 9, "relu_zero"), (64, 12, "relu_zero"), (40, 5, "ints"), (20, 9, "allzero"),
        (50, 33, "rand"), (35, 64, "relu_rand"), (128, 9, "relu_rand"),
    ]:
        if mode == "rand":
            s = rng.standard_normal((l, T)).astype(np.float32)
            tr = (rng.standard_normal((T, T)) * 0.5).astype(np.float32)
        elif mode == "relu_zero":  # ReLU'd emissions, zero-init transitions: heavy ties
            s = np.maximum(rng.standard_normal((l, T)), 0).astype(np.float32)
            tr = np.zeros((T, T), np.float32)
        elif mode == "relu_rand":
            s = np.maximum(rng.standard_normal((l, T)), 0).astype(np.float32)
            tr = (rng.standard_normal((T, T)) * 0.1).astype(np.float32)
        elif mode == "ints":
            s = rng.integers(-2, 3, (l, T)).astype(np.float32)
            tr = rng.integers(-1, 2, (T, T)).astype(np.float32)
        else:
            s = np.zeros((l, T), np.float32)
            tr = np.zeros((T, T), np.float32)
        tags, best = crf.viterbi_decode(s, tr)
        arrs[f"v{n}_scores"] = s
        arrs[f"v{n}_trans"] = tr
        arrs[f"v{n}_tags"] = np.asarray(tags, dtype=np.int64)
        arrs[f"v{n}_best"] = np.float32(best)
        n += 1
    arrs["n"] = np.int64(n)
    save("viterbi.npz", **arrs)


# ---------------------------------------------------------------------------- F4 CRF loss
def gen_crf():
    g = torch.Generator().manual_seed(404)
    arrs = {}
    n = 0
    for (B, L, T, relu) in [(4, 11, 9, True), (3, 20, 12, False), (1, 1, 5, False), (6, 33, 9, True)]:
        m = crf.CRFLoss(T)
        with torch.no_grad():
            m._transitions.copy_(torch.randn(T, T, generator=g) * 0.3)
        x = torch.randn(B, L, T, generator=g)
        if relu:
            x = torch.relu(x)
        lens = sorted([int(v) for v in torch.randint(1, L + 1, (B,), generator=g)], reverse=True)
        lens[0] = L
        masks = torch.ones(B, L, dtype=torch.bool)
        for b, ln in enumerate(lens):
            masks[b, :ln] = False
            x[b, ln:] = 0  # pad rows are zeros in the real pipeline
        tags = torch.randint(0, T, (B, L), generator=g)
        with torch.no_grad():
            m.bs, m.sl, m.nc = x.size()
            log_norm = m.crf_log_norm(x, masks, tags)
            loss, _ = m(x, masks, tags)
        arrs[f"c{n}_inputs"] = x.numpy()
        arrs[f"c{n}_masks"] = masks.numpy()
        arrs[f"c{n}_tags"] = tags.numpy()
        arrs[f"c{n}_trans"] = m._transitions.detach().numpy()
        arrs[f"c{n}_log_norm"] = log_norm.numpy()
        arrs[f"c{n}_loss"] = np.float32(loss.item())
        # the gradients the reference's own autograd derives for the emissions and the transitions
        xg = x.clone().requires_grad_(True)
        m._transitions.grad = None
        lg, _ = m(xg, masks, tags)
        lg.backward()
        arrs[f"c{n}_d_inputs"] = xg.grad.numpy()
        arrs[f"c{n}_d_trans"] = m._transitions.grad.numpy().copy()
        n += 1
    arrs["n"] = np.int64(n)
    save("crf.npz", **arrs)


# ---------------------------------------------------------------------------- F5 MST
def log_softmax_rows(a):
    a = a - a.max(axis=1, keepdims=True)
    return (a - np.log(np.exp(a).sum(axis=1, keepdims=True))).astype(np.float32)


def gen_mst():
    rng = np.random.default_rng(505)
    mats, outs, kinds = [], [], []

    def add(m, kind):
        m = np.ascontiguousarray(m, dtype=np.float32)
        t = cle.chuliu_edmonds_one_root(m.copy())
        mats.append(m)
        outs.append(np.asarray(t, dtype=np.int64))
        kinds.append(kind)

    for l in [2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 15, 17, 20, 24, 31, 32, 33, 40, 48, 64]:
        for rep in range(6):
            add(log_softmax_rows(rng.standard_normal((l, l)) * 2), "randn")
    for l in [3, 5, 8, 13, 21, 30]:
        for rep in range(4):
            add(rng.integers(-3, 1, (l, l)).astype(np.float32), "ints")  # massive ties
        add(np.zeros((l, l), np.float32), "zeros")
        add(log_softmax_rows(np.zeros((l, l))), "uniform")
    # distance-prior-like (what a zero-init model produces): heads from |i-j| only
    for l in [6, 11, 25, 50]:
        i = np.arange(l)[:, None]
        j = np.arange(l)[None, :]
        add(log_softmax_rows(-np.log((np.abs(i - j) - 1.69) ** 2 / 2 + 1.0)), "distprior")
    # many-root cases: column 0 strongly preferred
    for l in [5, 9, 16, 28, 45]:
        for rep in range(3):
            m = rng.standard_normal((l, l)) * 1.5
            m[:, 0] += 3.0
            add(log_softmax_rows(m), "rooty")
    # cycle-heavy: neighbours prefer each other
    for l in [6, 10, 18, 36, 60]:
        for rep in range(3):
            m = rng.standard_normal((l, l))
            for k in range(1, l - 1, 2):
                m[k, k + 1] += 4
                m[k + 1, k] += 4
            add(log_softmax_rows(m), "pairs")
    for l in [96, 128]:
        for rep in range(2):
            add(log_softmax_rows(rng.standard_normal((l, l)) * 2), "randn_big")
    # -inf columns (as produced by masking)
    for l in [5, 12]:
        m = rng.standard_normal((l, l)).astype(np.float32)
        m[:, l - 1] = -np.inf
        add(m, "neginf_col")
    sizes = np.asarray([m.shape[0] for m in mats], dtype=np.int64)
    save(
        "mst.npz",
        sizes=sizes,
        flat=np.concatenate([m.ravel() for m in mats]),
        heads=np.concatenate(outs),
        kinds=np.asarray(kinds),
    )
    # statistics on how many cases exercise contraction / the one-root retry
    ncyc = nmulti = 0
    for m in mats:
        s = m.astype(np.float64)
        t = cle.chuliu_edmonds(s.copy())
        s2 = s.copy()
        np.fill_diagonal(s2, -np.inf)
        s2[0] = -np.inf
        s2[0, 0] = 0
        if len(cle.tarjan(s2.argmax(axis=1))):
            ncyc += 1
        if (t[1:] == 0).sum() != 1:
            nmulti += 1
    print(f"mst fixture: {len(mats)} cases, {ncyc} with >=1 contraction, {nmulti} needing the one-root retry")


# ---------------------------------------------------------------------------- F2/F6 JointModel
class ToyVocab:
    """Minimal stand-in for the reference's vocab objects (id<->string maps only)."""

    def __init__(self, units):
        self.units = ["<PAD>", "<UNK>", "<EMPTY>", "<ROOT>"] + list(units)
        self.index = {u: i for i, u in enumerate(self.units)}

    def __len__(self):
        return len(self.units)

    def map(self, units):
        return [self.index.get(u, 1) for u in units]

    def unmap(self, ids):
        return [self.units[int(i)] for i in ids]


class ToyTokenizer:
    """Deterministic stand-in tokenizer: word -> [0, 3 + hash, (optional 2nd piece), 2]."""

    def __init__(self, vocab_size):
        self.vocab_size = vocab_size

    def encode(self, w):
        h = sum((k + 1) * ord(ch) for k, ch in enumerate(w))
        pieces = [3 + h % (self.vocab_size - 3)]
        if len(w) > 6:
            pieces.append(3 + (h * 7 + 1) % (self.vocab_size - 3))
        return [0] + pieces + [2]


def build_joint_model(seed, hidden=32, tag_emb=8, biaff=16, n_upos=6, n_rel=7, n_ner=5, zero_init=False):
    sys.modules.setdefault("gdown", types.ModuleType("gdown"))  # download helper; never called
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    from transformers import RobertaConfig
    from phonlp.annotate_model import JointModel

    torch.manual_seed(seed)
    cfg = RobertaConfig(
        vocab_size=200, hidden_size=hidden, num_hidden_layers=2, num_attention_heads=4,
        intermediate_size=64, max_position_embeddings=258, pad_token_id=1, output_hidden_states=True,
    )
    vocab = {
        "upos": ToyVocab([f"P{k}" for k in range(n_upos - 4)]),
        "deprel": ToyVocab([f"rel{k}" for k in range(n_rel - 4)]),
        "ner_tag": ToyVocab([f"N{k}" for k in range(n_ner - 4)]),
        "word": ToyVocab([]),
        "char": ToyVocab([]),
    }
    args = {"tag_emb_dim": tag_emb, "deep_biaff_hidden_dim": biaff, "dropout": 0.5, "word_dropout": 0.33,
            "linearization": True, "distance": True}
    model = JointModel(args, vocab, cfg, ToyTokenizer(200)).eval()
    if not zero_init:
        with torch.no_grad():
            for name in ["unlabeled", "deprel", "linearization", "distance"]:
                getattr(model, name).scorer.W_bilin.weight.normal_(0, 0.3)
            model.upos_clf.weight.normal_(0, 0.3)
            model.crit_ner._transitions.normal_(0, 0.3)
    return model


def head_state(model):
    return {k: v.detach().numpy() for k, v in model.state_dict().items() if not k.startswith("phobert.")}


def gen_joint(name, zero_init):
    model = build_joint_model(seed=606, zero_init=zero_init)
    with open(os.path.join(REF, "sample_data", "input.txt")) as f:
        lines = [ln.strip() for ln in f if ln.strip()]
    sents = [ln.split(" ") for ln in lines[:12]] + [["a"], ["b", "c"]]
    arrs = {f"p_{k}": v for k, v in head_state(model).items()}
    arrs["n_sent"] = np.int64(len(sents))
    arrs["sent_text"] = np.asarray([" ".join(s) for s in sents])

    # --- capture the hot-path inputs/outputs of dep_forward / tagger_forward on one batch
    captured = {}
    orig_phobert_forward = model.phobert.forward

    def spy(*a, **kw):
        out = orig_phobert_forward(*a, **kw)
        captured.setdefault("enc_calls", []).append(out[2][-1].detach().clone())
        return out

    model.phobert.forward = spy
    data_t = model.chunk_batches(model.process_data_tagger(batch_text=sents), len(sents))
    data_p = model.chunk_batches(model.process_data_parser(batch_text=sents), len(sents))
    tok, fsw, wmask, _, oidx, lens = model.get_batch(0, data_t)
    tok1, fsw1, wmask1, _, oidx1, lens1 = model.get_batch(0, data_p)
    with torch.no_grad():
        preds_dep = model.dep_forward(tok1, fsw1, lens1)
        enc_dep = captured["enc_calls"][-1]
        preds_pos, logits = model.tagger_forward(tok, fsw, lens)
        enc_tag = captured["enc_calls"][-1]
    gather = lambda e, idx: torch.stack([e[i].index_select(0, idx[i]) for i in range(e.size(0))])
    arrs["dep_enc"] = gather(enc_dep, fsw1).numpy()  # B x L x hidden: hot-path input (ROOT first)
    arrs["dep_sentlens"] = np.asarray(lens1, dtype=np.int64)
    arrs["dep_arc_logp"] = preds_dep[0]
    arrs["dep_label_argmax"] = preds_dep[1]
    arrs["tag_enc"] = gather(enc_tag, fsw).numpy()
    arrs["tag_sentlens"] = np.asarray(lens, dtype=np.int64)
    arrs["tag_pos_pred"] = preds_pos[0].numpy()
    arrs["tag_ner_logits"] = logits.numpy()
    arrs["tag_words_mask"] = wmask.numpy()
    # decoders exactly as annotate() drives them (ref annotate_model.py:260-266, 279-286)
    heads = [cle.chuliu_edmonds_one_root(adj[:l, :l])[1:] for adj, l in zip(preds_dep[0], lens1)]
    rels = [[preds_dep[1][i][j + 1][h] for j, h in enumerate(hs)] for i, hs in enumerate(heads)]
    trans = model.crit_ner._transitions.data.numpy()
    ner = [crf.viterbi_decode(logits.numpy()[i, : lens[i]], trans)[0] for i in range(len(lens))]
    arrs["dec_heads"] = np.concatenate(heads).astype(np.int64)
    arrs["dec_rels"] = np.concatenate([np.asarray(r) for r in rels]).astype(np.int64)
    arrs["dec_ner"] = np.concatenate([np.asarray(t) for t in ner]).astype(np.int64)
    # CRF loss as predict_ner evaluates it (ref jointmodel/model.py:159)
    gold = torch.randint(0, len(model.vocab["ner_tag"]), wmask.shape, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        loss, _ = model.crit_ner(logits, wmask, gold)
    arrs["crf_gold"] = gold.numpy()
    arrs["crf_loss"] = np.float32(loss.item())

    # --- F6: end-to-end annotate() over the same sentences, batch_size 5 (several ragged batches)
    model.phobert.forward = orig_phobert_forward
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.txt"), os.path.join(td, "out.txt")
        with open(fin, "w") as f:
            f.write("\n".join(" ".join(s) for s in sents) + "\n")
        with torch.no_grad():
            model.annotate(input_file=fin, output_file=fout, batch_size=5)
        arrs["annotate_file"] = np.asarray(open(fout).read())
        with torch.no_grad():
            model.annotate(input_file=fin, output_file=fout, batch_size=5, output_type="conll")
        arrs["annotate_file_conll"] = np.asarray(open(fout).read())
    with torch.no_grad():
        data, pos, nerp, dep = model.annotate(text=" ".join(sents[0]))
    arrs["annotate_text_pos"] = np.asarray([p[0] for p in pos[0]])
    arrs["annotate_text_ner"] = np.asarray(nerp[0])
    arrs["annotate_text_dep_head"] = np.asarray([d[0] for d in dep[0]])
    arrs["annotate_text_dep_rel"] = np.asarray([d[1] for d in dep[0]])
    # encoder outputs per annotate() batch are needed to replay F6 without the encoder: store the
    # encoder weights instead (tiny), under their own prefix.
    for k, v in model.state_dict().items():
        if k.startswith("phobert."):
            arrs[f"enc_{k}"] = v.detach().numpy()
    save(name, **arrs)


# ---------------------------------------------------------------------------- F8 training-side predict_* seams
def gen_trainer():
    """JointTrainer.predict_dep / predict_pos / predict_ner of the reference (models/jointmodel/trainer.py:101-166 over
    jointmodel/model.py eval paths) on one ragged batch each: the evaluation-loop twins of annotate()."""
    import tempfile
    amodel = build_joint_model(seed=707)
    from phonlp.models.jointmodel.trainer import JointTrainer
    with tempfile.TemporaryDirectory() as td:
        amodel.phobert.save_pretrained(td)
        args = dict(amodel.args, pretrained_lm=td)
        trainer = JointTrainer(args=args, vocab=amodel.vocab, config_phobert=amodel.phobert.config, use_cuda=False)
    missing, unexpected = trainer.model.load_state_dict(amodel.state_dict(), strict=False)
    assert not unexpected and all(m.startswith("drop_replacement") for m in missing), (missing, unexpected)
    trainer.model.eval()
    with open(os.path.join(REF, "sample_data", "input.txt")) as f:
        lines = [ln.strip() for ln in f if ln.strip()]
    sents = [ln.split(" ") for ln in lines[20:31]] + [["x"], ["y", "z", "w"]]
    arrs = {f"p_{k}": v for k, v in head_state(amodel).items()}
    for k, v in amodel.state_dict().items():
        if k.startswith("phobert."):
            arrs[f"enc_{k}"] = v.detach().numpy()
    arrs["sent_text"] = np.asarray([" ".join(s_) for s_ in sents])
    data_t = amodel.chunk_batches(amodel.process_data_tagger(batch_text=sents), len(sents))
    data_p = amodel.chunk_batches(amodel.process_data_parser(batch_text=sents), len(sents))
    tok, fsw, wmask, nw, oidx, lens = amodel.get_batch(0, data_t)
    tok1, fsw1, wmask1, nw1, oidx1, lens1 = amodel.get_batch(0, data_p)
    g = torch.Generator().manual_seed(9)
    gold_head = torch.randint(0, 3, (len(lens1), max(lens1) - 1), generator=g)
    gold_rel = torch.randint(0, len(amodel.vocab["deprel"]), gold_head.shape, generator=g)
    gold_upos = torch.randint(0, len(amodel.vocab["upos"]), (len(lens), max(lens)), generator=g)
    gold_ner = torch.randint(0, len(amodel.vocab["ner_tag"]), gold_upos.shape, generator=g)
    dep_batch = (tok1, fsw1, wmask1, gold_head, gold_rel, nw1, oidx1, lens1)
    pos_batch = (tok, fsw, gold_upos, oidx, lens)
    ner_batch = (tok, fsw, wmask, gold_ner, oidx, lens)
    with torch.no_grad():
        for unsort in (True, False):
            tag = "u" if unsort else "s"
            dep = trainer.predict_dep(dep_batch, unsort=unsort)
            pos = trainer.predict_pos(pos_batch, unsort=unsort)
            ner = trainer.predict_ner(ner_batch, unsort=unsort)
            arrs[f"dep_{tag}"] = np.asarray(["|".join(h + ":" + r for h, r in sent) for sent in dep])
            arrs[f"pos_{tag}"] = np.asarray(["|".join(w[0] for w in sent) for sent in pos])
            arrs[f"ner_{tag}"] = np.asarray(["|".join(sent) for sent in ner])
    # the joint loss as JointTrainer.update forms it before backward (trainer.py:62-99 over jointmodel/model.py:243-281),
    # with the model in .eval(): every dropout the identity.  Gold trees here never point a word at itself.
    loss_head = torch.zeros_like(gold_head)
    for b, ln in enumerate(lens1):
        for i in range(1, ln):
            h = int(torch.randint(0, ln - 1, (1,), generator=g))
            loss_head[b, i - 1] = h if h < i else h + 1
    with torch.no_grad():
        lp, ln_, ld, lt, _ = trainer.model.forward(tok1, fsw1, wmask1, nw1, loss_head, gold_rel, lens1, tok, fsw, gold_upos,
                                                  lens, tok, fsw, wmask, lens, gold_ner, 0.4, 0.2, 0.4)
    arrs["loss_gold_head"] = loss_head.numpy()
    arrs["joint_losses"] = np.asarray([lt.item(), lp.item(), ln_.item(), ld.item()], dtype=np.float32)  # update()'s order
    print("joint losses (total, pos, ner, dep):", arrs["joint_losses"])
    for name, t in (("dep_tokens", tok1), ("dep_first", fsw1), ("dep_mask", wmask1), ("dep_gold_head", gold_head),
                    ("dep_gold_rel", gold_rel), ("tag_tokens", tok), ("tag_first", fsw), ("tag_mask", wmask),
                    ("gold_upos", gold_upos), ("gold_ner", gold_ner)):
        arrs[name] = t.numpy()
    arrs["dep_nwords"], arrs["dep_oidx"], arrs["dep_lens"] = np.int64(nw1), np.asarray(oidx1), np.asarray(lens1)
    arrs["tag_oidx"], arrs["tag_lens"] = np.asarray(oidx), np.asarray(lens)
    save("trainer_small.npz", **arrs)


# ---------------------------------------------------------------------------- F9 training-side parser loss
def gen_dep_loss():
    """The eval=False branch of the training model's dep_forward (models/jointmodel/model.py:162-236) in .eval() mode
    (every dropout the identity) on ragged batches with random gold trees: the loss value and the three
    cross-entropy sums the reference's own `crit_dep` returned on the way (the distance term is the rest)."""
    import tempfile
    amodel = build_joint_model(seed=808)
    from phonlp.models.jointmodel.trainer import JointTrainer
    with tempfile.TemporaryDirectory() as td:
        amodel.phobert.save_pretrained(td)
        args = dict(amodel.args, pretrained_lm=td)
        trainer = JointTrainer(args=args, vocab=amodel.vocab, config_phobert=amodel.phobert.config, use_cuda=False)
    missing, unexpected = trainer.model.load_state_dict(amodel.state_dict(), strict=False)
    assert not unexpected and all(m.startswith("drop_replacement") for m in missing), (missing, unexpected)
    model = trainer.model.eval()
    with open(os.path.join(REF, "sample_data", "input.txt")) as f:
        lines = [ln.strip() for ln in f if ln.strip()]
    arrs = {f"p_{k}": v for k, v in head_state(amodel).items()}
    recorded = []
    crit = model.crit_dep

    class Recording(torch.nn.Module):  # the reference's own criterion, its return values noted on the way
        def forward(self, x, t):
            out = crit(x, t)
            recorded.append(out.item())  # the value now: the caller goes on adding to the first one in place
            return out

    model.crit_dep = Recording()
    captured = []
    orig = model.phobert.forward
    model.phobert.forward = lambda *a, **kw: (captured.append(orig(*a, **kw)), captured[-1])[1]
    g = torch.Generator().manual_seed(11)
    cases = [lines[40:52] + ["x", "y z w"], lines[60:63], ["p q"]]
    for c, text in enumerate(cases):
        sents = [ln.split(" ") for ln in text]
        data_p = amodel.chunk_batches(amodel.process_data_parser(batch_text=sents), len(sents))
        tok, fsw, wmask, _, _, lens = amodel.get_batch(0, data_p)
        B, L = len(lens), max(lens)
        head = torch.zeros(B, L - 1, dtype=torch.long)
        rel = torch.zeros(B, L - 1, dtype=torch.long)
        for b, ln in enumerate(lens):
            for i in range(1, ln):  # any head but the word itself, as in a treebank (zeros stay on padding)
                h = int(torch.randint(0, ln - 1, (1,), generator=g))
                head[b, i - 1] = h if h < i else h + 1
                rel[b, i - 1] = int(torch.randint(4, len(amodel.vocab["deprel"]), (1,), generator=g))
        nwords = int(sum(lens))
        del recorded[:]
        with torch.no_grad():
            loss, _ = model.dep_forward(tok, fsw, wmask, nwords, lens, head, rel, eval=False)
        enc = captured[-1][2][-1].detach()
        arrs[f"c{c}_enc"] = torch.stack([enc[i].index_select(0, fsw[i]) for i in range(B)]).numpy()
        arrs[f"c{c}_lens"] = np.asarray(lens, dtype=np.int64)
        arrs[f"c{c}_head"], arrs[f"c{c}_rel"] = head.numpy(), rel.numpy()
        arrs[f"c{c}_nwords"] = np.int64(nwords)
        arrs[f"c{c}_loss"] = np.float32(loss.item())
        arrs[f"c{c}_ce"] = np.asarray(recorded, dtype=np.float32)  # arc, label, linearization
        print(f"dep loss case {c}: B={B} L={L} loss={loss.item():.6f} ce={[round(r, 4) for r in recorded]}")
        # POS loss of the same sentences (jointmodel/model.py:107-135: crit_pos, ignore_index 1, mean over the rest)
        data_t = amodel.chunk_batches(amodel.process_data_tagger(batch_text=sents), len(sents))
        tok_t, fsw_t, _, _, _, lens_t = amodel.get_batch(0, data_t)
        upos = torch.randint(0, len(amodel.vocab["upos"]), (len(lens_t), max(lens_t)), generator=g)
        with torch.no_grad():
            ploss, ppred = model.pos_forward(tok_t, fsw_t, lens_t, use_soft_pos=False, upos=upos)
        enc_t = captured[-1][2][-1].detach()
        arrs[f"c{c}_tag_enc"] = torch.stack([enc_t[i].index_select(0, fsw_t[i]) for i in range(len(lens_t))]).numpy()
        arrs[f"c{c}_tag_lens"] = np.asarray(lens_t, dtype=np.int64)
        arrs[f"c{c}_upos"] = upos.numpy()
        arrs[f"c{c}_pos_loss"] = np.float32(ploss.item())
        print(f"pos loss case {c}: {ploss.item():.6f}")
    arrs["n_cases"] = np.int64(len(cases))
    save("dep_loss_small.npz", **arrs)



# ---------------------------------------------------------------------------- F9b backward of the parser loss
def gen_dep_loss_grad():
    """The backward that autograd derives from models/jointmodel/model.py:209-236 (eval=False branch of dep_forward,
    model in .eval()): gradients of the loss w.r.t. the OUTPUTS of the four scorers (unlabeled / deprel / linearization /
    distance), read off the reference's own graph after loss.backward().  The deprel / linearization / distance
    gradients are non-zero only at the gold head (checked here), so only those entries are stored."""
    import tempfile
    amodel = build_joint_model(seed=909)
    from phonlp.models.jointmodel.trainer import JointTrainer
    with tempfile.TemporaryDirectory() as td:
        amodel.phobert.save_pretrained(td)
        args = dict(amodel.args, pretrained_lm=td)
        trainer = JointTrainer(args=args, vocab=amodel.vocab, config_phobert=amodel.phobert.config, use_cuda=False)
    missing, unexpected = trainer.model.load_state_dict(amodel.state_dict(), strict=False)
    assert not unexpected and all(m.startswith("drop_replacement") for m in missing), (missing, unexpected)
    model = trainer.model.eval()
    with open(os.path.join(REF, "sample_data", "input.txt")) as f:
        lines = [ln.strip() for ln in f if ln.strip()]
    arrs = {f"p_{k}": v for k, v in head_state(amodel).items()}
    captured = []
    orig = model.phobert.forward
    def spy(*a, **kw):  # the encoder's last layer stays in the graph: its .grad is the gradient w.r.t. the encoder rows
        out = orig(*a, **kw)
        out[2][-1].retain_grad()
        captured.append(out)
        return out

    model.phobert.forward = spy
    grads = {}

    def keep(name):
        def hook(_mod, _inp, out):
            # the gradient that flows back INTO the scorer.  dep_forward edits the scorer's output in place through a
            # view, which detaches tensor hooks from it; so the caller gets a copy to edit and the hook sits on the
            # untouched original (a clone is the identity for values and gradients)
            out.register_hook(lambda gr: grads.__setitem__(name, gr.detach().clone()))
            return out.clone()
        return hook

    for name in ("unlabeled", "deprel", "linearization", "distance"):
        getattr(model, name).register_forward_hook(keep(name))
    g = torch.Generator().manual_seed(12)
    cases = [lines[70:80] + ["x", "y z w"], lines[5:8], ["p q r"]]
    for c, text in enumerate(cases):
        sents = [ln.split(" ") for ln in text]
        data_p = amodel.chunk_batches(amodel.process_data_parser(batch_text=sents), len(sents))
        tok, fsw, wmask, _, _, lens = amodel.get_batch(0, data_p)
        B, L = len(lens), max(lens)
        head = torch.zeros(B, L - 1, dtype=torch.long)
        rel = torch.zeros(B, L - 1, dtype=torch.long)
        for b, ln in enumerate(lens):
            for i in range(1, ln):  # any head but the word itself, as in a treebank (zeros stay on padding)
                h = int(torch.randint(0, ln - 1, (1,), generator=g))
                head[b, i - 1] = h if h < i else h + 1
                rel[b, i - 1] = int(torch.randint(4, len(amodel.vocab["deprel"]), (1,), generator=g))
        nwords = int(sum(lens))
        model.zero_grad()
        grads.clear()
        loss, _ = model.dep_forward(tok, fsw, wmask, nwords, lens, head, rel, eval=False)
        loss.backward()
        enc = captured[-1][2][-1].detach()
        O = len(amodel.vocab["deprel"])
        d_unl = grads["unlabeled"].squeeze(3)      # B x L x L
        d_dep = grads["deprel"]                    # B x L x L x O
        d_lin = grads["linearization"].squeeze(3)  # B x L x L
        d_dst = grads["distance"].squeeze(3)
        hd = head.unsqueeze(2)
        at = lambda t: torch.gather(t[:, 1:], 2, hd).squeeze(2)  # B x (L-1): the entry at the gold head
        d_dep_h = torch.gather(d_dep[:, 1:], 2, head.view(B, L - 1, 1, 1).expand(-1, -1, -1, O)).squeeze(2)  # B x (L-1) x O
        # everything else is exactly zero in the reference's gradient
        z = d_dep.clone()
        z[:, 1:].scatter_(2, head.view(B, L - 1, 1, 1).expand(-1, -1, -1, O), 0.0)
        assert not z.any()
        for t in (d_lin, d_dst):
            z = t.clone()
            z[:, 1:].scatter_(2, hd, 0.0)
            assert not z.any()
        assert not d_unl[:, 0].any()
        arrs[f"c{c}_enc"] = torch.stack([enc[i].index_select(0, fsw[i]) for i in range(B)]).numpy()
        arrs[f"c{c}_d_enc"] = torch.stack([captured[-1][2][-1].grad[i].index_select(0, fsw[i]) for i in range(B)]).numpy()
        arrs[f"c{c}_lens"] = np.asarray(lens, dtype=np.int64)
        arrs[f"c{c}_head"], arrs[f"c{c}_rel"] = head.numpy(), rel.numpy()
        arrs[f"c{c}_nwords"] = np.int64(nwords)
        arrs[f"c{c}_loss"] = np.float32(loss.item())
        arrs[f"c{c}_d_unlabeled"] = d_unl.numpy()
        arrs[f"c{c}_d_label"] = d_dep_h.numpy()
        arrs[f"c{c}_d_lin"] = at(d_lin).numpy()
        arrs[f"c{c}_d_dist"] = at(d_dst).numpy()
        # ... and what the same backward left in the parameters of the four scorers and of dep_hid
        for pname, prm in model.named_parameters():
            if prm.grad is not None and pname.split(".")[0] in ("unlabeled", "deprel", "linearization", "distance", "dep_hid",
                                                                "upos_emb_matrix_dep", "upos_hid", "upos_clf"):
                arrs[f"c{c}_dp_{pname}"] = prm.grad.detach().numpy().copy()
        print(f"dep loss grad case {c}: B={B} L={L} loss={loss.item():.6f} |d_unl|={d_unl.abs().sum():.4f} "
              f"|d_label|={d_dep_h.abs().sum():.4f} |d_lin|={at(d_lin).abs().sum():.4f} |d_dist|={at(d_dst).abs().sum():.4f}")
    arrs["n_cases"] = np.int64(len(cases))
    save("dep_loss_grad_small.npz", **arrs)


# ---------------------------------------------------------------------------- F9c backward of the POS loss
def gen_pos_backward():
    """The training model's pos_forward(use_soft_pos=False) (models/jointmodel/model.py:107-135) in .eval() mode, then
    loss.backward(): the .grad of upos_hid / upos_clf and the gradient w.r.t. the gathered encoder rows."""
    import tempfile
    amodel = build_joint_model(seed=1010)
    from phonlp.models.jointmodel.trainer import JointTrainer
    with tempfile.TemporaryDirectory() as td:
        amodel.phobert.save_pretrained(td)
        args = dict(amodel.args, pretrained_lm=td)
        trainer = JointTrainer(args=args, vocab=amodel.vocab, config_phobert=amodel.phobert.config, use_cuda=False)
    missing, unexpected = trainer.model.load_state_dict(amodel.state_dict(), strict=False)
    assert not unexpected and all(m.startswith("drop_replacement") for m in missing), (missing, unexpected)
    model = trainer.model.eval()
    with open(os.path.join(REF, "sample_data", "input.txt")) as f:
        lines = [ln.strip() for ln in f if ln.strip()]
    arrs = {f"p_{k}": v for k, v in head_state(amodel).items()}
    captured = []
    orig = model.phobert.forward

    def spy(*a, **kw):  # the encoder's last layer, kept in the graph: its .grad is the gradient w.r.t. the encoder rows
        out = orig(*a, **kw)
        out[2][-1].retain_grad()
        captured.append(out)
        return out

    model.phobert.forward = spy
    g = torch.Generator().manual_seed(13)
    cases = [lines[20:31] + ["x", "y z w"], lines[90:93], ["p q"]]
    for c, text in enumerate(cases):
        sents = [ln.split(" ") for ln in text]
        data_t = amodel.chunk_batches(amodel.process_data_tagger(batch_text=sents), len(sents))
        tok_t, fsw_t, _, _, _, lens_t = amodel.get_batch(0, data_t)
        upos = torch.randint(0, len(amodel.vocab["upos"]), (len(lens_t), max(lens_t)), generator=g)
        model.zero_grad()
        ploss, _ = model.pos_forward(tok_t, fsw_t, lens_t, use_soft_pos=False, upos=upos)
        ploss.backward()
        last = captured[-1][2][-1]
        B = len(lens_t)
        arrs[f"c{c}_enc"] = torch.stack([last[i].detach().index_select(0, fsw_t[i]) for i in range(B)]).numpy()
        arrs[f"c{c}_d_enc"] = torch.stack([last.grad[i].index_select(0, fsw_t[i]) for i in range(B)]).numpy()
        arrs[f"c{c}_lens"] = np.asarray(lens_t, dtype=np.int64)
        arrs[f"c{c}_upos"] = upos.numpy()
        arrs[f"c{c}_loss"] = np.float32(ploss.item())
        for pname in ("upos_hid.weight", "upos_hid.bias", "upos_clf.weight", "upos_clf.bias"):
            mod, attr = pname.split(".")
            arrs[f"c{c}_dp_{pname}"] = getattr(getattr(model, mod), attr).grad.detach().numpy().copy()
        print(f"pos backward case {c}: B={B} loss={ploss.item():.6f}")
    arrs["n_cases"] = np.int64(len(cases))
    save("pos_backward_small.npz", **arrs)


# ---------------------------------------------------------------------------- F9d backward of the NER loss
def gen_ner_backward():
    """The training model's ner_forward (models/jointmodel/model.py:139-160: soft POS embedding -> ner_tag_clf -> ReLU ->
    CRFLoss) in .eval() mode, then loss.backward(): the .grad of every parameter it reaches and the gradient w.r.t. the
    gathered encoder rows."""
    import tempfile
    amodel = build_joint_model(seed=1111)
    from phonlp.models.jointmodel.trainer import JointTrainer
    with tempfile.TemporaryDirectory() as td:
        amodel.phobert.save_pretrained(td)
        args = dict(amodel.args, pretrained_lm=td)
        trainer = JointTrainer(args=args, vocab=amodel.vocab, config_phobert=amodel.phobert.config, use_cuda=False)
    missing, unexpected = trainer.model.load_state_dict(amodel.state_dict(), strict=False)
    assert not unexpected and all(m.startswith("drop_replacement") for m in missing), (missing, unexpected)
    model = trainer.model.eval()
    with torch.no_grad():  # the reference starts the transitions at zero (crf.py:21): make them count
        model.crit_ner._transitions.copy_(torch.randn(model.crit_ner._transitions.shape, generator=torch.Generator().manual_seed(5)) * 0.3)
    with open(os.path.join(REF, "sample_data", "input.txt")) as f:
        lines = [ln.strip() for ln in f if ln.strip()]
    arrs = {f"p_{k}": v for k, v in head_state(model).items()}
    captured = []
    orig = model.phobert.forward

    def spy(*a, **kw):
        out = orig(*a, **kw)
        out[2][-1].retain_grad()
        captured.append(out)
        return out

    model.phobert.forward = spy
    g = torch.Generator().manual_seed(14)
    cases = [lines[30:41] + ["x", "y z w"], lines[50:53], ["p q"]]
    names = ("ner_tag_clf.weight", "ner_tag_clf.bias", "crit_ner._transitions", "upos_emb_matrix_ner", "upos_hid.weight",
             "upos_hid.bias", "upos_clf.weight", "upos_clf.bias")
    params = dict(model.named_parameters())
    for c, text in enumerate(cases):
        sents = [ln.split(" ") for ln in text]
        data_t = amodel.chunk_batches(amodel.process_data_tagger(batch_text=sents), len(sents))
        tok_t, fsw_t, wmask_t, _, _, lens_t = amodel.get_batch(0, data_t)
        B, L = len(lens_t), max(lens_t)
        tags = torch.randint(0, len(amodel.vocab["ner_tag"]), (B, L), generator=g)
        model.zero_grad()
        loss, logits = model.ner_forward(tok_t, fsw_t, wmask_t, lens_t, tags)
        loss.backward()
        last = captured[-1][2][-1]
        arrs[f"c{c}_enc"] = torch.stack([last[i].detach().index_select(0, fsw_t[i]) for i in range(B)]).numpy()
        arrs[f"c{c}_d_enc"] = torch.stack([last.grad[i].index_select(0, fsw_t[i]) for i in range(B)]).numpy()
        arrs[f"c{c}_lens"] = np.asarray(lens_t, dtype=np.int64)
        arrs[f"c{c}_tags"] = tags.numpy()
        arrs[f"c{c}_loss"] = np.float32(loss.item())
        for pname in names:
            arrs[f"c{c}_dp_{pname}"] = params[pname].grad.detach().numpy().copy()
        print(f"ner backward case {c}: B={B} L={L} loss={loss.item():.6f}")
    arrs["n_cases"] = np.int64(len(cases))
    save("ner_backward_small.npz", **arrs)


# ---------------------------------------------------------------------------- F9e the three losses in TRAINING mode
def gen_train_dropout():
    """dep_forward(eval=False), ner_forward and pos_forward(use_soft_pos=False) of the training model (models/jointmodel/
    model.py:107-236) in .train() mode -- word dropout with its replacement vectors (models/common/dropout.py:5-27), the
    nn.Dropout layers on the hidden states and inside the scorers (biaffine.py:82-85) -- with the masks the reference's own
    modules drew read off their inputs and outputs by forward hooks, then loss.backward(): loss, every parameter's .grad
    (the replacement vectors' included) and the gradient w.r.t. the gathered encoder rows.  The encoder itself runs in
    .eval() so that its rows are reproducible.  Masks are stored in the padded [B, L, ...] layout."""
    import tempfile
    from torch.nn.utils.rnn import pack_padded_sequence
    amodel = build_joint_model(seed=1212)
    from phonlp.models.jointmodel.trainer import JointTrainer
    from phonlp.models.common.dropout import WordDropout
    with tempfile.TemporaryDirectory() as td:
        amodel.phobert.save_pretrained(td)
        args = dict(amodel.args, pretrained_lm=td)
        trainer = JointTrainer(args=args, vocab=amodel.vocab, config_phobert=amodel.phobert.config, use_cuda=False)
    missing, unexpected = trainer.model.load_state_dict(amodel.state_dict(), strict=False)
    assert not unexpected and all(m.startswith("drop_replacement") for m in missing), (missing, unexpected)
    model = trainer.model
    with torch.no_grad():
        model.crit_ner._transitions.copy_(torch.randn(model.crit_ner._transitions.shape, generator=torch.Generator().manual_seed(6)) * 0.3)
    model.train()
    model.phobert.eval()
    arrs = {f"p_{k}": v.detach().numpy() for k, v in model.state_dict().items() if not k.startswith("phobert.")}
    arrs["p_dropout"], arrs["p_word_dropout"] = np.float32(args["dropout"]), np.float32(args["word_dropout"])
    captured = []
    orig = model.phobert.forward

    def spy(*a, **kw):
        out = orig(*a, **kw)
        out[2][-1].retain_grad()
        captured.append(out)
        return out

    model.phobert.forward = spy
    rec = {}

    def word_hook(name):
        def hook(_m, inp, out):
            rec.setdefault(name, []).append((out != inp[0]).any(-1).detach().clone())  # the rows that were replaced
        return hook

    def drop_hook(name, p):
        def hook(_m, inp, out):
            # kept entries carry 1/(1-p); where the input is zero the mask cannot be read off and does not matter (the
            # input is a ReLU output or a padded row: no gradient passes)
            rec.setdefault(name, []).append(((out != 0).float() / (1.0 - p)).detach().clone())
        return hook

    for name in ("worddrop_pos", "worddrop_dep", "worddrop_ner"):
        assert isinstance(getattr(model, name), WordDropout)
        getattr(model, name).register_forward_hook(word_hook(name))
    for name in ("drop_pos", "drop_dep"):
        getattr(model, name).register_forward_hook(drop_hook(name, args["dropout"]))
    for name in ("unlabeled", "deprel", "linearization", "distance"):
        getattr(model, name).dropout.register_forward_hook(drop_hook("scorer_" + name, args["dropout"]))
    with open(os.path.join(REF, "sample_data", "input.txt")) as f:
        lines = [ln.strip() for ln in f if ln.strip()]
    g = torch.Generator().manual_seed(15)
    torch.manual_seed(77)  # the reference's modules draw from the global generator
    params = dict(model.named_parameters())
    head_names = [k for k in params if not k.startswith("phobert.")]

    def to_padded(rows, lens):  # PackedSequence-ordered rows [m, ...] -> [B, L, ...] (zeros on padding)
        B, L = len(lens), max(lens)
        idx = pack_padded_sequence(torch.arange(B * L).view(B, L), lens, batch_first=True).data
        out = torch.zeros((B * L,) + tuple(rows.shape[1:]), dtype=rows.dtype)
        out[idx] = rows
        return out.view((B, L) + tuple(rows.shape[1:]))

    def finish(tag, c, loss, lens, fsw):
        last = captured[-1][2][-1]
        B = len(lens)
        arrs[f"{tag}{c}_enc"] = torch.stack([last[i].detach().index_select(0, fsw[i]) for i in range(B)]).numpy()
        arrs[f"{tag}{c}_d_enc"] = torch.stack([last.grad[i].index_select(0, fsw[i]) for i in range(B)]).numpy()
        arrs[f"{tag}{c}_lens"] = np.asarray(lens, dtype=np.int64)
        arrs[f"{tag}{c}_loss"] = np.float32(loss.item())
        for k in head_names:
            if params[k].grad is not None and bool(params[k].grad.abs().sum() > 0):
                arrs[f"{tag}{c}_dp_{k}"] = params[k].grad.detach().numpy().copy()

    cases = [lines[10:19] + ["x", "y z w"], lines[33:36]]
    for c, text in enumerate(cases):
        sents = [ln.split(" ") for ln in text]
        # ---- parser
        data_p = amodel.chunk_batches(amodel.process_data_parser(batch_text=sents), len(sents))
        tok, fsw, wmask, _, _, lens = amodel.get_batch(0, data_p)
        B, L = len(lens), max(lens)
        head = torch.zeros(B, L - 1, dtype=torch.long)
        rel = torch.zeros(B, L - 1, dtype=torch.long)
        for b, ln in enumerate(lens):
            for i in range(1, ln):
                h = int(torch.randint(0, ln - 1, (1,), generator=g))
                head[b, i - 1] = h if h < i else h + 1
                rel[b, i - 1] = int(torch.randint(4, len(amodel.vocab["deprel"]), (1,), generator=g))
        nwords = int(sum(lens))
        model.zero_grad()
        rec.clear()
        loss, _ = model.dep_forward(tok, fsw, wmask, nwords, lens, head, rel, eval=False)
        loss.backward()
        finish("dep", c, loss, lens, fsw)
        arrs[f"dep{c}_head"], arrs[f"dep{c}_rel"], arrs[f"dep{c}_nwords"] = head.numpy(), rel.numpy(), np.int64(nwords)
        assert len(rec["worddrop_pos"]) == 1 and len(rec["worddrop_dep"]) == 1 and len(rec["drop_pos"]) == 1 and len(rec["drop_dep"]) == 8
        arrs[f"dep{c}_m_wd_pos"] = to_padded(rec["worddrop_pos"][0], lens).numpy()
        arrs[f"dep{c}_m_wd"] = to_padded(rec["worddrop_dep"][0], lens).numpy()
        arrs[f"dep{c}_m_drop_pos"] = to_padded(rec["drop_pos"][0], lens).numpy()
        arrs[f"dep{c}_m_hidden"] = torch.stack(rec["drop_dep"]).view(4, 2, B, L, -1).numpy()  # scorer (model.py:185-200 order), W1 / W2 input
        arrs[f"dep{c}_m_scorer"] = torch.stack([torch.stack(rec["scorer_" + n]) for n in ("unlabeled", "deprel", "linearization", "distance")]).numpy()
        print(f"train-mode parser case {c}: B={B} L={L} loss={loss.item():.6f} dropped words {int(rec['worddrop_dep'][0].sum())}")
        # ---- tagger side: NER and POS losses
        data_t = amodel.chunk_batches(amodel.process_data_tagger(batch_text=sents), len(sents))
        tok_t, fsw_t, wmask_t, _, _, lens_t = amodel.get_batch(0, data_t)
        Bt, Lt = len(lens_t), max(lens_t)
        tags = torch.randint(0, len(amodel.vocab["ner_tag"]), (Bt, Lt), generator=g)
        model.zero_grad()
        rec.clear()
        loss, _ = model.ner_forward(tok_t, fsw_t, wmask_t, lens_t, tags)
        loss.backward()
        finish("ner", c, loss, lens_t, fsw_t)
        arrs[f"ner{c}_tags"] = tags.numpy()
        arrs[f"ner{c}_m_wd_pos"] = to_padded(rec["worddrop_pos"][0], lens_t).numpy()
        arrs[f"ner{c}_m_wd"] = to_padded(rec["worddrop_ner"][0], lens_t).numpy()
        arrs[f"ner{c}_m_drop_pos"] = to_padded(rec["drop_pos"][0], lens_t).numpy()
        print(f"train-mode ner case {c}: loss={loss.item():.6f}")
        upos = torch.randint(0, len(amodel.vocab["upos"]), (Bt, Lt), generator=g)
        model.zero_grad()
        rec.clear()
        loss, _ = model.pos_forward(tok_t, fsw_t, lens_t, use_soft_pos=False, upos=upos)
        loss.backward()
        finish("pos", c, loss, lens_t, fsw_t)
        arrs[f"pos{c}_upos"] = upos.numpy()
        arrs[f"pos{c}_m_wd_pos"] = to_padded(rec["worddrop_pos"][0], lens_t).numpy()
        arrs[f"pos{c}_m_drop_pos"] = to_padded(rec["drop_pos"][0], lens_t).numpy()
        print(f"train-mode pos case {c}: loss={loss.item():.6f}")
    arrs["n_cases"] = np.int64(len(cases))
    save("train_dropout_small.npz", **arrs)


# ---------------------------------------------------------------------------- F10 BASELINE config 1 at full head sizes
def gen_config1():
    """BASELINE config 1: the reference's annotate() (annotate_model.py:218-339) over all 100 sentences of
    sample_data/input.txt at PhoBERT-base head sizes (E=768, H=400, TE=100) and sample_data's label-set sizes
    (U=24, O=30, T=12), at several batch sizes.  The encoder is replaced by tests/stub_model.TableEncoder (seeded
    tables) and the head parameters are overwritten from phonlp_amd.synthetic.make_state_dict (seeded), so the fixture
    holds only the sentences and the reference's output files -- no weights."""
    sys.modules.setdefault("gdown", types.ModuleType("gdown"))
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    from transformers import RobertaConfig
    from phonlp.annotate_model import JointModel
    from tests.stub_model import CONFIG1, TableEncoder, ToyTokenizer as StubTok, config1_head_state, config1_vocab
    c = CONFIG1
    cfg = RobertaConfig(vocab_size=c["vocab_size"], hidden_size=c["E"], num_hidden_layers=1, num_attention_heads=12,
                        intermediate_size=64, max_position_embeddings=258, pad_token_id=1, output_hidden_states=True)
    args = {"tag_emb_dim": c["TE"], "deep_biaff_hidden_dim": c["H"], "dropout": 0.5, "word_dropout": 0.33,
            "linearization": True, "distance": True}
    model = JointModel(args, config1_vocab(), cfg, StubTok(c["vocab_size"])).eval()
    model.phobert = TableEncoder(c["vocab_size"], c["E"], seed=c["enc_seed"])
    missing, unexpected = model.load_state_dict(config1_head_state(), strict=False)
    assert not unexpected and all(m.startswith(("phobert.", "drop_replacement")) for m in missing), (missing, unexpected)
    model.eval()
    with open(os.path.join(REF, "sample_data", "input.txt")) as f:
        lines = [ln.strip() for ln in f if ln.strip()]
    arrs = {"sent_text": np.asarray(lines), "batch_sizes": np.asarray([1, 8, 32, 100], dtype=np.int64)}
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.txt"), os.path.join(td, "out.txt")
        with open(fin, "w") as f:
            f.write("\n".join(lines) + "\n")
        for bs in arrs["batch_sizes"].tolist():
            with torch.no_grad():
                model.annotate(input_file=fin, output_file=fout, batch_size=bs)
            arrs[f"out_bs{bs}"] = np.asarray(open(fout).read())
            print(f"config 1, batch_size {bs}: {len(open(fout).read().splitlines())} lines")
    for bs in arrs["batch_sizes"].tolist()[1:]:
        a, b = str(arrs["out_bs1"]).splitlines(), str(arrs[f"out_bs{bs}"]).splitlines()
        print(f"  reference output at batch_size {bs} vs 1: {sum(x != y for x, y in zip(a, b))} of {len(a)} lines differ")
    save("config1_sample.npz", **arrs)


# ---------------------------------------------------------------------------- F7 NER entity scoring
def gen_ner_entities():
    """Runs the reference's own score_by_entity / decode_from_bioes (models/ner/scorer.py, ner/utils.py) on random
    tag sequences over a BIOES vocabulary with the four specials, including sequences that are not well formed
    (I- with no B-, type switches inside an entity, specials in the middle)."""
    sys.modules.setdefault("gdown", types.ModuleType("gdown"))
    sys.path.insert(0, REF_ROOT)
    from phonlp.models.ner import scorer as ref_scorer
    from phonlp.models.ner.utils import decode_from_bioes as ref_decode
    vocab = ["<PAD>", "<UNK>", "<EMPTY>", "<ROOT>", "O"] + [f"{p}-{t}" for t in ("PER", "LOC", "ORG") for p in "BIES"]
    rng = np.random.default_rng(7)
    arrs = {"vocab": np.asarray(vocab)}
    n_cases = 6
    for c in range(n_cases):
        B, L = int(rng.integers(3, 40)), int(rng.integers(5, 70))
        lens = rng.integers(1, L + 1, B)
        lens[0] = L
        # gold: mostly well formed; pred: gold with noise; case 0 uniform random ids (incl. specials)
        gold = rng.integers(0, len(vocab), (B, L))
        if c > 0:
            gold = np.where(rng.random((B, L)) < 0.5, 4, rng.integers(4 if c > 1 else 0, len(vocab), (B, L)))
        pred = np.where(rng.random((B, L)) < (0.1 + 0.15 * c), rng.integers(0, len(vocab), (B, L)), gold)
        gs = [[vocab[t] for t in gold[b, :lens[b]]] for b in range(B)]
        ps = [[vocab[t] for t in pred[b, :lens[b]]] for b in range(B)]
        prec, rec, f1 = ref_scorer.score_by_entity(ps, gs, verbose=False)
        gold_ents = [dict(e, sent_id=i) for i, t in enumerate(gs) for e in ref_decode(t)]
        pred_ents = [dict(e, sent_id=i) for i, t in enumerate(ps) for e in ref_decode(t)]
        correct = sum(1 for p_ in pred_ents if p_ in gold_ents)
        arrs[f"c{c}_pred"], arrs[f"c{c}_gold"], arrs[f"c{c}_lens"] = pred.astype(np.int32), gold.astype(np.int32), lens.astype(np.int32)
        arrs[f"c{c}_counts"] = np.asarray([correct, len(pred_ents), len(gold_ents)], dtype=np.int64)
        arrs[f"c{c}_prf"] = np.asarray([prec, rec, f1], dtype=np.float64)
        # entity lists of the first sentences, flattened (start, end, type index in sorted type names)
        types_ = sorted({e["type"] for e in pred_ents + gold_ents})
        arrs[f"c{c}_pred_ents"] = np.asarray([[e["sent_id"], e["start"], e["end"], types_.index(e["type"])] for e in pred_ents],
                                             dtype=np.int64).reshape(-1, 4)
        arrs[f"c{c}_types"] = np.asarray(types_)
    arrs["n_cases"] = np.int64(n_cases)
    save("ner_entities.npz", **arrs)


if __name__ == "__main__":
    if "--only-ner-entities" in sys.argv:
        gen_ner_entities()
        sys.exit(0)
    if "--only-trainer" in sys.argv:
        gen_trainer()
        sys.exit(0)
    if "--only-crf" in sys.argv:
        gen_crf()
        sys.exit(0)
    if "--only-config1" in sys.argv:
        torch.set_num_threads(8)
        gen_config1()
        sys.exit(0)
    if "--only-dep-loss" in sys.argv:
        gen_dep_loss()
        sys.exit(0)
    if "--only-dep-loss-grad" in sys.argv:
        gen_dep_loss_grad()
        sys.exit(0)
    if "--only-ner-backward" in sys.argv:
        gen_ner_backward()
        sys.exit(0)
    if "--only-pos-backward" in sys.argv:
        gen_pos_backward()
        sys.exit(0)
    if "--only-train-dropout" in sys.argv:
        gen_train_dropout()
        sys.exit(0)
    if "--only-biaffine-backward" in sys.argv:
        gen_biaffine_backward()
        sys.exit(0)
    torch.set_num_threads(4)
    gen_biaffine()
    gen_viterbi()
    gen_crf()
    gen_mst()
    gen_joint("joint_small.npz", zero_init=False)
    gen_joint("joint_zero_init.npz", zero_init=True)
    gen_ner_entities()
    gen_trainer()
    gen_dep_loss()
    gen_dep_loss_grad()
    gen_train_dropout()
    gen_biaffine_backward()
    gen_pos_backward()
    gen_ner_backward()
    gen_config1()
