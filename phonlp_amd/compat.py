"""Drop-in replacements with the reference's own names, signatures and return types, running on the HIP
kernels.  Each mirrors one seam of SURVEY.md section 8b; argument meaning and error behaviour follow the
reference (paths under /root/reference/phonlp).

    from phonlp_amd.compat import viterbi_decode, chuliu_edmonds_one_root, CRFLoss, DeepBiaffineScorer, patch
    model = phonlp.load(save_dir); patch(model); model.annotate(text=...)
"""
import types

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .heads import HeadWeights, TaskHeads, as_sentlens


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("phonlp_amd needs a ROCm device: there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


# ---------------------------------------------------------------------------------------------------
def viterbi_decode(scores, transition_params):
    """ref models/common/crf.py:103-126.  scores: [l, T] numpy, transition_params: [T, T] numpy.
    Returns (list of tag ids, best score) like the reference."""
    dev = _dev()
    s = torch.as_tensor(np.ascontiguousarray(scores, dtype=np.float32)).unsqueeze(0).to(dev)
    tr = torch.as_tensor(np.ascontiguousarray(transition_params, dtype=np.float32)).to(dev)
    lens = torch.tensor([s.shape[1]], dtype=torch.int32, device=dev)
    tags, best = ops.crf_viterbi(s.contiguous(), tr.contiguous(), lens)
    return tags[0].cpu().tolist(), np.float32(best[0].item())


def chuliu_edmonds_one_root(scores):
    """ref models/common/chuliu_edmonds.py:133-169.  scores: [l, l] numpy (rows = dependents).  Returns
    int64[l].  Where the reference asserts (no finite single-root tree) this raises AssertionError too."""
    dev = _dev()
    s = torch.as_tensor(np.ascontiguousarray(scores, dtype=np.float32)).unsqueeze(0).to(dev)
    l = s.shape[1]
    heads, status = ops.mst_one_root(s.contiguous(), torch.tensor([l], dtype=torch.int32, device=dev))
    if int(status[0].item()) != 0:
        raise AssertionError("chuliu_edmonds_one_root: no finite single-root tree (reference asserts here)")
    return heads[0].cpu().numpy().astype(np.int64)


class _CRFNLL(torch.autograd.Function):
    """Batch-averaged CRF NLL with its gradient by forward-backward on the GPU (phonlp_crf_log_partition /
    phonlp_crf_nll_grad)."""

    @staticmethod
    def forward(ctx, x, tr, masks, tags, denom):
        ln, un, bi = ops.crf_log_partition(x, tr, masks, None, tags)
        ctx.save_for_backward(x, tr, masks, tags)
        ctx.denom = denom
        return (-(un + bi - ln)).sum() / denom

    @staticmethod
    def backward(ctx, grad):
        x, tr, masks, tags = ctx.saved_tensors
        dx, dtr = ops.crf_nll_grad(x, tr, masks, None, tags, 1.0 / ctx.denom)
        return dx * grad, dtr * grad, None, None, None


class CRFLoss(nn.Module):
    """ref models/common/crf.py:13-100.  forward(inputs, masks, tag_indices) -> (loss, transitions); masks is
    True on padding.  Differentiable: the loss carries autograd edges to `inputs` and `_transitions` whose backward is
    the forward-backward kernel (the result lives on the GPU; gradients flow back to wherever the inputs are)."""

    def __init__(self, num_tag, batch_average=True):
        super().__init__()
        self._transitions = nn.Parameter(torch.zeros(num_tag, num_tag))
        self._batch_average = batch_average

    def forward(self, inputs, masks, tag_indices):
        dev = _dev()
        x = inputs.to(dev, torch.float32).contiguous()
        tr = self._transitions.to(dev, torch.float32).contiguous()
        denom = float(x.shape[0]) if self._batch_average else float(masks.eq(0).sum().item()) + 1e-8
        loss = _CRFNLL.apply(x, tr, masks.to(dev, torch.uint8).contiguous(),
                             tag_indices.to(dev, torch.int32).contiguous(), denom)
        return loss, self._transitions


class _PosCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, lens, ignore_index):
        loss = ops.pos_loss(x, target, lens, ignore_index)
        ctx.save_for_backward(x, target, lens, loss)
        ctx.ignore_index = ignore_index
        return loss[0]

    @staticmethod
    def backward(ctx, grad):
        x, target, lens, loss = ctx.saved_tensors
        return ops.pos_loss_grad(x, target, lens, loss, ctx.ignore_index) * grad, None, None, None


class PosCrossEntropy(nn.Module):
    """Drop-in for the reference's `crit_pos = nn.CrossEntropyLoss(ignore_index=1)` (models/jointmodel/model.py:94),
    called as at :134 with packed rows: forward(logits [N, U], target [N]) -> mean cross-entropy over the targets
    that are not `ignore_index`.  Differentiable (phonlp_pos_loss / phonlp_pos_loss_grad): assign it to
    `model.crit_pos` and the training step's POS loss and its gradient w.r.t. the logits come from the GPU kernels."""

    def __init__(self, ignore_index=1):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, logits, target):
        dev = _dev()
        x = logits.to(dev, torch.float32).contiguous().unsqueeze(0)
        t = target.to(dev, torch.int32).contiguous().view(1, -1)
        # nn.CrossEntropyLoss raises on a target outside [0, U); the kernel would skip it silently (costs one sync)
        if bool(((t != self.ignore_index) & ((t < 0) | (t >= x.shape[-1]))).any()):
            raise IndexError(f"PosCrossEntropy: target out of range for {x.shape[-1]} classes")
        lens = torch.tensor([x.shape[1]], dtype=torch.int32, device=dev)
        return _PosCE.apply(x, t, lens, self.ignore_index)


class DeepBiaffineScorer(nn.Module):
    """ref models/common/biaffine.py:68-85 with pairwise=True (the only form the reference instantiates).
    forward(input1, input2) -> [B, L, L, O] (a non-contiguous view of [B, L, O, L] memory, like the
    reference).  Both inputs must be the same tensor, as at every reference call site."""

    def __init__(self, input1_size, input2_size, hidden_size, output_size, hidden_func=None, dropout=0, pairwise=True):
        super().__init__()
        if not pairwise or input1_size != input2_size:
            raise NotImplementedError("only the pairwise, same-input form used by PhoNLP is implemented")
        self.W1 = nn.Linear(input1_size, hidden_size)
        self.W2 = nn.Linear(input2_size, hidden_size)
        self.scorer = nn.Module()
        self.scorer.W_bilin = nn.Module()
        self.scorer.W_bilin.weight = nn.Parameter(torch.zeros(hidden_size + 1, hidden_size + 1, output_size))
        self.scorer.W_bilin.bias = nn.Parameter(torch.zeros(output_size))  # loaded, never used (as the reference)
        self._packed = None

    def forward(self, input1, input2):
        if input1 is not input2 and not (input1.data_ptr() == input2.data_ptr() and input1.shape == input2.shape):
            raise NotImplementedError("phonlp_amd.DeepBiaffineScorer needs input1 is input2")
        dev = _dev()
        if self._packed is None:
            g = lambda p: p.detach().to(dev, torch.float32).contiguous()
            self._packed = ops.pack_scorer(g(self.W1.weight), g(self.W1.bias), g(self.W2.weight), g(self.W2.bias),
                                           g(self.scorer.W_bilin.weight))
        W12, b12, Wbt = self._packed
        H, O = self.W1.weight.shape[0], self.scorer.W_bilin.weight.shape[2]
        _, scores = ops.deep_biaffine_label(input1.detach().to(dev, torch.float32).contiguous(), W12, b12, Wbt, H, O, True)
        return scores.permute(0, 1, 3, 2)


# ---------------------------------------------------------------------------------------------------
def _head_state(model):
    """The head parameters and buffers under their state_dict names WITHOUT building the encoder's part of the
    state_dict (that detaches ~200 PhoBERT tensors per call, on a path whose GPU step takes ~2 ms): the model's own
    parameters and its children other than `phobert`, walked directly.  A model that keeps its heads elsewhere and
    overrides state_dict() (tests/stub_model.py) is asked for its state_dict as before."""
    sd = {}
    for name, p in model.named_parameters(recurse=False):
        sd[name] = p.detach()
    for name, b in model.named_buffers(recurse=False):
        sd[name] = b
    for cname, child in model.named_children():
        if cname == "phobert":
            continue
        for k, v in child.named_parameters(prefix=cname):
            sd[k] = v.detach()
        for k, v in child.named_buffers(prefix=cname):
            sd[k] = v
    if "dep_hid.weight" not in sd:
        sd = {k: v for k, v in model.state_dict().items() if not k.startswith("phobert.")}
    return sd


def _fingerprint(sd):
    # in-place updates (optimizer steps, load_state_dict) bump a tensor's version counter; a replaced tensor has
    # another address
    return tuple((k, v._version, v.data_ptr()) for k, v in sd.items())


def refresh_heads(model, device=None, force=False):
    """(Re)packs the head weights of a patched model if any head parameter changed since the last pack -- the packs
    (scorer images, float64 folds, split images) are COPIES, so a training step or load_state_dict after patch() /
    patch_trainer() would otherwise leave the HIP path on stale weights.  Called on entry by every patched seam; costs
    one pass over ~35 tensors' version counters when nothing changed, a full re-pack (tens of ms) when something did."""
    sd = _head_state(model)
    fp = _fingerprint(sd)
    if force or getattr(model, "_amd_fingerprint", None) != fp:
        dev = device or getattr(model, "_amd_device", None) or _dev()
        weights = HeadWeights.from_state_dict(sd, dev, linearization=bool(model.args.get("linearization", True)),
                                              distance=bool(model.args.get("distance", True)))
        model._amd_heads = TaskHeads(weights)
        model._amd_device = dev
        model._amd_fingerprint = _fingerprint(_head_state(model))
    return model._amd_heads


def _encode(model, tokens, first_subword):
    """The part of the path that stays on PyTorch-ROCm: encoder + first-sub-word gather
    (ref annotate_model.py:119-123 / :159-163), as one batched gather instead of a Python loop."""
    dev = _dev()
    enc_dev = next(model.phobert.parameters()).device  # wherever the caller keeps the encoder
    emb = model.phobert(tokens.to(enc_dev))[2][-1]
    return ops.gather_first_subword(emb.float().to(dev).contiguous(), first_subword.to(dev).contiguous(), None)


def _tagger_forward(self, tokens_phobert, words_phobert, sentlens):
    """ref annotate_model.py:113-152 -> ([LongTensor B x L], FloatTensor B x L x T)."""
    out = refresh_heads(self).tagger(_encode(self, tokens_phobert, words_phobert), sentlens)
    return [out["pos_pred"].long()], out["ner_logits"]


def _dep_forward(self, tokens_phobert, first_subword, sentlens):
    """ref annotate_model.py:154-216 -> [np.float32 B x L x L log-probs, np.int64 B x L x L label arg-max]."""
    out = refresh_heads(self).parser(_encode(self, tokens_phobert, first_subword), sentlens)
    return [out["arc_logp"].cpu().numpy(), out["label_argmax"].cpu().numpy().astype(np.int64)]


def _annotate(self, text=None, input_file=None, output_file=None, batch_size=1, output_type=""):
    """ref annotate_model.py:218-339, same arguments, return value and file formats.  Host-side batching uses the
    model's own process_data_* / chunk_batches / get_batch; heads and decoders run on the GPU and only
    O(B*L) integers come back per batch."""
    if text is not None:
        data = [text.split(" ")]
    else:
        with open(input_file) as f:
            data = [ln.strip().split(" ") for ln in f if len(ln.strip()) != 0]
        print("The number of sentences: ", len(data))
    parser_batches = self.chunk_batches(self.process_data_parser(batch_text=data), batch_size)
    tagger_batches = self.chunk_batches(self.process_data_tagger(batch_text=data), batch_size)
    heads_ = refresh_heads(self)
    dev = heads_.w.trans.device
    # Corpus-level result tables on the device: every batch scatters its rows to the sentences' places in the input
    # (the chunk sort and the batch sort composed on the host, ref :270,277,287,291-293), so the whole corpus comes
    # back in ONE device->host copy and the MST status is checked once, not per batch.
    # The tables are kept per LENGTH CLASS (sentence lengths rounded up to a power of two, 8 at least): a corpus's one
    # very long sentence sizes its own small table, not every sentence's row (the reference keeps ragged host lists); at
    # most twice the words of the corpus in all.  slot[k] = (class, row) of sentence k.
    n = len(data)
    width = [max(8, 1 << max(len(s_) - 1, 0).bit_length()) for s_ in data]
    classes = sorted(set(width))
    rows_in = {c: 0 for c in classes}
    slot = []
    for c in width:
        slot.append((c, rows_in[c]))
        rows_in[c] += 1
    tables = {c: torch.zeros((4, rows_in[c], c), dtype=torch.int32, device=dev) for c in classes}  # pos, ner, head, rel
    mst_failures = torch.zeros(1, dtype=torch.int32, device=dev)
    chunk_order = self.data_orig_idx
    start = 0
    with torch.no_grad():
        for i in range(len(tagger_batches)):
            tok, fsw, _, _, oidx, lens = self.get_batch(i, tagger_batches)
            tok1, fsw1, _, _, oidx1, lens1 = self.get_batch(i, parser_batches)
            assert list(oidx) == list(oidx1), "tagger and parser batches must share their sort"
            out = heads_.annotate_batch(_encode(self, tok1, fsw1), lens1, _encode(self, tok, fsw), lens)
            sent = [chunk_order[start + k] for k in oidx]
            counted = False
            for c in sorted({slot[k][0] for k in sent}):  # a batch is sorted by length: one class, seldom two
                dest = torch.tensor([slot[k][1] if slot[k][0] == c else -1 for k in sent],
                                    dtype=torch.int32).to(dev, non_blocking=True)
                table = tables[c]
                ops.scatter_rows(out["pos"], 0, dest, table[0])
                ops.scatter_rows(out["ner"], 0, dest, table[1])
                if counted:
                    ops.scatter_rows(out["heads"], 1, dest, table[2])
                else:  # (the status count covers the whole batch: once)
                    ops.scatter_rows(out["heads"], 1, dest, table[2], out["mst_status"], mst_failures)
                    counted = True
                ops.scatter_rows(out["rels"], 1, dest, table[3])
            start += len(lens)
    host_tables = {c: t.cpu().numpy() for c, t in tables.items()}

    class _Rows:  # host[f, k, :len] of the old dense table, read through the sentence's slot
        def __getitem__(self, key):
            f, k, cols = key
            c, r = slot[k]
            return host_tables[c][f, r, cols]

    host = _Rows()
    if int(mst_failures.item()):
        raise AssertionError("chuliu_edmonds_one_root: no finite single-root tree (reference asserts here)")
    pos_all = [[[u] for u in self.vocab["upos"].unmap(host[0, k, :len(s_)])] for k, s_ in enumerate(data)]
    ner_all = [self.vocab["ner_tag"].unmap(host[1, k, :len(s_)]) for k, s_ in enumerate(data)]
    dep_all = [[[str(int(h)), r] for h, r in zip(host[2, k, :len(s_)], self.vocab["deprel"].unmap(host[3, k, :len(s_)]))]
               for k, s_ in enumerate(data)]
    if text is not None:
        return (data, pos_all, ner_all, dep_all)
    with open(output_file, "w") as f:
        for i, sent in enumerate(data):
            for j, word in enumerate(sent):
                p, n, (h, r) = pos_all[i][j][0], ner_all[i][j], dep_all[i][j]
                if output_type == "conll":
                    cols = [str(j + 1), word, "_", "_", p, "_", h, r, "_", n]
                else:
                    cols = [str(j + 1), word, p, n, h, r]
                f.write("\t".join(cols) + "\n")
            f.write("\n")


# --------------------------------------------------------------------------------------------------- trainer seams
def _predict_dep(self, batch, unsort=True):
    """ref models/jointmodel/trainer.py:101-129 (over jointmodel/model.py:162-239 with eval=True): same batch tuple,
    same return value -- per sentence a list of [head string, deprel string] per word."""
    tokens, first_subword, orig_idx, sentlens = batch[0], batch[1], batch[6], batch[7]
    self.model.eval()  # as the reference (trainer.py:104): update() leaves the model in train mode
    h = refresh_heads(self.model)
    with torch.no_grad():
        dep = h.parser(_encode(self.model, tokens, first_subword), sentlens, want_label_argmax=False)
        heads, status = ops.mst_one_root(dep["arc_logp"], dep["sentlens"])
        if int((status != 0).sum().item()):
            raise AssertionError("chuliu_edmonds_one_root: no finite single-root tree (reference asserts here)")
        XF, T = ops.label_project(dep["hidden_out"], h.w.lab_W12, h.w.lab_b12, h.w.lab_Wbt, h.w.H, h.w.O)
        rels, _ = ops.label_at_heads(T, XF, heads, dep["sentlens"], h.w.H, h.w.O)
    heads, rels = heads.cpu().numpy(), rels.cpu().numpy()
    pred = [[[str(int(heads[b][j])), r] for j, r in
             zip(range(1, sentlens[b]), self.vocab["deprel"].unmap(rels[b][1:sentlens[b]]))] for b in range(len(sentlens))]
    return _unsort(pred, orig_idx) if unsort else pred


def _predict_pos(self, batch, unsort=True):
    """ref trainer.py:131-144 (over jointmodel/model.py:107-137): per sentence a list of [upos string] per word."""
    tokens, first_subword, orig_idx, sentlens = batch[0], batch[1], batch[3], batch[4]
    self.model.eval()  # trainer.py:134
    h = refresh_heads(self.model)
    with torch.no_grad():
        enc = _encode(self.model, tokens, first_subword)
        lens = as_sentlens(sentlens, enc.device)
        _, pos, _ = ops.pos_head(enc, lens, h.w.upos_hid_w, h.w.upos_hid_b, h.w.upos_clf_w, h.w.upos_clf_b, h.w.emb_ner)
    pos = pos.cpu().numpy()
    pred = [[[u] for u in self.vocab["upos"].unmap(pos[b][:sentlens[b]])] for b in range(len(sentlens))]
    return _unsort(pred, orig_idx) if unsort else pred


def _predict_ner(self, batch, unsort=True):
    """ref trainer.py:146-166 (over jointmodel/model.py:139-160 + crf.py:103-126): per sentence a list of tag strings."""
    tokens, first_subword, orig_idx, sentlens = batch[0], batch[1], batch[4], batch[5]
    self.model.eval()  # trainer.py:150
    h = refresh_heads(self.model)
    with torch.no_grad():
        tag = h.tagger(_encode(self.model, tokens, first_subword), sentlens)
        ner, _ = ops.crf_viterbi(tag["ner_logits"], h.w.trans, tag["sentlens"])
    ner = ner.cpu().numpy()
    pred = [self.vocab["ner_tag"].unmap(ner[b][:sentlens[b]]) for b in range(len(sentlens))]
    return _unsort(pred, orig_idx) if unsort else pred


def _dev_loss(self, batch_dep=None, batch_pos=None, batch_ner=None, lambda_pos=1.0, lambda_ner=1.0, lambda_dep=1.0):
    """The forward half of ref trainer.py:62-99 (`update`, over jointmodel/model.py:243-281) with every dropout off
    and no backward: same batches and lambdas, same return order -- (loss, loss_pos, loss_ner, loss_dep) as floats.
    For dev-loss monitoring; the kernels carry no autograd.  Puts the model in eval mode (every dropout off, the
    encoder's included), like the reference's predict_* seams."""
    self.model.eval()
    h = refresh_heads(self.model)
    with torch.no_grad():
        tok, fsw, _, head, rel, nwords, _, lens = batch_dep[:8]
        l_dep = h.parser_loss(_encode(self.model, tok, fsw), lens, head, rel, nwords)[4]
        tok, fsw, upos, _, lens = batch_pos[:5]
        l_pos = h.pos_loss(_encode(self.model, tok, fsw), lens, upos)[0]
        tok, fsw, mask, tags, _, lens = batch_ner[:6]
        tag = h.tagger(_encode(self.model, tok, fsw), lens)
        ln, un, bi = h.crf_log_partition(tag["ner_logits"], pad_mask=mask, gold=tags)
        l_ner = (-(un + bi - ln)).sum() / len(lens)  # CRFLoss, batch_average (crf.py:39-42)
        loss = lambda_pos * l_pos + lambda_ner * l_ner + lambda_dep * l_dep
    return loss.item(), l_pos.item(), l_ner.item(), l_dep.item()


# --------------------------------------------------------------------------------------------------- autograd seam
class _HeadLossFn(torch.autograd.Function):
    """One of the three training losses as a node of a PyTorch graph: forward = the HIP heads' loss value, backward = the
    gradients the HIP kernels computed for the gathered encoder rows and for every head parameter the loss reaches
    (TaskHeads.parser_backward / pos_backward / ner_backward: what autograd derives from ref jointmodel/model.py:107-236).
    Both are computed in forward() -- the kernels keep no graph -- and backward() scales them by the incoming gradient."""

    @staticmethod
    def forward(ctx, enc, run, names, *params):
        out = run(enc.detach())
        ctx.names, ctx.grads, ctx.d_enc = names, out["grads"], out["d_enc"]
        ctx.meta = [(p.shape, p.device, p.dtype) for p in params]
        ctx.enc_meta = (enc.device, enc.dtype)
        return out["loss"].detach().reshape(()).clone()

    @staticmethod
    def backward(ctx, g):
        d_enc = (ctx.d_enc * g.to(ctx.d_enc.device)).to(device=ctx.enc_meta[0], dtype=ctx.enc_meta[1])
        outs = []
        for name, (shape, dev, dt) in zip(ctx.names, ctx.meta):
            gr = ctx.grads.get(name)
            outs.append(None if gr is None else (gr * g.to(gr.device)).reshape(shape).to(device=dev, dtype=dt))
        return (d_enc, None, None, *outs)


def _loss_node(run, enc, params):
    names = tuple(params.keys())
    return _HeadLossFn.apply(enc, run, names, *[params[k] for k in names])


def parser_loss(heads, params, enc, sentlens, head, deprel, number_of_words=None, masks=None):
    """Differentiable `loss` of ref jointmodel/model.py:162-236 (dep_forward, eval=False, every dropout the identity) from the
    gathered encoder rows `enc` [B, L, E].  `params`: {reference parameter name: the live parameter} of the head modules
    (e.g. dict(model.named_parameters()) without the encoder's); `heads` must hold the same values (refresh_heads).
    loss.backward() then leaves in every parameter's .grad, and sends into `enc`, what the reference's own backward does.
    `masks` (TaskHeads._parser_backward_train): the model in .train(), the dropout masks drawn by the caller."""
    run = lambda e: (lambda o: {**o, "loss": o["loss"][4]})(heads.parser_backward(e, sentlens, head, deprel, number_of_words, masks))
    return _loss_node(run, enc, params)


def pos_loss(heads, params, enc, sentlens, upos, ignore_index=1, masks=None):
    """Differentiable `loss` of ref jointmodel/model.py:107-135 (pos_forward, use_soft_pos=False)."""
    run = lambda e: (lambda o: {**o, "loss": o["loss"][0]})(heads.pos_backward(e, sentlens, upos, ignore_index, masks))
    return _loss_node(run, enc, params)


def ner_loss(heads, params, enc, sentlens, tags, masks=None):
    """Differentiable `loss` of ref jointmodel/model.py:139-160 (ner_forward: CRF NLL averaged over the batch)."""
    return _loss_node(lambda e: heads.ner_backward(e, sentlens, tags, masks), enc, params)


def _encode_train(model, tokens, first_subword):
    """Encoder + first-sub-word gather with autograd through both (ref jointmodel/model.py:112-116)."""
    dev = _dev()
    enc_dev = next(model.phobert.parameters()).device
    emb = model.phobert(tokens.to(enc_dev))[2][-1]
    idx = first_subword.to(enc_dev).long().unsqueeze(-1).expand(-1, -1, emb.shape[-1])
    return torch.gather(emb, 1, idx).float().to(dev)


def draw_masks(heads, task, B, L, sentlens, p_drop, p_word, device, generator=None):
    """The random masks of one training step of one task, drawn as the reference's modules draw them (independent Bernoulli
    draws: WordDropout `rand < p` per live word, models/common/dropout.py:19-21; nn.Dropout keeps with 1 - p and scales by
    1 / (1 - p)), in the layout TaskHeads.*_backward(masks=...) takes.  task: "parser" | "ner" | "pos"."""
    w = heads.w
    lens = torch.as_tensor(list(sentlens) if not isinstance(sentlens, torch.Tensor) else sentlens).to(device)
    live = torch.arange(L, device=device)[None, :] < lens[:, None]
    rnd = lambda *shape: torch.rand(*shape, device=device, generator=generator)
    word = lambda: (rnd(B, L) < p_word) & live
    drop = lambda n: (rnd(B, L, n) >= p_drop).float() / max(1.0 - p_drop, 1e-12)
    Hp = w.upos_hid_w.shape[0]
    masks = {"wd_pos": word(), "drop_pos": drop(Hp)}
    if task in ("parser", "ner"):
        masks["wd"] = word()
    if task == "parser":
        D = w.dep_hid_w.shape[0]
        names = ("unlabeled", "deprel", "linearization", "distance")
        masks["hidden"] = {n: (drop(D), drop(D)) for n in names}
        masks["scorer"] = {n: (drop(w.H), drop(w.H)) for n in names}
    return masks


def _update(self, batch_dep=None, batch_pos=None, batch_ner=None, lambda_pos=1.0, lambda_ner=1.0, lambda_dep=1.0):
    """ref trainer.py:62-99 with the three heads' losses AND their backward on the HIP path: same batches, lambdas and
    return value; after it every head parameter's .grad and the encoder's hold what the reference's loss.backward()
    leaves there.  The heads' dropouts (args `dropout`, `word_dropout`) act through masks drawn here per step
    (draw_masks) and handed to the kernels' callers as inputs; the encoder trains as it is (its own dropout included)."""
    args = self.model.args
    p_drop, p_word = float(args.get("dropout", 0.0)), float(args.get("word_dropout", 0.0))
    self.model.train()
    h = refresh_heads(self.model)
    dev = h.w.trans.device
    params = {k: p for k, p in self.model.named_parameters() if not k.startswith("phobert.")}
    if "dep_hid.weight" not in params:  # a model that keeps its heads outside the module tree (tests/stub_model.py)
        params = {k: v for k, v in _head_state(self.model).items() if v.requires_grad}
    train = p_drop > 0.0 or p_word > 0.0
    if train and h.w.repl_dep is None:
        raise ValueError("phonlp_amd update(): word dropout needs the training model's drop_replacement_* parameters")

    def masks(task, enc, lens):
        return draw_masks(h, task, enc.shape[0], enc.shape[1], lens, p_drop, p_word, dev) if train else None

    tok, fsw, _, head, rel, nwords, _, lens = batch_dep[:8]
    enc = _encode_train(self.model, tok, fsw)
    l_dep = parser_loss(h, params, enc, lens, head, rel, nwords, masks("parser", enc, lens))
    tok, fsw, upos, _, lens = batch_pos[:5]
    enc = _encode_train(self.model, tok, fsw)
    l_pos = pos_loss(h, params, enc, lens, upos, masks=masks("pos", enc, lens))
    tok, fsw, _, tags, _, lens = batch_ner[:6]
    enc = _encode_train(self.model, tok, fsw)
    l_ner = ner_loss(h, params, enc, lens, tags, masks("ner", enc, lens))
    loss = lambda_pos * l_pos + lambda_ner * l_ner + lambda_dep * l_dep
    vals = (loss.item(), l_pos.item(), l_ner.item(), l_dep.item())
    loss.backward()
    return vals


def patch_trainer(trainer, device=None, move_encoder=True, train_seam=False):
    """The evaluation-loop twin of patch(): rebinds `predict_dep`, `predict_pos` and `predict_ner` of a reference
    `JointTrainer` (models/jointmodel/trainer.py) so that the dev/test scoring loops of the training script run the
    heads and decoders on the MI355X, and adds `dev_loss` (the loss values of `update` on held-out batches).  `trainer.model` (jointmodel/model.py JointModel) keeps its encoder and its
    training-mode forward; only the eval-mode seams change.  The seams follow the live parameters: they re-pack the
    head weights (refresh_heads) when an optimizer step or load_state_dict changed them since the last call.
    `train_seam=True` also rebinds `update` (trainer.py:62-99): the three losses and their backward through the heads on
    the HIP path as autograd nodes, the heads' dropout masks drawn per step and passed in."""
    dev = device or _dev()
    model = trainer.model
    if move_encoder:
        model.phobert.to(dev)
    refresh_heads(model, dev, force=True)  # re-packed by the seams themselves whenever a head parameter changes
    trainer.predict_dep = types.MethodType(_predict_dep, trainer)
    trainer.predict_pos = types.MethodType(_predict_pos, trainer)
    trainer.predict_ner = types.MethodType(_predict_ner, trainer)
    trainer.dev_loss = types.MethodType(_dev_loss, trainer)  # no counterpart by name: `update` minus dropout and backward
    if train_seam:
        trainer.update = types.MethodType(_update, trainer)
    return trainer


def _unsort(items, original_index):
    """Inverse of the reference's descending-length sort (ref models/common/utils.py:189-195)."""
    assert len(items) == len(original_index), "Number of list elements must match with original indices."
    out = [None] * len(items)
    for item, k in zip(items, original_index):
        out[k] = item
    return out


def patch(model, device=None, move_encoder=True):
    """Swaps the heads of a reference `JointModel` (from `phonlp.load()`) for the HIP path, leaving
    `model.phobert`, the vocab and the tokenizer untouched.  Rebinds `tagger_forward`, `dep_forward` and
    `annotate`; `print_out` and the data helpers keep working unchanged.  `move_encoder=False` leaves the
    encoder where it is (e.g. on the CPU, to reproduce CPU encoder outputs bit for bit)."""
    dev = device or _dev()
    if move_encoder:
        model.phobert.to(dev)
    refresh_heads(model, dev, force=True)
    model.tagger_forward = types.MethodType(_tagger_forward, model)
    model.dep_forward = types.MethodType(_dep_forward, model)
    model.annotate = types.MethodType(_annotate, model)
    return model
