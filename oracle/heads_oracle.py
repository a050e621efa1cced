"""TEST INFRASTRUCTURE ONLY -- CPU oracle for PhoNLP's task-head hot path.

This module is a from-scratch CPU restatement (torch-CPU GEMMs + numpy decoders) of the
algorithms the reference runs after the PhoBERT encoder.  It exists so that the HIP
kernels in ``phonlp_amd/csrc`` can be checked on a GPU box where ``/root/reference`` does
not exist.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product path (``phonlp_amd``) never does.

Pinning: every function below is checked against outputs of the reference's own code
(loaded by file path in the authoring container) through the fixtures in
``tests/golden/*.npz`` -- see ``tests/golden/make_golden.py`` and
``tests/test_oracle_golden.py``.  The reference ships no value-asserting tests for this
path, so those generated fixtures are the only pin.

All ``ref:`` citations are paths below ``/root/reference/phonlp``.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

NEG_INF = -float("inf")


# --------------------------------------------------------------------------------------
# a1-a3: deep biaffine scorer
# --------------------------------------------------------------------------------------
def pairwise_bilinear(x1: torch.Tensor, x2: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """ref: models/common/biaffine.py:21-37 (PairwiseBilinear.forward).

    out[b,i,j,o] = sum_{d1,d2} x1[b,i,d1] * Wq[d1,o,d2] * x2[b,j,d2] where Wq is the FLAT
    reinterpretation ``weight.reshape(D1, O, D2)`` of the stored (D1, D2, O) parameter
    (quirk 1 in SURVEY.md: the reference views the stage-A rows [d2][o] as [o][d2]).
    The module's ``bias`` parameter is never added (quirk 2).  Returned tensor is
    B x L1 x L2 x O.
    """
    B, L1, D1 = x1.shape
    _, L2, D2 = x2.shape
    O = weight.shape[2]
    wq = weight.contiguous().reshape(D1, O * D2)  # flat view: column index = o*D2 + d2
    stage_a = (x1.reshape(B * L1, D1) @ wq).reshape(B, L1 * O, D2)  # rows (i, o)
    stage_b = torch.bmm(stage_a, x2.transpose(1, 2))  # B x (L1*O) x L2
    return stage_b.reshape(B, L1, O, L2).permute(0, 1, 3, 2)


def deep_biaffine(h1, h2, W1, b1, W2, b2, Wbil):
    """ref: biaffine.py:82-85 (DeepBiaffineScorer.forward), :62-65 (ones column).

    Dropout is the identity in eval mode (ref: models/common/dropout.py:16).
    """
    x1 = F.relu(h1 @ W1.t() + b1)
    x2 = F.relu(h2 @ W2.t() + b2)
    ones1 = x1.new_ones(*x1.shape[:-1], 1)
    ones2 = x2.new_ones(*x2.shape[:-1], 1)
    return pairwise_bilinear(torch.cat([x1, ones1], -1), torch.cat([x2, ones2], -1), Wbil)


def _row_mask(sentlens, B, L):
    lens = torch.as_tensor(list(sentlens), dtype=torch.long)
    return torch.arange(L).view(1, L) < lens.view(B, 1)  # True on real words


# --------------------------------------------------------------------------------------
# a7: POS head (shared by the tagger and the parser path)
# --------------------------------------------------------------------------------------
def pos_head(enc, sentlens, p, emb_key):
    """ref: annotate_model.py:132-142 (tagger) / :172-179 (parser).

    enc: B x L x 768 encoder rows (rows >= sentlen are ignored).
    Returns (upos_logits B x L x U with zero pad rows, pos_emb B x L x E with zero pad rows).
    The reference computes on packed rows; padded rows never reach a later stage, so they
    are zeroed here.
    """
    B, L, _ = enc.shape
    mask = _row_mask(sentlens, B, L).unsqueeze(-1)
    hid = F.relu(enc @ p["upos_hid.weight"].t() + p["upos_hid.bias"])
    logits = hid @ p["upos_clf.weight"].t() + p["upos_clf.bias"]
    logits = logits * mask  # pad(upos_pred): zeros on pad rows
    dis = F.softmax(logits, dim=-1)
    emb = (dis @ p[emb_key]) * mask
    return logits, emb


# --------------------------------------------------------------------------------------
# a7 + a8: tagger_forward heads
# --------------------------------------------------------------------------------------
def tagger_heads(enc, sentlens, p):
    """ref: annotate_model.py:113-152 (JointModel.tagger_forward), everything after the
    first-subword gather.  Returns dict with pos_pred (B x L int64, 0 on pads),
    ner_logits (B x L x T float32, 0 on pads) and intermediates.
    """
    B, L, _ = enc.shape
    mask = _row_mask(sentlens, B, L).unsqueeze(-1)
    logits, emb = pos_head(enc, sentlens, p, "upos_emb_matrix_ner")
    pos_pred = logits.max(2)[1]  # ref :138 (ties -> lowest index on CPU)
    inp = torch.cat([enc, emb], -1)
    ner = F.relu(inp @ p["ner_tag_clf.weight"].t() + p["ner_tag_clf.bias"]) * mask  # ref :149-150
    return {"upos_logits": logits, "pos_pred": pos_pred, "pos_emb": emb, "ner_logits": ner.contiguous()}


# --------------------------------------------------------------------------------------
# a1-a6, a9: dep_forward heads
# --------------------------------------------------------------------------------------
def arc_epilogue(unl, lin, dist):
    """ref: annotate_model.py:192-211.  unl/lin/dist: B x L x L raw scorer outputs
    (lin/dist may be None).  Returns masked pre-softmax arc scores."""
    B, L, _ = unl.shape
    j = torch.arange(L).view(1, 1, L)
    i = torch.arange(L).view(1, L, 1)
    off = (j - i).expand(B, L, L)
    arc = unl.clone()
    if lin is not None:
        arc = arc + F.logsigmoid(lin * torch.sign(off).float())
    if dist is not None:
        pred = 1 + F.softplus(dist)
        arc = arc + (-torch.log((torch.abs(off).float() - pred) ** 2 / 2 + 1))
    eye = torch.eye(L, dtype=torch.bool).unsqueeze(0)
    return arc.masked_fill(eye, NEG_INF)


def _scorer(hidden, p, name):
    return deep_biaffine(
        hidden, hidden,
        p[f"{name}.W1.weight"], p[f"{name}.W1.bias"],
        p[f"{name}.W2.weight"], p[f"{name}.W2.bias"],
        p[f"{name}.scorer.W_bilin.weight"],
    )


def dep_heads(enc, sentlens, p, linearization=True, distance=True, keep_label_scores=True):
    """ref: annotate_model.py:154-216 (JointModel.dep_forward) after the first-subword
    gather; twin: models/jointmodel/model.py:162-241 (eval branch :237-240).

    enc: B x L x 768 (row 0 of each sentence = ROOT token).  Returns dict with
    arc_logp (B x L x L float32), label_argmax (B x L x L int64) and intermediates.
    """
    B, L, _ = enc.shape
    mask = _row_mask(sentlens, B, L).unsqueeze(-1)
    logits, emb = pos_head(enc, sentlens, p, "upos_emb_matrix_dep")
    inp = torch.cat([enc, emb], -1)
    hidden = (inp @ p["dep_hid.weight"].t() + p["dep_hid.bias"]) * mask  # ref :186-187 pad rows exact zeros
    unl = _scorer(hidden, p, "unlabeled").squeeze(3)
    lab = _scorer(hidden, p, "deprel")
    lin = _scorer(hidden, p, "linearization").squeeze(3) if linearization else None
    dist = _scorer(hidden, p, "distance").squeeze(3) if distance else None
    arc = arc_epilogue(unl, lin, dist)
    out = {
        "upos_logits": logits,
        "pos_emb": emb,
        "hidden_out": hidden,
        "unlabeled_raw": unl,
        "lin_raw": lin,
        "dist_raw": dist,
        "arc_logp": F.log_softmax(arc, 2),  # ref :214 -- over ALL L columns, pads included
        "label_argmax": lab.max(3)[1],  # ref :215
    }
    if keep_label_scores:
        out["label_scores"] = lab
    return out


def pos_loss(upos_logits, sentlens, gold_upos, ignore_index=1):
    """ref: models/jointmodel/model.py:128-135 (pos_forward, use_soft_pos=False), dropout off: crit_pos (:94) =
    CrossEntropyLoss(ignore_index=1), i.e. the MEAN cross-entropy over the sentence's own words whose gold tag is not
    id 1 (<UNK>).  upos_logits B x L x U (``pos_head``), gold_upos B x L."""
    B, L, U = upos_logits.shape
    live = _row_mask(sentlens, B, L)
    t = torch.as_tensor(gold_upos, dtype=torch.long).masked_fill(~live, ignore_index)  # pack() drops padded rows
    return F.cross_entropy(upos_logits.reshape(-1, U), t.reshape(-1), ignore_index=ignore_index)


def dep_loss(dep, sentlens, gold_head, gold_rel, number_of_words, linearization=True, distance=True):
    """ref: models/jointmodel/model.py:209-236 (the eval=False branch of dep_forward), dropout off.

    dep: the dict of ``dep_heads`` (needs unlabeled_raw / lin_raw / dist_raw / label_scores); gold_head, gold_rel:
    B x (L-1) int64, zeros on padding, heads counted with ROOT = 0.  Returns (total, [arc, label, lin, dist]):
    total = (arc CE + label CE + lin CE + dist term) / number_of_words, each term a SUM over words.
    Quirks kept: arc and label cross-entropies skip padded words (target -1, ignore_index) and the arc soft-max
    runs over the sentence's own columns only (:212); the linearization and distance terms (:225-234) are NOT
    masked -- padded rows contribute with whatever head the tensor holds there (0 from get_long_tensor)."""
    unl, lin, dist, lab = dep["unlabeled_raw"], dep["lin_raw"], dep["dist_raw"], dep["label_scores"]
    B, L, _ = unl.shape
    # the two terms are added .detach()ed (:197, :204): under autograd the arc cross-entropy reaches the unlabeled scorer only
    arc = arc_epilogue(unl, lin.detach() if linearization else None, dist.detach() if distance else None)  # :188-207
    pad = ~_row_mask(sentlens, B, L)  # word_mask: True on padding
    head = torch.as_tensor(gold_head, dtype=torch.long)
    rel = torch.as_tensor(gold_rel, dtype=torch.long)
    ce = lambda x, t: F.cross_entropy(x, t, ignore_index=-1, reduction="sum")  # crit_dep (:93)
    arc_t = arc[:, 1:, :].masked_fill(pad.unsqueeze(1), NEG_INF)  # :211-212
    t_arc = ce(arc_t.reshape(-1, L), head.masked_fill(pad[:, 1:], -1).reshape(-1))  # :213-216
    O = lab.shape[3]
    lab_h = torch.gather(lab[:, 1:], 2, head.view(B, L - 1, 1, 1).expand(-1, -1, -1, O)).reshape(-1, O)  # :218-221
    t_lab = ce(lab_h, rel.masked_fill(pad[:, 1:], -1).reshape(-1))  # :222-223
    off = (torch.arange(L).view(1, 1, L) - torch.arange(L).view(1, L, 1)).expand(B, L, L)
    t_lin = unl.new_zeros(())
    if linearization:  # :225-230
        s = torch.gather(lin[:, 1:], 2, head.unsqueeze(2)).view(-1)
        two = torch.stack([-s / 2, s / 2], 1)
        t_lin = ce(two, torch.gather((off[:, 1:] > 0).long(), 2, head.unsqueeze(2)).view(-1))
    t_dist = unl.new_zeros(())
    if distance:  # :199-203, :232-234
        kld = -torch.log((torch.abs(off).float() - (1 + F.softplus(dist))) ** 2 / 2 + 1)
        t_dist = -torch.gather(kld[:, 1:], 2, head.unsqueeze(2)).sum()
    total = (t_arc + t_lab + t_lin + t_dist) / number_of_words  # :236
    return total, [t_arc, t_lab, t_lin, t_dist]


def dep_loss_grad(dep, sentlens, gold_head, gold_rel, number_of_words, linearization=True, distance=True):
    """The backward of ``dep_loss`` w.r.t. the four scorers' outputs, in closed form -- what autograd derives from
    models/jointmodel/model.py:209-236 (the terms added to the unlabeled scores are .detach()ed, :197,:204, so the arc
    cross-entropy reaches the unlabeled scorer only; crit_dep = CrossEntropyLoss(ignore_index=-1, reduction="sum"), :93):

      d_unlabeled[b,i,j] = (softmax_{j' < len}(arc[b,i,:])[j] - [j == head_i]) / nw   rows 1 <= i < len, columns j < len
      d_label[b,i,o]     = (softmax_o(label[b,i,head_i,:])[o] - [o == rel_i]) / nw    rows 1 <= i < len (at the gold head)
      d_lin[b,i]         = (sigmoid(s) - [head_i > i]) / nw,  s = lin[b,i,head_i]     EVERY row i >= 1 (:225-230: the
                           two-class cross-entropy of (-s/2, s/2) is not masked)
      d_dist[b,i]        = -(|head_i - i| - pred) * sigmoid(x) / ((|head_i - i| - pred)^2 / 2 + 1) / nw,
                           pred = 1 + softplus(x), x = dist[b,i,head_i]               EVERY row i >= 1 (:232-234)
    Everything else is zero.  Returns a dict of B x L x L, B x L x O, B x L, B x L tensors (row 0 zero)."""
    unl, lin, dist, lab = dep["unlabeled_raw"], dep["lin_raw"], dep["dist_raw"], dep["label_scores"]
    B, L, _ = unl.shape
    O = lab.shape[3]
    nw = float(number_of_words)
    arc = arc_epilogue(unl, lin if linearization else None, dist if distance else None)
    live = _row_mask(sentlens, B, L)  # True on real words
    head = torch.zeros(B, L, dtype=torch.long)
    head[:, 1:] = torch.as_tensor(gold_head, dtype=torch.long)
    rel = torch.zeros(B, L, dtype=torch.long)
    rel[:, 1:] = torch.as_tensor(gold_rel, dtype=torch.long)
    rows = live.clone()
    rows[:, 0] = False
    d_unl = torch.zeros(B, L, L)
    d_lab = torch.zeros(B, L, O)
    for b in range(B):
        n = int(live[b].sum())
        for i in range(1, n):
            p = torch.softmax(arc[b, i, :n].double(), 0)
            p[head[b, i]] -= 1.0
            d_unl[b, i, :n] = (p / nw).float()
            q = torch.softmax(lab[b, i, head[b, i]].double(), 0)
            q[rel[b, i]] -= 1.0
            d_lab[b, i] = (q / nw).float()
    ii = torch.arange(L).view(1, L).expand(B, L)
    d_lin = torch.zeros(B, L)
    d_dist = torch.zeros(B, L)
    if linearization:
        s = torch.gather(lin, 2, head.unsqueeze(2)).squeeze(2).double()
        d_lin = ((torch.sigmoid(s) - (head > ii).double()) / nw).float()
        d_lin[:, 0] = 0.0
    if distance:
        x = torch.gather(dist, 2, head.unsqueeze(2)).squeeze(2).double()
        diff = (head - ii).abs().double() - (1 + F.softplus(x))
        d_dist = (-diff * torch.sigmoid(x) / (diff * diff / 2 + 1) / nw).float()
        d_dist[:, 0] = 0.0
    return {"d_unlabeled": d_unl, "d_label": d_lab, "d_lin": d_lin, "d_dist": d_dist}


# --------------------------------------------------------------------------------------
# a10: Viterbi
# --------------------------------------------------------------------------------------
def viterbi_decode(scores: np.ndarray, trans: np.ndarray):
    """ref: models/common/crf.py:103-126.  scores l x T, trans[prev, cur] T x T.
    Every value is one float32 add; argmax ties -> first index.  Returns (tags list, best)."""
    scores = np.asarray(scores)
    l, T = scores.shape
    back = np.zeros((l, T), dtype=np.int32)
    prev = scores[0].copy()
    for t in range(1, l):
        cand = prev[:, None] + trans  # [prev, cur]
        back[t] = cand.argmax(axis=0)
        prev = scores[t] + cand.max(axis=0)
    last = int(prev.argmax())
    tags = [last]
    for t in range(l - 1, 0, -1):
        last = int(back[t][last])
        tags.append(last)
    tags.reverse()
    return tags, prev.max()


# --------------------------------------------------------------------------------------
# a11: CRF forward algorithm / NLL
# --------------------------------------------------------------------------------------
def _lse(x, dim):
    m = x.max(dim=dim, keepdim=True)[0]
    return m.squeeze(dim) + torch.log(torch.exp(x - m).sum(dim=dim))


def crf_log_norm(inputs: torch.Tensor, masks: torch.Tensor, trans: torch.Tensor) -> torch.Tensor:
    """ref: crf.py:79-100 (+ log_sum_exp :129-138).  inputs B x L x T, masks B x L bool with
    True on PADDING, trans[prev, cur].  Returns log partition, shape B."""
    alphas = inputs[:, 0, :]
    for t in range(1, inputs.shape[1]):
        new = inputs[:, t, :] + _lse(alphas.unsqueeze(2) + trans.unsqueeze(0), 1)
        pad = masks[:, t].unsqueeze(1)
        alphas = torch.where(pad, alphas, new)
    return _lse(alphas, 1)


def crf_nll(inputs, masks, tags, trans, batch_average=True):
    """ref: crf.py:24-46 (CRFLoss.forward), unary :48-59, binary :61-77."""
    B, L, T = inputs.shape
    unary = torch.gather(inputs, 2, tags.unsqueeze(2)).squeeze(2).masked_fill(masks, 0).sum(1)
    pair = trans[tags[:, :-1], tags[:, 1:]].masked_fill(masks[:, 1:], 0).sum(1)
    ll = unary + pair - crf_log_norm(inputs, masks, trans)
    loss = (-ll).sum()
    if batch_average:
        return loss / B
    return loss / (masks.eq(0).sum() + 1e-8)


# --------------------------------------------------------------------------------------
# a12: Chu-Liu-Edmonds, single root
# --------------------------------------------------------------------------------------
def last_listed_cycle(heads: np.ndarray):
    """The cycle that ref chuliu_edmonds.py:67 ``cycles.pop()`` selects.

    ref ``tarjan`` (:6-49) walks head->dependent edges from node 0, 1, ... and appends each
    strongly connected component of size > 1 when its first-visited member is closed.  In
    a head-pointer graph a cycle can only be entered from its own members, so cycles are
    appended in ascending order of their smallest node; ``pop()`` therefore returns the
    cycle whose minimum node index is largest.  Returns a bool mask or None.
    """
    n = len(heads)
    state = np.zeros(n, dtype=np.int8)
    best_min, best = -1, None
    for s in range(n):
        path = []
        v = s
        while state[v] == 0:
            state[v] = 1
            path.append(v)
            v = int(heads[v])
        if state[v] == 1:  # closed a cycle inside the current walk
            cyc = path[path.index(v):]
            if len(cyc) > 1 and min(cyc) > best_min:
                best_min = min(cyc)
                best = np.zeros(n, dtype=bool)
                best[cyc] = True
        for u in path:
            state[u] = 2
    return best


def chuliu_edmonds(scores: np.ndarray) -> np.ndarray:
    """ref: chuliu_edmonds.py:52-129.  ``scores`` (float64, rows = dependents, cols = heads)
    is modified IN PLACE at this level exactly like the reference (:55-57)."""
    n = scores.shape[0]
    np.fill_diagonal(scores, NEG_INF)
    scores[0, :] = NEG_INF
    scores[0, 0] = 0
    heads = scores.argmax(axis=1)
    in_cyc = last_listed_cycle(heads)
    if in_cyc is None:
        return heads
    cyc = np.flatnonzero(in_cyc)
    rest = np.flatnonzero(~in_cyc)
    edge = scores[cyc, heads[cyc]]  # score of every cycle edge
    total = edge.sum()
    # entering the cycle at member c from outside head r: (s[c,r] - edge[c]) + total  (ref :84)
    enter = scores[np.ix_(cyc, rest)] - edge[:, None] + total
    leave = scores[np.ix_(rest, cyc)]  # outside dependent r picks a head inside the cycle
    enter_pick = enter.argmax(axis=0)
    leave_pick = leave.argmax(axis=1)
    k = len(rest)
    sub = np.zeros((k + 1, k + 1), dtype=scores.dtype)
    sub[:k, :k] = scores[np.ix_(rest, rest)]
    sub[k, :k] = enter[enter_pick, np.arange(k)]
    sub[:k, k] = leave[np.arange(k), leave_pick]
    small = chuliu_edmonds(sub)
    meta_head = int(small[k])
    small = small[:k]
    out = -np.ones(n, dtype=heads.dtype)
    keep = small < k
    out[rest[keep]] = rest[small[keep]]
    out[rest[~keep]] = cyc[leave_pick[~keep]]
    out[cyc] = heads[cyc]
    out[cyc[enter_pick[meta_head]]] = rest[meta_head]
    return out


class NoFiniteTree(Exception):
    """Every single-root candidate scored -inf (ref :162-168 asserts with a broken handler)."""


def chuliu_edmonds_one_root(scores: np.ndarray) -> np.ndarray:
    """ref: chuliu_edmonds.py:133-169.  scores l x l (any float dtype), node 0 = ROOT.
    Returns int64[l] heads (element 0 is ROOT's self head)."""
    work = np.asarray(scores).astype(np.float64)
    tree = chuliu_edmonds(work)
    roots = np.flatnonzero(tree[1:] == 0) + 1
    if len(roots) == 1:
        return tree
    best_score, best_tree = -np.inf, None
    for r in roots:
        trial = work.copy()  # carries the in-place edits of the first call
        r_score = trial[r, 0]
        trial[1:, 0] = NEG_INF
        trial[r, :] = NEG_INF
        trial[r, 0] = 0
        t = chuliu_edmonds(trial)
        picked = trial[np.arange(len(trial)), t]
        score = picked.sum() + r_score if (picked > -np.inf).all() else -np.inf
        if score > best_score:
            best_score, best_tree = score, t
    if best_tree is None:
        raise NoFiniteTree()
    return best_tree


# ------------------------------------------------------------------------------------------ NER entity scoring
def decode_from_bioes(tags):
    """ref models/ner/utils.py:85-128: entities as (first index, last index, type of the last tag) from BIOES tag
    strings; 'O'/None closes an entity, B-/S- close and open, I-/E- extend (opening one if none is open), any other
    string (e.g. <PAD>) is ignored."""
    res, idxs, ctype = [], [], None
    for i, tag in enumerate(tags):
        tag = "O" if tag is None else tag
        if tag == "O":
            if idxs:
                res.append((idxs[0], idxs[-1], ctype))
            idxs = []
        elif tag.startswith("B-") or tag.startswith("S-"):
            if idxs:
                res.append((idxs[0], idxs[-1], ctype))
            idxs, ctype = [i], tag[2:]
            if tag.startswith("S-"):
                res.append((i, i, ctype))
                idxs = []
        elif tag.startswith("I-") or tag.startswith("E-"):
            idxs.append(i)
            ctype = tag[2:]
            if tag.startswith("E-"):
                res.append((idxs[0], idxs[-1], ctype))
                idxs = []
    if idxs:
        res.append((idxs[0], idxs[-1], ctype))
    return res


def ner_entity_counts(pred_seqs, gold_seqs):
    """ref models/ner/scorer.py:13-57: (correct, guessed, gold) entity counts over parallel lists of tag-string
    sequences; an entity matches when sentence, first index, last index and type agree."""
    correct = guessed = gold = 0
    for p, g in zip(pred_seqs, gold_seqs):
        ge = set(decode_from_bioes(g))
        pe = decode_from_bioes(p)
        gold += len(decode_from_bioes(g))
        guessed += len(pe)
        correct += sum(1 for e in pe if e in ge)
    return correct, guessed, gold
