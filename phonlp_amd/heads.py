"""Host-side mirror of the reference's head modules, running on the HIP kernels.

`HeadWeights` takes the reference's own `state_dict` names (ref
annotate_model.py:34-100) and packs them once into the layouts the kernels read.  `TaskHeads`
exposes the two forward seams (`tagger` ~ JointModel.tagger_forward, `parser` ~ JointModel.dep_forward,
both starting AFTER the first-subword gather) and the decoders, everything staying on the device.
"""
import os
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Union

import torch

from . import ops
from . import ops as _ops

SCORERS = ("unlabeled", "linearization", "distance")


def as_sentlens(sentlens: Union[Sequence[int], torch.Tensor], device) -> torch.Tensor:
    if isinstance(sentlens, torch.Tensor):
        return sentlens.to(device=device, dtype=torch.int32).contiguous()
    return torch.tensor(list(sentlens), dtype=torch.int32, device=device)


@dataclass
class HeadWeights:
    """Device-resident head parameters in kernel layout."""
    upos_hid_w: torch.Tensor
    upos_hid_b: torch.Tensor
    upos_clf_w: torch.Tensor
    upos_clf_b: torch.Tensor
    emb_ner: torch.Tensor
    emb_dep: torch.Tensor
    ner_w: torch.Tensor
    ner_b: torch.Tensor
    trans: torch.Tensor
    dep_hid_w: torch.Tensor
    dep_hid_b: torch.Tensor
    arc_W12: torch.Tensor
    arc_b12: torch.Tensor
    arc_Wbt: torch.Tensor
    lab_W12: torch.Tensor
    lab_b12: torch.Tensor
    lab_Wbt: torch.Tensor
    # dep_hid folded into the scorers' first layer (W1.Wdep, W1.bdep + b1; same for W2), used by annotate_batch
    arc_W12f: torch.Tensor
    arc_b12f: torch.Tensor
    lab_W12f: torch.Tensor
    lab_b12f: torch.Tensor
    H: int
    O: int
    n_arc_scorers: int
    linearization: bool
    distance: bool
    # engine 2 split images (planes, inv_scale) of the weights annotate_batch's large layers read
    arc_W12f_s: Optional[tuple] = None
    lab_W12f_s: Optional[tuple] = None
    lab_Wbt_s: Optional[tuple] = None
    lab_Wbtc_s: Optional[tuple] = None  # the same with every relation's rows padded to a multiple of 32 (fused form)
    # fused tagger head (ops.tagger_head): upos_hid rows followed by the enc columns of ner_tag_clf; its embedding columns
    tag_Whn: Optional[torch.Tensor] = None
    tag_bhn: Optional[torch.Tensor] = None
    ner_w_emb: Optional[torch.Tensor] = None
    ner_w_tags: Optional[torch.Tensor] = None
    lab_ones_col: Optional[torch.Tensor] = None  # column H of lab_Wbt: the weights that meet X1's ones column
    # dep_hid AND the soft POS embedding folded into the scorers' first layers: input columns = [enc | tag probabilities]
    # (split images; emb_eye [U, Up] makes the POS tail write the probabilities where it wrote the embedding)
    arc_W12p_s: Optional[tuple] = None
    lab_W12p_s: Optional[tuple] = None
    emb_eye: Optional[torch.Tensor] = None
    # fused arc stages on engine 2 (ops.arc_scores_split): the first layer's weights with every half-segment padded to
    # KC rows (its result is written as the image the arc kernel reads), the bound that fixes that image's row scales
    # (include/phonlp_heads.h phonlp_linear_split_image), the bilinear weights padded to KC x KC per scorer, and the image
    # row of a padded head.  None when H + 1 > 416.
    arc_KC: int = 0
    arc_W12q_s: Optional[tuple] = None
    arc_b12q: Optional[torch.Tensor] = None
    arc_b12q_pad: Optional[torch.Tensor] = None
    arc_bound: Optional[tuple] = None
    arc_Wbq_s: Optional[tuple] = None
    arc_pad_img: Optional[tuple] = None  # (planes [1, S*2*KC*4], inverse scale as a float)
    # word dropout's replacement vectors (training model only, ref jointmodel/model.py:27-35); None for an annotate model
    repl_pos: Optional[torch.Tensor] = None
    repl_ner: Optional[torch.Tensor] = None
    repl_dep: Optional[torch.Tensor] = None

    @classmethod
    def from_state_dict(cls, sd: Dict[str, torch.Tensor], device="cuda", linearization=True, distance=True):
        """`sd`: the reference checkpoint's `model` dict (or `JointModel.state_dict()`); encoder keys are
        ignored.  `linearization` / `distance` are the reference's `args` flags (annotate_model.py:77-94)."""
        def g(name):
            return sd[name].detach().to(device=device, dtype=torch.float32).contiguous()

        def pack(name):
            return ops.pack_scorer(g(f"{name}.W1.weight"), g(f"{name}.W1.bias"), g(f"{name}.W2.weight"),
                                   g(f"{name}.W2.bias"), g(f"{name}.scorer.W_bilin.weight"))

        # dep_hid (ref annotate_model.py:186-187) has no activation, so it folds into the first layer of every
        # scorer: relu(W1 (Wd x + bd) + b1) = relu((W1 Wd) x + (W1 bd + b1)).  The products are formed once in
        # float64 and rounded to fp32 (one rounding per weight, like any stored fp32 weight).
        Wd64, bd64 = sd["dep_hid.weight"].detach().double().cpu(), sd["dep_hid.bias"].detach().double().cpu()

        def fold(w_name, b_name):
            W, b = sd[w_name].detach().double().cpu(), sd[b_name].detach().double().cpu()
            return ((W @ Wd64).to(device=device, dtype=torch.float32).contiguous(),
                    (W @ bd64 + b).to(device=device, dtype=torch.float32).contiguous())

        def pack_folded(name):
            W1, b1 = fold(f"{name}.W1.weight", f"{name}.W1.bias")
            W2, b2 = fold(f"{name}.W2.weight", f"{name}.W2.bias")
            return ops.pack_scorer(W1, b1, W2, b2, g(f"{name}.scorer.W_bilin.weight"))[:2]

        # The soft POS embedding is pos_dis @ upos_emb_matrix_dep (ref annotate_model.py:177-179) with pos_dis a
        # U-vector, so W [x ; pos_dis E] = W[:, :768] x + (W[:, 768:] E^T) pos_dis: the scorers can read
        # [x ; pos_dis] (768 + U columns, U padded to a multiple of 4) instead of [x ; pos_emb] (768 + 100) -- five
        # k16 stages less per W1/W2 GEMM at U = 24.  Folded in float64 together with dep_hid, rounded to fp32 once.
        emb64 = sd["upos_emb_matrix_dep"].detach().double().cpu()  # [U, TE]
        U_ = emb64.shape[0]
        Up = (U_ + 3) // 4 * 4

        def fold_p(w_name, b_name):
            W, b = sd[w_name].detach().double().cpu(), sd[b_name].detach().double().cpu()
            Wf = W @ Wd64
            Wp = torch.zeros((W.shape[0], E_ + Up), dtype=torch.float64)
            Wp[:, :E_] = Wf[:, :E_]
            Wp[:, E_:E_ + U_] = Wf[:, E_:] @ emb64.T
            return (Wp.to(device=device, dtype=torch.float32).contiguous(),
                    (W @ bd64 + b).to(device=device, dtype=torch.float32).contiguous())

        def pack_folded_p(name):
            W1, b1 = fold_p(f"{name}.W1.weight", f"{name}.W1.bias")
            W2, b2 = fold_p(f"{name}.W2.weight", f"{name}.W2.bias")
            return ops.pack_scorer(W1, b1, W2, b2, g(f"{name}.scorer.W_bilin.weight"))[0]

        H = sd["unlabeled.W1.weight"].shape[0]
        O = sd["deprel.scorer.W_bilin.weight"].shape[2]
        E_ = sd["upos_hid.weight"].shape[1]  # encoder width
        if linearization or distance:
            # the arc kernel keeps one slot per scorer; a disabled scorer gets zero weights and is ignored
            packs = []
            fpacks = []
            ppacks = []
            for name, on in zip(SCORERS, (True, linearization, distance)):
                if on:
                    packs.append(pack(name))
                    fpacks.append(pack_folded(name))
                    ppacks.append(pack_folded_p(name))
                else:
                    packs.append(tuple(torch.zeros_like(t) for t in packs[0]))
                    fpacks.append(tuple(torch.zeros_like(t) for t in fpacks[0]))
                    ppacks.append(torch.zeros_like(ppacks[0]))
            arc = tuple(torch.cat([p[k] for p in packs], 0).contiguous() for k in range(3))
            arcf = tuple(torch.cat([p[k] for p in fpacks], 0).contiguous() for k in range(2))
            arcp = torch.cat(ppacks, 0).contiguous()
            n_arc = 3
        else:
            arc = pack("unlabeled")
            arcf = pack_folded("unlabeled")
            arcp = pack_folded_p("unlabeled")
            n_arc = 1
        lab = pack("deprel")
        labf = pack_folded("deprel")
        labp = pack_folded_p("deprel")
        fused_arc = cls._fused_arc_pack(arcp, arcf[1], arc[1], arc[2], H, n_arc)
        return cls(
            upos_hid_w=g("upos_hid.weight"), upos_hid_b=g("upos_hid.bias"),
            upos_clf_w=g("upos_clf.weight"), upos_clf_b=g("upos_clf.bias"),
            emb_ner=g("upos_emb_matrix_ner"), emb_dep=g("upos_emb_matrix_dep"),
            ner_w=g("ner_tag_clf.weight"), ner_b=g("ner_tag_clf.bias"),
            trans=g("crit_ner._transitions"),
            dep_hid_w=g("dep_hid.weight"), dep_hid_b=g("dep_hid.bias"),
            arc_W12=arc[0], arc_b12=arc[1], arc_Wbt=arc[2],
            lab_W12=lab[0], lab_b12=lab[1], lab_Wbt=lab[2],
            arc_W12f=arcf[0], arc_b12f=arcf[1], lab_W12f=labf[0], lab_b12f=labf[1],
            H=H, O=O, n_arc_scorers=n_arc, linearization=bool(linearization), distance=bool(distance),
            arc_W12f_s=tuple(ops.split_rows(arcf[0])), lab_W12f_s=tuple(ops.split_rows(labf[0])),
            lab_Wbt_s=tuple(ops.split_rows(lab[2])),
            lab_Wbtc_s=tuple(ops.split_rows(cls._pad_relations(lab[2], H, O))),
            tag_Whn=torch.cat([g("upos_hid.weight"), g("ner_tag_clf.weight")[:, :E_]], 0).contiguous(),
            tag_bhn=torch.cat([g("upos_hid.bias"), torch.zeros_like(g("ner_tag_clf.bias"))], 0).contiguous(),
            ner_w_emb=g("ner_tag_clf.weight")[:, E_:].contiguous(),
            # ... and with upos_emb_matrix_ner folded in (float64, one rounding): ner_tag_clf.weight[:, E:] @ emb^T [T, U]
            ner_w_tags=(sd["ner_tag_clf.weight"].detach().double().cpu()[:, E_:]
                        @ sd["upos_emb_matrix_ner"].detach().double().cpu().T).to(device=device, dtype=torch.float32).contiguous(),
            lab_ones_col=lab[2][:, H].contiguous(),
            arc_W12p_s=tuple(ops.split_rows(arcp)), lab_W12p_s=tuple(ops.split_rows(labp)),
            emb_eye=torch.eye(U_, Up, dtype=torch.float32, device=device).contiguous(),
            repl_pos=g("drop_replacement_pos") if "drop_replacement_pos" in sd else None,
            repl_ner=g("drop_replacement_ner") if "drop_replacement_ner" in sd else None,
            repl_dep=g("drop_replacement_dep") if "drop_replacement_dep" in sd else None,
            **fused_arc,
        )

    @staticmethod
    def _fused_arc_pack(W12p: torch.Tensor, b12f: torch.Tensor, b12: torch.Tensor, Wbt: torch.Tensor, H: int, S: int):
        """W12p [S*2*KP, K] (first layer on [enc | tag probabilities]), its bias b12f and the unfolded bias b12 (what a
        padded hidden row sees), Wbt [S*KP, KP] -> the operands of ops.linear_split_image / ops.arc_scores_split."""
        KP, KC = ops.kp(H), ops.arc_kc(H)
        if KC == 0:
            return {}
        dev, K = W12p.device, W12p.shape[1]

        def pad_rows(t, rows_per_seg):  # [n_seg * KP, ...] -> [n_seg * KC, ...] with zero rows
            n_seg = t.shape[0] // KP
            out = torch.zeros((n_seg, KC) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
            out[:, :KP] = t.view((n_seg, KP) + tuple(t.shape[1:]))
            return out.view((n_seg * KC,) + tuple(t.shape[1:])).contiguous()

        W12q, b12q, b12q_pad = pad_rows(W12p, KP), pad_rows(b12f, KP), pad_rows(b12, KP)
        # |y[m][n]| <= ||x[m]||_2 ||W[n]||_2 + |b[n]| <= sqrt(K) max|x[m]| max_n ||W[n]||_2 + max|b|; 1 % of slack for
        # the fp32 roundings of the product itself
        bound_c = 1.01 * float(K) ** 0.5 * float(W12q.double().norm(dim=1).max())
        bound_b = 1.01 * max(float(b12q.abs().max()), float(b12q_pad.abs().max())) + 1e-30
        Wbq = torch.zeros((S, KC, KC), dtype=torch.float32, device=dev)
        Wbq[:, :KP, :KP] = Wbt.view(S, KP, KP)
        pad_planes, pad_inv = ops.split_rows(torch.relu(b12q_pad).view(1, -1).contiguous())
        return dict(arc_KC=KC, arc_W12q_s=tuple(ops.split_rows(W12q)), arc_b12q=b12q, arc_b12q_pad=b12q_pad,
                    arc_bound=(bound_c, bound_b), arc_Wbq_s=tuple(ops.split_rows(Wbq.view(S * KC, KC))),
                    arc_pad_img=(pad_planes, float(pad_inv.item())))

    @staticmethod
    def _pad_relations(Wbt: torch.Tensor, H: int, O: int) -> torch.Tensor:
        """[O*KP, KP] -> [O*KPC, KP] for phonlp_label_at_heads_fused: every relation's rows padded with zero rows to KPC
        (a multiple of 32) and ordered by blocks of 32 head features: row (blk*O + o)*32 + r = row o*KP + 32*blk + r."""
        KP, KPC = ops.kp(H), ops.label_kpc(H)
        out = torch.zeros((O, KPC, KP), dtype=Wbt.dtype, device=Wbt.device)
        out[:, :KP] = Wbt.view(O, KP, KP)
        return out.view(O, KPC // 32, 32, KP).permute(1, 0, 2, 3).contiguous().view(O * KPC, KP)


class TaskHeads:
    """POS + NER + dependency heads on the GPU.  All inputs/outputs are device tensors."""

    def __init__(self, weights: HeadWeights):
        self.w = weights
        # A/B switches of annotate_batch, read ONCE here (never per batch); the defaults are the measured best and
        # what the tests run
        env = os.environ.get
        self.cfg = {
            # stage A fused with labels-at-heads behind the MST (T never reaches HBM): default since the round-based MST
            "label_fused": env("PHONLP_LABEL_FUSED", "1") == "1",
            # ... up to this sentence length: beyond it the MST (matrix in global memory, csrc/mst_round.hip) takes longer
            # than stage A, so the unfused form -- stage A beside the MST, labels-at-heads after it -- is shorter
            # (B=64, L=512: 5.0 vs 5.6 ms)
            "fused_max_len": int(env("PHONLP_FUSED_MAX_LEN", "128")),
            # ragged batches whose lengths the host knows: the live words packed (no arithmetic on padding); 0 = off,
            # else the largest live fraction (percent) that is still packed
            "pack_pct": int(env("PHONLP_PACK", "90")),
            # engine 2, sentences of up to 512 words: the arc scorers' stages A and B in one kernel, fed by the image the
            # W1/W2 layer writes (0: stage A as a GEMM of its own, stage B converting fp32 operands: the round-2 form)
            "arc_fused": env("PHONLP_ARC_FUSED", "1") == "1",
        }
        self._side = None
        # what the arc scorers' first layer gives a padded word (zero hidden row): relu of the layer's own bias
        self._arc_pad_row = torch.relu(weights.arc_b12).contiguous()

    # ---- JointModel.tagger_forward after the gather (ref annotate_model.py:132-152)
    def tagger(self, enc: torch.Tensor, sentlens) -> Dict[str, torch.Tensor]:
        w = self.w
        lens = as_sentlens(sentlens, enc.device)
        feat, pos_pred, upos_logits = ops.pos_head(enc.contiguous(), lens, w.upos_hid_w, w.upos_hid_b, w.upos_clf_w,
                                                   w.upos_clf_b, w.emb_ner)
        ner_logits = ops.ner_emission(feat, lens, w.ner_w, w.ner_b)
        return {"pos_pred": pos_pred, "ner_logits": ner_logits, "upos_logits": upos_logits, "feat": feat,
                "sentlens": lens}

    # ---- JointModel.dep_forward after the gather (ref annotate_model.py:172-215)
    def parser(self, enc: torch.Tensor, sentlens, want_raw=False, want_label_scores=False,
               want_label_argmax=True) -> Dict[str, torch.Tensor]:
        w = self.w
        lens = as_sentlens(sentlens, enc.device)
        feat, _, upos_logits = ops.pos_head(enc.contiguous(), lens, w.upos_hid_w, w.upos_hid_b, w.upos_clf_w,
                                            w.upos_clf_b, w.emb_dep)
        hidden = ops.dep_hidden(feat, lens, w.dep_hid_w, w.dep_hid_b)
        arc_logp, raw = ops.deep_biaffine_arc(hidden, w.arc_W12, w.arc_b12, w.arc_Wbt, w.H, w.n_arc_scorers,
                                              w.linearization, w.distance, want_raw)
        out = {"arc_logp": arc_logp, "hidden_out": hidden, "upos_logits": upos_logits, "feat": feat, "sentlens": lens}
        if want_label_argmax or want_label_scores:  # the dense L x L x O contraction (the reference's tensor)
            label_argmax, label_scores = ops.deep_biaffine_label(hidden, w.lab_W12, w.lab_b12, w.lab_Wbt, w.H, w.O,
                                                                 want_label_scores)
            out["label_argmax"] = label_argmax
        if want_raw:
            out["raw"] = raw
        if want_label_scores:
            out["label_scores"] = label_scores  # [B, L, O, L] -- the reference's memory order (biaffine.py:35)
        return out

    # ---- training model's dep_forward(eval=False) after the gather, loss value only (ref jointmodel/model.py:209-236)
    def parser_loss(self, enc: torch.Tensor, sentlens, gold_head, gold_rel, number_of_words=None) -> torch.Tensor:
        """-> fp32 [5] on the device: arc CE, label CE, linearization CE, distance term, total / number_of_words --
        what the reference's training-side `dep_forward` returns as `loss` with the model in .eval() (every dropout
        the identity); forward value only, no autograd.  gold_head / gold_rel: the reference's B x (L-1) tensors
        (ROOT excluded, zeros on padding) or B x L with column 0 unused.  number_of_words defaults to sum(sentlens),
        which is what the reference's loader passes (models/jointmodel/data.py:250)."""
        w = self.w
        dep = self.parser(enc, sentlens, want_raw=w.n_arc_scorers > 1, want_label_argmax=False)
        lens = dep["sentlens"]
        B, L = enc.shape[0], enc.shape[1]

        def with_root(t):
            t = torch.as_tensor(t).to(device=enc.device, dtype=torch.int32)
            if t.shape[1] == L - 1:
                t = torch.nn.functional.pad(t, (1, 0))
            return t.contiguous()

        XF, T = ops.label_project(dep["hidden_out"], w.lab_W12, w.lab_b12, w.lab_Wbt, w.H, w.O)
        nw = float(lens.sum().item()) if number_of_words is None else float(number_of_words)
        return ops.dep_loss(dep["arc_logp"], dep.get("raw"), T, XF, with_root(gold_head), with_root(gold_rel), lens,
                            w.H, w.O, w.linearization and w.n_arc_scorers > 1, w.distance and w.n_arc_scorers > 1, nw)

    # ---- backward of that loss w.r.t. the four scorers' outputs (autograd of ref jointmodel/model.py:209-236)
    def parser_loss_grad(self, enc: torch.Tensor, sentlens, gold_head, gold_rel, number_of_words=None):
        """-> dict of device tensors: d_unlabeled [B,L,L], d_label [B,L,O] (the gradient of deprel_scores[b,i,head_i,:]),
        d_lin, d_dist [B,L] (of lin / dist[b,i,head_i]) -- everything the reference's loss.backward() sends into the
        scorers with the model in .eval(); all other entries of their outputs get no gradient."""
        w = self.w
        dep = self.parser(enc, sentlens, want_raw=w.n_arc_scorers > 1, want_label_argmax=False)
        lens = dep["sentlens"]
        L = enc.shape[1]

        def with_root(t):
            t = torch.as_tensor(t).to(device=enc.device, dtype=torch.int32)
            if t.shape[1] == L - 1:
                t = torch.nn.functional.pad(t, (1, 0))
            return t.contiguous()

        XF, T = ops.label_project(dep["hidden_out"], w.lab_W12, w.lab_b12, w.lab_Wbt, w.H, w.O)
        nw = float(lens.sum().item()) if number_of_words is None else float(number_of_words)
        d_arc, d_label, d_lin, d_dist = ops.dep_loss_grad(
            dep["arc_logp"], dep.get("raw"), T, XF, with_root(gold_head), with_root(gold_rel), lens, w.H, w.O,
            w.linearization and w.n_arc_scorers > 1, w.distance and w.n_arc_scorers > 1, nw)
        return {"d_unlabeled": d_arc, "d_label": d_label, "d_lin": d_lin, "d_dist": d_dist}

    # ---- ... and on through the scorers and dep_hid: what loss.backward() leaves in the parameters' .grad
    def parser_backward(self, enc: torch.Tensor, sentlens, gold_head, gold_rel, number_of_words=None, masks=None):
        """Backward of the parser loss (ref jointmodel/model.py:162-236, model in .eval()) through the four scorers and
        dep_hid -> {"grads": {reference parameter name: gradient in the reference's shape} for `unlabeled` / `deprel` /
        `linearization` / `distance` (.W1 / .W2 .weight, .bias, .scorer.W_bilin.weight), `dep_hid`, `upos_emb_matrix_dep`,
        `upos_hid`, `upos_clf`; "loss": fp32 [5] as parser_loss; "d_hidden": the gradient w.r.t. dep_hid's output [B,L,D]; "d_feat": w.r.t. its input
        [B,L,D] (= [enc ; pos_emb]; padded rows zero: the reference's padding is a constant); "d_enc": w.r.t. the
        gathered encoder rows [B,L,E], both routes added}.  Un-folded weights (the folds are an inference-time rewrite).
        `masks`: the model in .train(), see _parser_backward_train."""
        if masks is not None:
            return self._parser_backward_train(enc, sentlens, gold_head, gold_rel, number_of_words, masks)
        w = self.w
        n = w.n_arc_scorers
        dep = self.parser(enc, sentlens, want_raw=n > 1, want_label_argmax=False)
        lens, hidden, feat = dep["sentlens"], dep["hidden_out"], dep["feat"]
        B, L, D = hidden.shape
        M, H, O, KP = B * L, w.H, w.O, ops.kp(w.H)

        def with_root(t):
            t = torch.as_tensor(t).to(device=enc.device, dtype=torch.int32)
            if t.shape[1] == L - 1:
                t = torch.nn.functional.pad(t, (1, 0))
            return t.contiguous()

        gh, gr = with_root(gold_head), with_root(gold_rel)
        XF, T = ops.label_project(hidden, w.lab_W12, w.lab_b12, w.lab_Wbt, H, O)
        nw = float(lens.sum().item()) if number_of_words is None else float(number_of_words)
        use_lin, use_dist = w.linearization and n > 1, w.distance and n > 1
        loss = ops.dep_loss(dep["arc_logp"], dep.get("raw"), T, XF, gh, gr, lens, H, O, use_lin, use_dist, nw)
        d_arc, d_label, d_lin, d_dist = ops.dep_loss_grad(dep["arc_logp"], dep.get("raw"), T, XF, gh, gr, lens, H, O,
                                                          use_lin, use_dist, nw)
        hm = hidden.reshape(M, D)
        # the one-output scorers: the unlabeled one gets the dense arc gradient, the other two theirs at the gold head
        at_head = lambda v: torch.zeros((B, L, L), dtype=torch.float32, device=enc.device).scatter_(
            2, gh.long().unsqueeze(2), v.unsqueeze(2))
        Gs = [d_arc] + ([at_head(d_lin) if use_lin else None, at_head(d_dist) if use_dist else None] if n > 1 else [])
        XFa = ops.linear(hm, w.arc_W12, w.arc_b12, True, None, 0)  # [M, n * 2KP]: (X1' | X2') per scorer
        dXFa = torch.zeros_like(XFa)
        grads = {}

        def unpack_bilinear(dWbt, Ox):  # [Ox*KP, KP] -> the reference's (H+1, H+1, Ox) parameter (flat (d1, o, d2): biaffine.py:27-35)
            return dWbt.view(Ox, KP, KP)[:, :H + 1, :H + 1].permute(2, 0, 1).contiguous().view(H + 1, H + 1, Ox)

        def first_layers(name, dW12, db12, row0):
            grads[f"{name}.W1.weight"], grads[f"{name}.W1.bias"] = dW12[row0:row0 + H], db12[row0:row0 + H]
            grads[f"{name}.W2.weight"], grads[f"{name}.W2.bias"] = dW12[row0 + KP:row0 + KP + H], db12[row0 + KP:row0 + KP + H]

        for s_, (name, G) in enumerate(zip(SCORERS, Gs)):
            if G is None:
                continue
            c0 = s_ * 2 * KP
            dXF_s, dWbt_s = ops.bilinear_backward(XFa[:, c0:c0 + 2 * KP].contiguous(),
                                                  w.arc_Wbt[s_ * KP:(s_ + 1) * KP].contiguous(), G.contiguous(), H)
            dXFa[:, c0:c0 + 2 * KP] = dXF_s
            grads[f"{name}.scorer.W_bilin.weight"] = unpack_bilinear(dWbt_s, 1)
        dh_arc, dW12a, db12a = ops.linear_backward(hm, w.arc_W12, XFa, dXFa, True)
        for s_, (name, G) in enumerate(zip(SCORERS, Gs)):
            if G is not None:
                first_layers(name, dW12a, db12a, s_ * 2 * KP)
        # the label scorer: its loss reads the scores at the gold heads
        dXFl, dWbtl = ops.label_backward(XF, w.lab_Wbt, d_label, gh, lens, H, O, T)
        grads["deprel.scorer.W_bilin.weight"] = unpack_bilinear(dWbtl, O)
        dh_lab, dW12l, db12l = ops.linear_backward(hm, w.lab_W12, XF, dXFl, True)
        first_layers("deprel", dW12l, db12l, 0)
        d_hidden = (dh_arc + dh_lab).view(B, L, D)
        # dep_hid (no activation): only the live rows reach it -- the padded rows of its output are the zeros of
        # pad_packed_sequence, constants of the graph (jointmodel/model.py:182-185)
        live = (torch.arange(L, device=enc.device)[None, :] < lens[:, None]).unsqueeze(2)
        d_live = (d_hidden * live).reshape(M, D).contiguous()
        d_feat, dWd, dbd = ops.linear_backward(feat.reshape(M, D).contiguous(), w.dep_hid_w, None, d_live, False)
        grads["dep_hid.weight"], grads["dep_hid.bias"] = dWd, dbd
        # [enc ; pos_emb]: the embedding half runs on through the soft POS distribution into the POS layers
        E = enc.shape[2]
        d_x, d_emb, g_pos = self._soft_pos_backward(enc.reshape(M, E).contiguous(), dep["upos_logits"].reshape(M, -1),
                                                    w.emb_dep, d_feat[:, E:])
        grads["upos_emb_matrix_dep"] = d_emb
        grads.update(g_pos)
        return {"loss": loss, "grads": grads, "d_hidden": d_hidden, "d_feat": d_feat.view(B, L, D),
                "d_enc": (d_feat[:, :E] + d_x).view(B, L, E)}

    # ---- the same three backward passes with the model in .train(): dropout masks as INPUTS
    @staticmethod
    def _word_drop(x, m, repl):
        """WordDropout (ref models/common/dropout.py:15-27): the rows m of x [M, D] replaced by the vector repl."""
        return torch.where(m.reshape(-1, 1), repl.reshape(1, -1).expand_as(x), x).contiguous()

    @staticmethod
    def _soft_pos_train(w, x, emb, m_wd, m_drop):
        """pos_forward(use_soft_pos=True) in .train() (ref jointmodel/model.py:125-137): word dropout on the encoder rows x [M,E]
        (rows m_wd -> drop_replacement_pos), ReLU(upos_hid), dropout mask m_drop [M,Hp] (kept entries 1/(1-p)), upos_clf,
        softmax, @ emb.  w: the HeadWeights.  -> (pos_emb [M,TE], upos logits, backward(d_pos_emb) -> (d_x, d_emb, {gradients}))."""
        xp = TaskHeads._word_drop(x, m_wd, w.repl_pos)
        hid = ops.linear(xp, w.upos_hid_w, w.upos_hid_b, True, None, 0)
        hid_d = (hid * m_drop).contiguous()
        logits = ops.linear(hid_d, w.upos_clf_w, w.upos_clf_b, False, None, 0)
        P = ops.softmax_rows(logits.contiguous())
        U = P.shape[1]
        U4 = (U + 3) // 4 * 4  # (the dense-layer kernels take reduction lengths that are multiples of 4)
        pos_emb = ops.linear(torch.nn.functional.pad(P, (0, U4 - U)).contiguous(),
                             torch.nn.functional.pad(emb.t(), (0, U4 - U)).contiguous(), None, False, None, 0)

        def backward(d_pos_emb):
            d_P, d_embT, _ = ops.linear_backward(P, emb.t().contiguous(), None, d_pos_emb.contiguous(), False, need_db=False)
            d_logits = ops.softmax_rows_backward(P, d_P)
            d_hid_d, dWc, dbc = ops.linear_backward(hid_d, w.upos_clf_w, None, d_logits, False)
            d_xp, dWh, dbh = ops.linear_backward(xp, w.upos_hid_w, hid, (d_hid_d * m_drop).contiguous(), True)
            keep = (~m_wd).reshape(-1, 1).to(d_xp.dtype)
            g = {"upos_hid.weight": dWh, "upos_hid.bias": dbh, "upos_clf.weight": dWc, "upos_clf.bias": dbc,
                 "drop_replacement_pos": (d_xp * (1.0 - keep)).sum(0)}
            return d_xp * keep, d_embT.t().contiguous(), g
        return pos_emb, logits, backward

    def _parser_backward_train(self, enc, sentlens, gold_head, gold_rel, number_of_words, masks):
        """dep_forward(eval=False) of the training model in .train() (ref jointmodel/model.py:162-236) and its backward, the
        random masks given by the caller in the padded layout: masks = {"wd_pos": bool [B,L] (word dropout in front of the
        POS layers), "drop_pos": [B,L,Hp] (nn.Dropout behind ReLU(upos_hid); kept entries carry 1/(1-p)), "wd": bool [B,L]
        (word dropout on [enc ; pos_emb]), "hidden": {scorer name: (m1, m2)} each [B,L,D] (the eight drop_dep draws on
        dep_hid's output, :185-200), "scorer": {scorer name: (m1, m2)} each [B,L,H] (the scorers' own dropout behind
        ReLU(W1 .) / ReLU(W2 .), biaffine.py:82-85)}; scorer names unlabeled / deprel / linearization / distance.
        Every dense product runs on the library's kernels; the masks are elementwise products between them."""
        w = self.w
        n = w.n_arc_scorers
        lens = as_sentlens(sentlens, enc.device)
        B, L, E = enc.shape
        M, H, O, KP = B * L, w.H, w.O, ops.kp(w.H)
        D = w.dep_hid_w.shape[0]
        dev = enc.device
        f32 = lambda t: torch.as_tensor(t).to(device=dev, dtype=torch.float32)
        bl = lambda t: torch.as_tensor(t).to(device=dev, dtype=torch.bool).reshape(M)

        def with_root(t):
            t = torch.as_tensor(t).to(device=dev, dtype=torch.int32)
            if t.shape[1] == L - 1:
                t = torch.nn.functional.pad(t, (1, 0))
            return t.contiguous()

        gh, gr = with_root(gold_head), with_root(gold_rel)
        live = (torch.arange(L, device=dev)[None, :] < lens[:, None]).reshape(M, 1)
        x = enc.reshape(M, E).contiguous()
        pos_emb, _, pos_bw = TaskHeads._soft_pos_train(w, x, w.emb_dep, bl(masks["wd_pos"]), f32(masks["drop_pos"]).reshape(M, -1))
        feat = torch.cat([x, pos_emb], 1).contiguous()
        m_wd = bl(masks["wd"])
        featd = TaskHeads._word_drop(feat, m_wd, w.repl_dep)
        # dep_hid; the padded rows of its output are the zeros of pad_packed_sequence (:182-183)
        hidden = (ops.linear(featd, w.dep_hid_w, w.dep_hid_b, False, None, 0) * live).contiguous()

        def first_layer(name, W12, b12, row0):
            """A scorer's X1' | X2' [M, 2KP] from its two masked copies of `hidden`, plus what its backward needs."""
            mh, ms = masks["hidden"][name], masks["scorer"][name]
            out = torch.empty((M, 2 * KP), dtype=torch.float32, device=dev)
            saved = []
            for k in range(2):
                hin = (hidden * f32(mh[k]).reshape(M, D)).contiguous()
                Wk, bk = W12[row0 + k * KP:row0 + (k + 1) * KP].contiguous(), b12[row0 + k * KP:row0 + (k + 1) * KP].contiguous()
                mk = torch.ones((M, KP), dtype=torch.float32, device=dev)  # the ones column (appended behind the dropout) stays
                mk[:, :H] = f32(ms[k]).reshape(M, H)
                y = (ops.linear(hin, Wk, bk, True, None, 0) * mk).contiguous()
                out[:, k * KP:(k + 1) * KP] = y
                saved.append((hin, Wk, mk, y, f32(mh[k]).reshape(M, D)))
            return out, saved

        names = [nm for nm, on in zip(SCORERS, (True, w.linearization and n > 1, w.distance and n > 1)) if on]
        XFa = torch.zeros((M, n * 2 * KP), dtype=torch.float32, device=dev)
        saved_arc = {}
        for s_, nm in enumerate(SCORERS[:n]):
            if nm in names:
                XFa[:, s_ * 2 * KP:(s_ + 1) * 2 * KP], saved_arc[nm] = first_layer(nm, w.arc_W12, w.arc_b12, s_ * 2 * KP)
        use_lin, use_dist = "linearization" in names, "distance" in names
        arc_logp, raw = ops.biaffine_arc_scores_raw(XFa, B, L, w.arc_Wbt, H, n, use_lin, use_dist)
        XF, saved_lab = first_layer("deprel", w.lab_W12, w.lab_b12, 0)
        T = ops.linear(XF[:, :KP].contiguous(), w.lab_Wbt, None, False, None, 0)
        nw = float(lens.sum().item()) if number_of_words is None else float(number_of_words)
        raw_in = raw if n > 1 else None
        loss = ops.dep_loss(arc_logp, raw_in, T, XF, gh, gr, lens, H, O, use_lin, use_dist, nw)
        d_arc, d_label, d_lin, d_dist = ops.dep_loss_grad(arc_logp, raw_in, T, XF, gh, gr, lens, H, O, use_lin, use_dist, nw)
        at_head = lambda v: torch.zeros((B, L, L), dtype=torch.float32, device=dev).scatter_(2, gh.long().unsqueeze(2), v.unsqueeze(2))
        Gs = {"unlabeled": d_arc, "linearization": at_head(d_lin) if use_lin else None, "distance": at_head(d_dist) if use_dist else None}
        grads = {}

        def unpack_bilinear(dWbt, Ox):
            return dWbt.view(Ox, KP, KP)[:, :H + 1, :H + 1].permute(2, 0, 1).contiguous().view(H + 1, H + 1, Ox)

        d_hidden = torch.zeros((M, D), dtype=torch.float32, device=dev)

        def first_layer_backward(name, dXF_s, saved):
            nonlocal d_hidden
            for k, wn in enumerate(("W1", "W2")):
                hin, Wk, mk, y, mhk = saved[k]
                dY = (dXF_s[:, k * KP:(k + 1) * KP] * mk).contiguous()
                d_hin, dW, db = ops.linear_backward(hin, Wk, y, dY, True)
                grads[f"{name}.{wn}.weight"], grads[f"{name}.{wn}.bias"] = dW[:H], db[:H]
                d_hidden = d_hidden + d_hin * mhk

        for s_, nm in enumerate(SCORERS[:n]):
            if nm not in names:
                continue
            c0 = s_ * 2 * KP
            dXF_s, dWbt_s = ops.bilinear_backward(XFa[:, c0:c0 + 2 * KP].contiguous(), w.arc_Wbt[s_ * KP:(s_ + 1) * KP].contiguous(),
                                                  Gs[nm].contiguous(), H)
            grads[f"{nm}.scorer.W_bilin.weight"] = unpack_bilinear(dWbt_s, 1)
            first_layer_backward(nm, dXF_s, saved_arc[nm])
        dXFl, dWbtl = ops.label_backward(XF, w.lab_Wbt, d_label, gh, lens, H, O, T)
        grads["deprel.scorer.W_bilin.weight"] = unpack_bilinear(dWbtl, O)
        first_layer_backward("deprel", dXFl, saved_lab)
        d_live = (d_hidden * live).contiguous()
        d_featd, dWd, dbd = ops.linear_backward(featd, w.dep_hid_w, None, d_live, False)
        grads["dep_hid.weight"], grads["dep_hid.bias"] = dWd, dbd
        keep = (~m_wd).reshape(M, 1).to(torch.float32)
        grads["drop_replacement_dep"] = (d_featd * (1.0 - keep)).sum(0)
        d_feat = d_featd * keep
        d_x, d_emb, g_pos = pos_bw(d_feat[:, E:].contiguous())
        grads["upos_emb_matrix_dep"] = d_emb
        grads.update(g_pos)
        return {"loss": loss, "grads": grads, "d_hidden": d_hidden.view(B, L, D), "d_feat": d_feat.view(B, L, -1),
                "d_enc": (d_feat[:, :E] + d_x).view(B, L, E)}

    # ---- training model's pos_forward(use_soft_pos=False) after the gather, loss value only (ref jointmodel/model.py:128-135)
    def pos_loss(self, enc: torch.Tensor, sentlens, gold_upos, ignore_index=1) -> torch.Tensor:
        """-> fp32 [3] on the device: mean cross-entropy (what the reference returns as `loss`), sum, words counted."""
        tag = self.tagger(enc, sentlens)
        gold = torch.as_tensor(gold_upos).to(device=enc.device, dtype=torch.int32).contiguous()
        return ops.pos_loss(tag["upos_logits"].contiguous(), gold, tag["sentlens"], ignore_index)

    # ---- the soft POS embedding's backward, shared by the NER and parser heads: pos_emb = softmax(upos_clf(relu(
    # upos_hid(x)))) @ emb (ref jointmodel/model.py:128-135 with use_soft_pos=True, :150-151 / :171-172)
    def _soft_pos_backward(self, x: torch.Tensor, upos_logits: torch.Tensor, emb: torch.Tensor, d_pos_emb: torch.Tensor):
        """x [M,E] encoder rows, upos_logits [M,U] of those rows, emb [U,TE], d_pos_emb [M,TE] (zero on padded rows) ->
        (d_x [M,E], d_emb [U,TE], {upos_hid / upos_clf gradients})."""
        w = self.w
        hid = ops.linear(x, w.upos_hid_w, w.upos_hid_b, True, None, 0)
        P = ops.softmax_rows(upos_logits.contiguous())
        # pos_emb = P @ emb = P (emb^T)^T: a bias-free linear layer with weight emb^T [TE, U]
        d_P, d_embT, _ = ops.linear_backward(P, emb.t().contiguous(), None, d_pos_emb.contiguous(), False, need_db=False)
        d_logits = ops.softmax_rows_backward(P, d_P)
        d_hid, dWc, dbc = ops.linear_backward(hid, w.upos_clf_w, None, d_logits, False)
        d_x, dWh, dbh = ops.linear_backward(x, w.upos_hid_w, hid, d_hid, True)
        return d_x, d_embT.t().contiguous(), {"upos_hid.weight": dWh, "upos_hid.bias": dbh, "upos_clf.weight": dWc,
                                              "upos_clf.bias": dbc}

    # ---- the NER loss's backward (ref jointmodel/model.py:139-160 + crf.py:24-46 under loss.backward(), model in .eval())
    def ner_backward(self, enc: torch.Tensor, sentlens, gold_tags, masks=None):
        """-> {"loss": the batch-averaged CRF NLL (0-dim), "grads": {"ner_tag_clf.weight", "ner_tag_clf.bias", "crit_ner._transitions", "upos_emb_matrix_ner",
        "upos_hid.*", "upos_clf.*"}, "d_enc": the gradient w.r.t. the gathered encoder rows [B,L,E]}: CRF NLL averaged
        over the batch -> ReLU -> ner_tag_clf -> [enc ; soft POS embedding] -> the POS layers.  `masks` (model in .train()):
        {"wd_pos", "drop_pos"} as in _parser_backward_train and "wd": bool [B,L], the word dropout on [enc ; pos_emb] (:155)."""
        w = self.w
        if masks is not None:
            lens = as_sentlens(sentlens, enc.device)
            B, L, E = enc.shape
            M, dev = B * L, enc.device
            bl = lambda t: torch.as_tensor(t).to(device=dev, dtype=torch.bool).reshape(M)
            live = (torch.arange(L, device=dev)[None, :] < lens[:, None]).reshape(M, 1)
            x = enc.reshape(M, E).contiguous()
            pos_emb, _, pos_bw = TaskHeads._soft_pos_train(w, x, w.emb_ner, bl(masks["wd_pos"]),
                                                           torch.as_tensor(masks["drop_pos"]).to(dev, torch.float32).reshape(M, -1))
            m_wd = bl(masks["wd"])
            featd = TaskHeads._word_drop(torch.cat([x, pos_emb], 1).contiguous(), m_wd, w.repl_ner)
            logits2 = (ops.linear(featd, w.ner_w, w.ner_b, True, None, 0) * live).contiguous()
            T = logits2.shape[1]
            logits = logits2.view(B, L, T)
            gold = torch.as_tensor(gold_tags).to(device=dev, dtype=torch.int32).contiguous()
            d_logits, d_trans = ops.crf_nll_grad(logits, w.trans, None, lens, gold, 1.0 / B)
            ln, un, bi = ops.crf_log_partition(logits, w.trans, None, lens, gold)
            loss = (-(un + bi - ln)).sum() / B
            d_featd, dWn, dbn = ops.linear_backward(featd, w.ner_w, logits2, d_logits.reshape(M, T).contiguous(), True)
            keep = (~m_wd).reshape(M, 1).to(torch.float32)
            d_feat = d_featd * keep
            d_x, d_emb, g_pos = pos_bw(d_feat[:, E:].contiguous())
            grads = {"ner_tag_clf.weight": dWn, "ner_tag_clf.bias": dbn, "crit_ner._transitions": d_trans,
                     "upos_emb_matrix_ner": d_emb, "drop_replacement_ner": (d_featd * (1.0 - keep)).sum(0), **g_pos}
            return {"loss": loss, "grads": grads, "d_enc": (d_feat[:, :E] + d_x).view(B, L, E)}
        tag = self.tagger(enc, sentlens)
        lens, feat, logits = tag["sentlens"], tag["feat"], tag["ner_logits"]
        B, L, E = enc.shape
        M, D = B * L, feat.shape[2]
        gold = torch.as_tensor(gold_tags).to(device=enc.device, dtype=torch.int32).contiguous()
        d_logits, d_trans = ops.crf_nll_grad(logits.contiguous(), w.trans, None, lens, gold, 1.0 / B)
        ln, un, bi = ops.crf_log_partition(logits.contiguous(), w.trans, None, lens, gold)
        loss = (-(un + bi - ln)).sum() / B  # CRFLoss with batch_average (ref crf.py:39-42)
        T = logits.shape[2]
        d_feat, dWn, dbn = ops.linear_backward(feat.reshape(M, D).contiguous(), w.ner_w, logits.reshape(M, T).contiguous(),
                                               d_logits.reshape(M, T).contiguous(), True)
        d_x, d_emb, g_pos = self._soft_pos_backward(enc.reshape(M, E).contiguous(), tag["upos_logits"].reshape(M, -1),
                                                    w.emb_ner, d_feat[:, E:])
        grads = {"ner_tag_clf.weight": dWn, "ner_tag_clf.bias": dbn, "crit_ner._transitions": d_trans,
                 "upos_emb_matrix_ner": d_emb, **g_pos}
        return {"loss": loss, "grads": grads, "d_enc": (d_feat[:, :E] + d_x).view(B, L, E)}

    # ---- ... and its backward through upos_clf and upos_hid (what loss.backward() leaves behind pos_forward)
    def pos_backward(self, enc: torch.Tensor, sentlens, gold_upos, ignore_index=1, masks=None):
        """-> {"loss": fp32 [3] as pos_loss, "grads": {"upos_hid.weight", "upos_hid.bias", "upos_clf.weight",
        "upos_clf.bias"}, "d_enc": the gradient w.r.t. the gathered encoder rows [B,L,E] (zero on padding)}."""
        w = self.w
        lens = as_sentlens(sentlens, enc.device)
        B, L, E = enc.shape
        M = B * L
        x = enc.reshape(M, E).contiguous()
        if masks is not None:  # model in .train(): masks = {"wd_pos": bool [B,L], "drop_pos": [B,L,Hp]} (:126-129)
            dev = enc.device
            m_wd = torch.as_tensor(masks["wd_pos"]).to(device=dev, dtype=torch.bool).reshape(M)
            m_dp = torch.as_tensor(masks["drop_pos"]).to(dev, torch.float32).reshape(M, -1)
            xp = TaskHeads._word_drop(x, m_wd, w.repl_pos)
            hid = ops.linear(xp, w.upos_hid_w, w.upos_hid_b, True, None, 0)
            hid_d = (hid * m_dp).contiguous()
            logits = ops.linear(hid_d, w.upos_clf_w, w.upos_clf_b, False, None, 0).view(B, L, -1)
            gold = torch.as_tensor(gold_upos).to(device=dev, dtype=torch.int32).contiguous()
            loss = ops.pos_loss(logits, gold, lens, ignore_index)
            d_logits = ops.pos_loss_grad(logits, gold, lens, loss, ignore_index)
            U = logits.shape[2]
            d_hid_d, dWc, dbc = ops.linear_backward(hid_d, w.upos_clf_w, None, d_logits.reshape(M, U).contiguous(), False)
            d_xp, dWh, dbh = ops.linear_backward(xp, w.upos_hid_w, hid, (d_hid_d * m_dp).contiguous(), True)
            keep = (~m_wd).reshape(M, 1).to(torch.float32)
            return {"loss": loss, "d_enc": (d_xp * keep).view(B, L, E),
                    "grads": {"upos_hid.weight": dWh, "upos_hid.bias": dbh, "upos_clf.weight": dWc, "upos_clf.bias": dbc,
                              "drop_replacement_pos": (d_xp * (1.0 - keep)).sum(0)}}
        hid = ops.linear(x, w.upos_hid_w, w.upos_hid_b, True, None, 0)            # relu(upos_hid(x))
        logits = ops.linear(hid, w.upos_clf_w, w.upos_clf_b, False, None, 0).view(B, L, -1)
        gold = torch.as_tensor(gold_upos).to(device=enc.device, dtype=torch.int32).contiguous()
        loss = ops.pos_loss(logits, gold, lens, ignore_index)
        d_logits = ops.pos_loss_grad(logits, gold, lens, loss, ignore_index)       # zero on padding / ignored words
        U = logits.shape[2]
        d_hid, dWc, dbc = ops.linear_backward(hid, w.upos_clf_w, None, d_logits.reshape(M, U).contiguous(), False)
        d_x, dWh, dbh = ops.linear_backward(x, w.upos_hid_w, hid, d_hid, True)
        return {"loss": loss, "d_enc": d_x.view(B, L, E),
                "grads": {"upos_hid.weight": dWh, "upos_hid.bias": dbh, "upos_clf.weight": dWc, "upos_clf.bias": dbc}}

    # ---- decoders
    def viterbi(self, ner_logits: torch.Tensor, sentlens):
        lens = as_sentlens(sentlens, ner_logits.device)
        return ops.crf_viterbi(ner_logits.contiguous(), self.w.trans, lens)

    def mst(self, arc_logp: torch.Tensor, sentlens):
        lens = as_sentlens(sentlens, arc_logp.device)
        return ops.mst_one_root(arc_logp.contiguous(), lens)

    def crf_log_partition(self, ner_logits, pad_mask=None, sentlens=None, gold=None):
        dev = ner_logits.device
        pm = None if pad_mask is None else pad_mask.to(device=dev, dtype=torch.uint8).contiguous()
        ln = None if sentlens is None else as_sentlens(sentlens, dev)
        gd = None if gold is None else gold.to(device=dev, dtype=torch.int32).contiguous()
        return ops.crf_log_partition(ner_logits.contiguous(), self.w.trans, pm, ln, gd)

    def _pack_plan(self, enc_dep, sentlens_dep, enc_tag, sentlens_tag):
        """(live parser words, live tagger words) when this batch runs packed, else None.  Packed: engine 2 with its
        default fused / folded label path, sentence lengths known to the HOST (a list / CPU tensor: the live row count
        sizes every launch, and reading it from a device tensor would stall the stream), every sentence non-empty,
        and enough padding to pay for the gather (cfg pack_pct)."""
        cfg, w = self.cfg, self.w
        if not cfg["pack_pct"] or ops.get_gemm_engine() != 2 or w.lab_Wbt_s is None:
            return None
        if not (cfg["label_fused"] and w.arc_W12p_s is not None):
            return None
        if enc_dep.shape[1] > cfg["fused_max_len"]:
            return None

        def host(l):
            if isinstance(l, torch.Tensor):
                return None if l.is_cuda else [int(v) for v in l.tolist()]
            return [int(v) for v in l]

        ld, lt = host(sentlens_dep), host(sentlens_tag)
        if ld is None or lt is None:
            return None
        Ld, Lt = enc_dep.shape[1], enc_tag.shape[1]
        if min(ld) < 1 or min(lt) < 1 or max(ld) > Ld or max(lt) > Lt:
            return None
        md, mt = sum(ld), sum(lt)
        if 100 * md > cfg["pack_pct"] * len(ld) * Ld:
            return None
        return md, mt

    def capture(self, B: int, Ld: int, Lt: Optional[int] = None, enc_dim: int = 768) -> "CapturedStep":
        """annotate_batch at a fixed shape as a replayable HIP graph (see CapturedStep)."""
        return CapturedStep(self, B, Ld, Lt, enc_dim)

    # ---- the whole head path for one batch: what annotate() needs per batch (ref :256-287)
    def annotate_batch(self, enc_dep, sentlens_dep, enc_tag, sentlens_tag, overlap=True,
                       join=True) -> Dict[str, torch.Tensor]:
        """Returns device int32 tensors: pos [B,Lt], ner [B,Lt], heads [B,Ld], rels [B,Ld], mst_status [B].

        `join=False` (with `overlap`) is the streaming form for back-to-back batches: the tail of the batch (labels
        at the decoded heads) is enqueued on a separate finishing stream, returned as out["stream"], and the
        calling stream is NOT made to wait for the MST -- so the next batch's GEMMs start while this batch's MST is
        still decoding.  The outputs are then only valid on out["stream"] (enqueue consumers there, or wait on
        out["done"]).

        With `overlap` the three independent chains run on separate HIP streams: the tagger chain (POS head,
        NER emissions, Viterbi), the MST decode (which only needs the arc log-probs) and the label scorer's first
        layers.  The decoders are dependent chains that leave most of a CU idle (an MST workgroup holds 86 KB of LDS
        and four wavefronts, so one dense-layer block fits beside it); sharing the chip with the MFMA-bound label
        path hides them.  Joined form: the MST continues the calling stream's queue and the label layers run from a
        side stream; streaming form: the MST on a side stream, the label layers on the calling stream (DESIGN.md
        section 5).
        """
        # eager calls go straight to the functions behind the custom ops (ops.direct: no dispatcher on the way, a third of
        # the host time of a small step); under torch.compile the registered ops
        ops = _ops if torch.compiler.is_compiling() else _ops.direct
        if not overlap:
            dep = self.parser(enc_dep, sentlens_dep, want_label_argmax=False)
            tag = self.tagger(enc_tag, sentlens_tag)
            ner, _ = ops.crf_viterbi(tag["ner_logits"], self.w.trans, tag["sentlens"])
            heads, status = ops.mst_one_root(dep["arc_logp"], dep["sentlens"])
            XF, T = ops.label_project(dep["hidden_out"], self.w.lab_W12, self.w.lab_b12, self.w.lab_Wbt, self.w.H, self.w.O)
            rels, _ = ops.label_at_heads(T, XF, heads, dep["sentlens"], self.w.H, self.w.O)
            return {"pos": tag["pos_pred"], "ner": ner, "heads": heads, "rels": rels, "mst_status": status}
        w = self.w
        main = torch.cuda.current_stream()
        cfg = self.cfg
        if self._side is None:
            # all at the default HIP priority (a high-priority MST stream was tried in round 2 and removed: DESIGN.md section 9)
            self._side = (torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream())
        s_tag, s_mst, s_fin, s_lab = self._side
        pack = self._pack_plan(enc_dep, sentlens_dep, enc_tag, sentlens_tag)
        lens_d = as_sentlens(sentlens_dep, enc_dep.device)
        lens_t = as_sentlens(sentlens_tag, enc_tag.device)
        # tagger chain on its own stream; up to 32 tags: the embedding folded into ner_tag_clf's columns, upos_clf inside the tail
        t_emb, t_wne = (None, w.ner_w_tags) if w.ner_w_tags is not None and w.upos_clf_w.shape[0] <= 32 else (w.emb_ner, w.ner_w_emb)

        def run_tagger(after):
            s_tag.wait_event(after)
            for t in (enc_tag, lens_t):  # the caller may drop its batch while the side stream still reads it
                t.record_stream(s_tag)
            with torch.cuda.stream(s_tag):
                if pack:  # the tagger head on the live words only; its two outputs scattered back (zeros on padding)
                    Bt, Lt = enc_tag.shape[0], enc_tag.shape[1]
                    off_t, map_t = ops.pack_descriptor(lens_t, Lt, pack[1])
                    encp = ops.pack_rows(enc_tag.reshape(Bt * Lt, -1), map_t)
                    pos_p, ner_p, _ = ops.tagger_head(encp.view(1, pack[1], -1), off_t[Bt:], w.tag_Whn, w.tag_bhn,
                                                      w.upos_clf_w, w.upos_clf_b, t_emb, t_wne, w.ner_b)
                    tag = {"pos_pred": ops.unpack_rows(pos_p.view(-1), map_t, Bt * Lt).view(Bt, Lt),
                           "ner_logits": ops.unpack_rows(ner_p.view(pack[1], -1), map_t, Bt * Lt).view(Bt, Lt, -1)}
                else:
                    pos_pred, ner_logits, _ = ops.tagger_head(enc_tag.contiguous(), lens_t, w.tag_Whn, w.tag_bhn,
                                                              w.upos_clf_w, w.upos_clf_b, t_emb, t_wne, w.ner_b)
                    tag = {"pos_pred": pos_pred, "ner_logits": ner_logits}
                ner, _ = ops.crf_viterbi(tag["ner_logits"], w.trans, lens_t)
                tag_done = torch.cuda.Event()
                tag_done.record(s_tag)
            return tag, ner, tag_done

        split = ops.get_gemm_engine() == 2 and w.lab_Wbt_s is not None
        # The MST on the main stream's queue (joined, fused form; see below): decided here because the tagger chain's
        # starting point depends on it -- with the MST on main and eager launches the chain starts when the arc scores
        # are done; otherwise (side-stream MST, or under graph capture, where the streams become a DAG and the launch
        # order no longer decides who reaches the CUs first: 2.45 against 2.58 ms per replay) behind the label scorer's
        # W1/W2 layer (engine 2) / at arc-done (other engines).
        mst_main = join and split and cfg["label_fused"] and enc_dep.shape[1] <= cfg["fused_max_len"]
        tagger_later = split and not (mst_main and not torch.cuda.is_current_stream_capturing())
        # parser chain up to the arc log-probs on the main stream
        B, Ld, D = enc_dep.shape[0], enc_dep.shape[1], enc_dep.shape[2] + w.emb_dep.shape[1]
        rows4 = fused = False
        # dep_hid is folded into the scorers' first layers (HeadWeights.from_state_dict): the scorers read `feat`
        arcW, labW = w.arc_W12f_s, w.lab_W12f_s
        if split:  # engine 2: the large layers run on split images; the POS head's tail writes feat as one directly
            emb = w.emb_dep
            if w.arc_W12p_s is not None:
                # the embedding is folded into the scorers' weights: the "feat" image is [enc | tag probabilities]
                emb, arcW, labW = w.emb_eye, w.arc_W12p_s, w.lab_W12p_s
                D = enc_dep.shape[2] + emb.shape[1]
                if emb.shape[0] <= 32:  # upos_clf inside the POS tail (pos_clf_tail_kernel): one launch for two
                    emb = None
            # the arc scorers' two bilinear stages as ONE kernel on the image their first layer writes (csrc/arc_f16x3.hip)
            arc_fused = cfg["arc_fused"] and w.arc_W12q_s is not None and Ld <= 512 and w.arc_W12p_s is not None
            if pack:
                # Live words only from here to the labels (the reference's dense layers run on PackedSequence data,
                # annotate_model.py:155-187): every row-wise op sees ONE sentence of m_live words; the arc kernel and
                # the labels-at-heads take the packing's descriptor
                off_d, map_d = ops.pack_descriptor(lens_d, Ld, pack[0])
                one_d = off_d[B:]
                encp = ops.pack_rows(enc_dep.reshape(B * Ld, -1), map_d)
                featp, feati, _, _, _ = ops.pos_head_split(encp.view(1, pack[0], -1), one_d, w.upos_hid_w, w.upos_hid_b,
                                                           w.upos_clf_w, w.upos_clf_b, emb, False)
                if arc_fused:
                    xfp, xfi = ops.linear_split_image(featp, feati, w.arc_W12q_s[0], w.arc_W12q_s[1], w.arc_b12q, D, True,
                                                      None, 0, None, w.arc_bound[0], w.arc_bound[1], 3)
                    arc_logp = ops.arc_scores_split(xfp, xfi, B, Ld, w.H, w.n_arc_scorers, w.linearization, w.distance,
                                                    w.arc_Wbq_s[0], w.arc_Wbq_s[1], off_d, lens_d, w.arc_pad_img[0],
                                                    w.arc_pad_img[1])
                else:
                    XFa = ops.linear_split(featp, feati, arcW[0], arcW[1], w.arc_b12f, D, True, None, 0, None, 3)
                    arc_logp = ops.biaffine_arc_scores_packed(XFa, off_d, lens_d, self._arc_pad_row, B, Ld, w.arc_Wbt, w.H,
                                                              w.n_arc_scorers, w.linearization, w.distance)
            else:
                featp, feati, _, _, _ = ops.pos_head_split(enc_dep.contiguous(), lens_d, w.upos_hid_w, w.upos_hid_b,
                                                           w.upos_clf_w, w.upos_clf_b, emb, False)
                if arc_fused:
                    xfp, xfi = ops.linear_split_image(featp, feati, w.arc_W12q_s[0], w.arc_W12q_s[1], w.arc_b12q, D, True,
                                                      lens_d, Ld, w.arc_b12q_pad, w.arc_bound[0], w.arc_bound[1], 3)
                    arc_logp = ops.arc_scores_split(xfp, xfi, B, Ld, w.H, w.n_arc_scorers, w.linearization, w.distance,
                                                    w.arc_Wbq_s[0], w.arc_Wbq_s[1], None, None, None, 0.0)
                else:
                    XFa = ops.linear_split(featp, feati, arcW[0], arcW[1], w.arc_b12f, D, True, lens_d, Ld,
                                           w.arc_b12, 3)
                    arc_logp = ops.biaffine_arc_scores(XFa, B, Ld, w.arc_Wbt, w.H, w.n_arc_scorers, w.linearization,
                                                       w.distance)
        else:
            feat, _, _ = ops.pos_head(enc_dep.contiguous(), lens_d, w.upos_hid_w, w.upos_hid_b, w.upos_clf_w,
                                      w.upos_clf_b, w.emb_dep)
            arc_logp, _ = ops.deep_biaffine_arc(feat, w.arc_W12f, w.arc_b12f, w.arc_Wbt, w.H, w.n_arc_scorers,
                                                w.linearization, w.distance, False, lens_d, w.arc_b12)
        arc_done = torch.cuda.Event()
        arc_done.record(main)
        if not tagger_later:
            tag, ner, tag_done = run_tagger(arc_done)
        # MST beside the label scorer
        if mst_main:
            # The decode continues the main stream's queue: it starts a few microseconds behind the arc scores, BEFORE the
            # label layers' blocks (side stream, one event away) reach the CUs, and those then run beside it.
            s_lab.wait_event(arc_done)
            for t in (featp, feati, lens_d):
                t.record_stream(s_lab)
            heads, status = ops.mst_one_root(arc_logp, lens_d)
            mst_done, lab_stream = None, s_lab
        else:
            s_mst.wait_event(arc_done)
            arc_logp.record_stream(s_mst)
            lens_d.record_stream(s_mst)
            with torch.cuda.stream(s_mst):
                heads, status = ops.mst_one_root(arc_logp, lens_d)
                mst_done = torch.cuda.Event()
                mst_done.record(s_mst)
            lab_stream = main
        if split:
            with torch.cuda.stream(lab_stream):
                if pack:
                    XF = ops.linear_split(featp, feati, labW[0], labW[1], w.lab_b12f, D, True, None, 0, None, 2)
                else:
                    XF = ops.linear_split(featp, feati, labW[0], labW[1], w.lab_b12f, D, True, lens_d, Ld,
                                          w.lab_b12, 2)
                x1p, x1i = ops.split_rows(XF[:, :ops.kp(w.H)])
                if tagger_later or mst_main:
                    w12_done = torch.cuda.Event()
                    w12_done.record(lab_stream)
            if tagger_later:
                tag, ner, tag_done = run_tagger(w12_done)
            if mst_main:
                main.wait_event(w12_done)
                for t in (XF, x1p, x1i):
                    t.record_stream(main)
            rows4 = True  # unfused form: T in the rows-by-4 layout, X1's ones column as a per-column bias
            # Fused form (stage A after the MST with the labels-at-heads contraction in its epilogue): T (1.8 GB at the
            # headline size) is neither written nor read.  It needs the heads, so it waits for the MST -- affordable since
            # the round-based decode (csrc/mst_fast.hip); the MST runs beside the label scorer's W1/W2 layer and split.
            fused = cfg["label_fused"] and Ld <= cfg["fused_max_len"]
            if fused:
                T = x1p  # stage A runs after the MST, inside label_at_heads_fused (T never reaches memory)
            else:
                T = ops.label_stage_a_split(x1p, x1i, w.lab_Wbt_s[0], w.lab_Wbt_s[1], B, Ld, w.H, w.O, w.lab_ones_col)  # T4
        else:
            XF, T = ops.label_project(feat, w.lab_W12f, w.lab_b12f, w.lab_Wbt, w.H, w.O, lens_d, w.lab_b12)
        if join:
            fin = main
        else:
            fin = s_fin
            proj_done = torch.cuda.Event()
            proj_done.record(main)
            fin.wait_event(proj_done)
            for t in (XF, T, lens_d, lens_t):
                t.record_stream(fin)
        if mst_done is not None:
            fin.wait_event(mst_done)
        for t in (heads, status, ner, tag["pos_pred"]):
            t.record_stream(fin)
        # labels only where annotate() reads them: at the decoded heads (ref annotate_model.py:263-266)
        with torch.cuda.stream(fin):
            if split and fused and pack:
                for t in (x1i, map_d, off_d):
                    t.record_stream(fin)
                rels, _ = ops.label_at_heads_fused_packed(x1p, x1i, w.lab_Wbtc_s[0], w.lab_Wbtc_s[1], XF, heads, lens_d,
                                                          map_d, off_d, w.H, w.O)
            elif split and fused:
                x1i.record_stream(fin)
                rels, _ = ops.label_at_heads_fused(x1p, x1i, w.lab_Wbtc_s[0], w.lab_Wbtc_s[1], XF, heads, lens_d, w.H, w.O)
            elif split and rows4:
                rels, _ = ops.label_at_heads_rows4(T, XF, heads, lens_d, w.H, w.O)
            else:
                rels, _ = ops.label_at_heads(T, XF, heads, lens_d, w.H, w.O)
            # the tagger chain joins here, behind the label kernels (they do not read its results)
            fin.wait_event(tag_done)
            done = torch.cuda.Event()
            done.record(fin)
        return {"pos": tag["pos_pred"], "ner": ner, "heads": heads, "rels": rels, "mst_status": status,
                "stream": fin, "done": done, "tag_done": tag_done, "tag_stream": s_tag,
                "arc_logp": arc_logp}  # the scores the heads were decoded from (diagnostics, tests)


class CapturedStep:
    """`TaskHeads.annotate_batch` at ONE shape as a HIP graph: the step's 19 launches on three streams become one
    `hipGraphLaunch`, so a loop that calls the heads at a fixed (bucketed) shape pays one launch per batch (B=1, L=32:
    0.18 against 0.33 ms; B=32, L=64: 0.40 against 0.47; bench.py other_configs, tools/graph_probe.py).

    The graph reads the STATIC input tensors below and writes the static tensors of `out`; the caller copies a batch in
    (`enc_dep.copy_(...)`, `lens_dep.copy_(...)`, ... on the replay stream), calls `replay()`, and reads `out`
    (device int32 tensors, as annotate_batch returns them) before the next replay.  Sentence lengths are device data
    here, so ragged batches run through the graph at the padded shape (the packed form needs host-known lengths to
    size its launches).  Everything the step allocates is allocated once, at capture.
    """

    def __init__(self, heads: "TaskHeads", B: int, Ld: int, Lt: Optional[int] = None, enc_dim: int = 768,
                 device=None):
        dev = torch.device(device if device is not None else heads.w.trans.device)
        Lt = Ld - 1 if Lt is None else Lt
        if B <= 0 or Ld <= 1 or Lt <= 0:
            raise ValueError("CapturedStep: B > 0, Ld > 1 (ROOT + words), Lt > 0")
        self.heads = heads
        self.enc_dep = torch.zeros(B, Ld, enc_dim, dtype=torch.float32, device=dev)
        self.enc_tag = torch.zeros(B, Lt, enc_dim, dtype=torch.float32, device=dev)
        self.lens_dep = torch.full((B,), Ld, dtype=torch.int32, device=dev)
        self.lens_tag = torch.full((B,), Lt, dtype=torch.int32, device=dev)
        run = lambda: heads.annotate_batch(self.enc_dep, self.lens_dep, self.enc_tag, self.lens_tag)
        # warm-up on a side stream (first-call set-up of the library and of torch's allocator must not be captured)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(2):
                run()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            out = run()
        self.out = {k: out[k] for k in ("pos", "ner", "heads", "rels", "mst_status")}

    def replay(self) -> Dict[str, torch.Tensor]:
        self.graph.replay()
        return self.out
