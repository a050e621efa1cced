"""Synthetic head weights and encoder outputs shared by tests and bench.py.

Head weights: nn.Linear default init for upos_hid / dep_hid / ner_tag_clf / W1 / W2; W_bilin.weight and
upos_clf.weight ~ N(0, 0.02^2) (the reference's zero init is degenerate: every arg-max ties);
upos_emb_matrix_* ~ U(0,1) as the reference; CRF transitions ~ N(0, 0.1^2).  Everything is drawn from
a seeded CPU generator so that the GPU box and the authoring container see identical values.
"""
import math

import torch


def _linear_init(gen, out_f, in_f):
    bound = 1.0 / math.sqrt(in_f)
    w = (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound
    b = (torch.rand(out_f, generator=gen) * 2 - 1) * bound
    return w, b


def make_state_dict(E=768, TE=100, H=400, U=24, O=33, T=9, seed=1234, zero_init=False, bilin_std=0.02):
    """Returns a dict keyed by the reference's parameter names (ref annotate_model.py:34-100)."""
    g = torch.Generator().manual_seed(seed)
    D = E + TE
    sd = {}
    sd["upos_hid.weight"], sd["upos_hid.bias"] = _linear_init(g, H, E)
    sd["upos_clf.weight"] = torch.zeros(U, H) if zero_init else torch.randn(U, H, generator=g) * 0.02
    sd["upos_clf.bias"] = torch.zeros(U)
    sd["upos_emb_matrix_ner"] = torch.rand(U, TE, generator=g)
    sd["upos_emb_matrix_dep"] = torch.rand(U, TE, generator=g)
    sd["ner_tag_clf.weight"], _ = _linear_init(g, T, D)
    sd["ner_tag_clf.bias"] = torch.zeros(T)
    sd["crit_ner._transitions"] = torch.zeros(T, T) if zero_init else torch.randn(T, T, generator=g) * 0.1
    sd["dep_hid.weight"], sd["dep_hid.bias"] = _linear_init(g, D, D)
    for name, o in (("unlabeled", 1), ("deprel", O), ("linearization", 1), ("distance", 1)):
        sd[f"{name}.W1.weight"], sd[f"{name}.W1.bias"] = _linear_init(g, H, D)
        sd[f"{name}.W2.weight"], sd[f"{name}.W2.bias"] = _linear_init(g, H, D)
        sd[f"{name}.scorer.W_bilin.weight"] = (
            torch.zeros(H + 1, H + 1, o) if zero_init else torch.randn(H + 1, H + 1, o, generator=g) * bilin_std
        )
        sd[f"{name}.scorer.W_bilin.bias"] = torch.zeros(o)  # loaded by the reference, never used
    return sd


def make_sentlens(B, L, ragged, seed=1234):
    if not ragged:
        return [L] * B
    g = torch.Generator().manual_seed(seed + 17)
    lens = torch.randint(max(2, L // 4), L + 1, (B,), generator=g).tolist()
    lens[0] = L
    return sorted(lens, reverse=True)


def make_encoder_outputs(B, L, E=768, seed=1234):
    """enc_dep [B, L, E] (row 0 = ROOT token) and enc_tag [B, L-1, E]: two independent encoder passes, as in
    the reference (annotate_model.py:256-257 runs the encoder once per task with different token strings)."""
    g = torch.Generator().manual_seed(seed + 1)
    enc_dep = torch.randn(B, L, E, generator=g)
    enc_tag = torch.randn(B, max(L - 1, 1), E, generator=g)
    return enc_dep, enc_tag
