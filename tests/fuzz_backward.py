"""On the GPU box: TaskHeads.parser_backward at random shapes, head sizes, scorer flags and matrix engines against torch
autograd through the oracle's dep_heads + dep_loss (which the reference's own .grad fixtures pin).  A checker, run by hand
(pytest does not collect it).  usage: python tests/fuzz_backward.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import heads_oracle as O  # noqa: E402
from phonlp_amd import HeadWeights, TaskHeads, ops  # noqa: E402
from phonlp_amd import synthetic as S  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
worst, bad = 0.0, 0
for case in range(n_cases):
    E = int(rng.choice([16, 24, 40, 64]))
    TE = int(rng.choice([4, 8, 12]))
    H = int(rng.choice([8, 16, 24, 40]))
    U = int(rng.integers(2, 33))
    Orel = int(rng.integers(2, 9))
    B = int(rng.integers(1, 13))
    L = int(rng.integers(3, 70))
    engine = int(rng.integers(0, 3))
    lin, dist = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    ops.set_gemm_engine(engine)
    sd = S.make_state_dict(E, TE, H, U, Orel, 5, seed=3000 + case, bilin_std=0.2)
    enc_dep, _ = S.make_encoder_outputs(B, L, E, seed=4000 + case)
    lens1 = sorted((int(x) for x in rng.integers(2, L + 1, B)), reverse=True)
    lens1[0] = L
    g = torch.Generator().manual_seed(5000 + case)
    head = torch.zeros(B, L, dtype=torch.int64)
    rel = torch.zeros(B, L, dtype=torch.int64)
    for b, ln in enumerate(lens1):
        h = torch.randint(0, ln - 1, (ln - 1,), generator=g)
        i = torch.arange(1, ln)
        head[b, 1:ln] = torch.where(h < i, h, h + 1)
        rel[b, 1:ln] = torch.randint(0, Orel, (ln - 1,), generator=g)
    nw = sum(lens1)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    enc_leaf = enc_dep.clone().requires_grad_(True)
    dep = O.dep_heads(enc_leaf, lens1, p, linearization=lin, distance=dist)
    total, _ = O.dep_loss(dep, lens1, head[:, 1:].numpy(), rel[:, 1:].numpy(), nw, linearization=lin, distance=dist)
    total.backward()
    heads = TaskHeads(HeadWeights.from_state_dict(sd, dev, linearization=lin, distance=dist))
    out = heads.parser_backward(enc_dep.to(dev), lens1, head, rel, nw)
    torch.cuda.synchronize()
    errs, scales = {}, {}
    for name, got in out["grads"].items():
        want = p[name].grad
        errs[name] = float((got.cpu() - want).abs().max()) / (float(want.abs().max()) + 1e-12)
    errs["d_enc"] = float((out["d_enc"].cpu() - enc_leaf.grad).abs().max()) / (float(enc_leaf.grad.abs().max()) + 1e-12)
    errs["loss"] = abs(float(out["loss"][4]) - float(total)) / abs(float(total))
    # ---- the NER loss (CRF NLL -> ReLU -> ner_tag_clf -> [enc ; soft POS embedding] -> POS layers) and the POS loss
    T = 5
    Lt = L - 1
    if Lt >= 1:
        _, enc_tag = S.make_encoder_outputs(B, L, E, seed=6000 + case)
        lens = [l - 1 for l in lens1]
        tags = torch.randint(0, T, (B, Lt), generator=g)
        upos = torch.randint(0, U, (B, Lt), generator=g)
        masks_ = torch.arange(Lt).view(1, Lt) >= torch.tensor(lens).view(B, 1)
        p2 = {k_: v.clone().requires_grad_(True) for k_, v in sd.items()}
        leaf2 = enc_tag.clone().requires_grad_(True)
        ot = O.tagger_heads(leaf2, lens, p2)
        nll = O.crf_nll(ot["ner_logits"], masks_, tags, p2["crit_ner._transitions"])
        nll.backward()
        outn = heads.ner_backward(enc_tag.to(dev), lens, tags)
        for name, got in outn["grads"].items():
            want = p2[name].grad
            errs["ner:" + name] = float((got.cpu() - want).abs().max()) / (float(want.abs().max()) + 1e-12)
            scales["ner:" + name] = float(want.abs().max())
        errs["ner:d_enc"] = float((outn["d_enc"].cpu() - leaf2.grad).abs().max()) / (float(leaf2.grad.abs().max()) + 1e-12)
        errs["ner:loss"] = abs(float(outn["loss"]) - float(nll)) / max(abs(float(nll)), 1e-12)
        p3 = {k_: v.clone().requires_grad_(True) for k_, v in sd.items()}
        leaf3 = enc_tag.clone().requires_grad_(True)
        pl = O.pos_loss(O.tagger_heads(leaf3, lens, p3)["upos_logits"], lens, upos)
        if torch.isfinite(pl):  # (every gold tag the ignored id: nothing to compare)
            pl.backward()
            outp = heads.pos_backward(enc_tag.to(dev), lens, upos)
            for name, got in outp["grads"].items():
                want = p3[name].grad
                errs["pos:" + name] = float((got.cpu() - want).abs().max()) / (float(want.abs().max()) + 1e-12)
            errs["pos:d_enc"] = float((outp["d_enc"].cpu() - leaf3.grad).abs().max()) / (float(leaf3.grad.abs().max()) + 1e-12)
            errs["pos:loss"] = abs(float(outp["loss"][0]) - float(pl)) / max(abs(float(pl)), 1e-12)
    # a gradient that is all cancellation (largest entry ~1e-6 of the others') is compared on the scale of fp32 noise instead
    for k_ in list(errs):
        if k_ in scales and errs[k_] * scales[k_] <= 1e-6:
            errs[k_] = min(errs[k_], 1e-6)
    k, e = max(errs.items(), key=lambda kv: kv[1])
    worst = max(worst, e)
    flag = "" if e <= 5e-4 else "  <-- look"
    bad += e > 5e-4
    print(f"case {case:3d} eng={engine} lin={int(lin)} dist={int(dist)} E={E} TE={TE} H={H} U={U} O={Orel} B={B} L={L}: "
          f"{len(errs)} quantities, worst relative error {e:.2e} ({k}, largest entry {scales.get(k, float('nan')):.2e}){flag}", flush=True)
print(f"cases with an error above 5e-4: {bad}; worst {worst:.2e}")
