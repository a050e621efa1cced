"""On the GPU box: annotate_batch at random shapes and head sizes against the oracle chain end to end (POS ids, NER tags,
heads, relations), plus the decoders against the C oracle on the GPU's own arc scores.  Prints one line per case and the
total of differing ids; cases with a difference are re-checked for near-ties (the oracle's own margin at that word).
usage: python tests/fuzz_step.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import heads_oracle as O  # noqa: E402  (a checker: this file lives under tests/ and is run by hand, pytest does not collect it)
from oracle import c_oracle as C  # noqa: E402
from phonlp_amd import HeadWeights, TaskHeads, ops  # noqa: E402
from phonlp_amd import synthetic as S  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
# FUZZ_MODE: letters r = the reference's head sizes (768 / 100 / 400 / 24 / 33 / 9), l = every fourth case a long sentence batch (129..520 words), e = a random matrix engine per case,
# f = random linearization / distance flags, s = every other case in the streaming form (join=False), g = every third
# case also captured as a HIP graph (TaskHeads.capture) and replayed on this and on a second batch: same ids as eager
MODE = os.environ.get("FUZZ_MODE", "")
LONG, ENGINES, FLAGS, STREAM, GRAPH = "l" in MODE, "e" in MODE, "f" in MODE, "s" in MODE, "g" in MODE
bad = 0
for case in range(n_cases):
    E = int(rng.choice([16, 24, 40, 64]))
    TE = int(rng.choice([4, 8, 12]))
    H = int(rng.choice([8, 16, 24, 40]))
    U = int(rng.integers(1, 33))
    Orel = int(rng.integers(2, 9))
    T = int(rng.integers(2, 13))
    B = int(rng.integers(1, 41))
    L = int(rng.integers(2, 141))
    ragged = bool(rng.integers(0, 2))
    if "r" in MODE:  # the reference's head sizes
        E, TE, H, U, Orel, T = 768, 100, 400, 24, 33, 9
        B, L = int(rng.integers(1, 25)), int(rng.integers(2, 101))
    if LONG and case % 4 == 0:  # long sentences: the arc kernel's LONG form, the round-based decode in global memory, unfused labels
        B, L = int(rng.integers(1, 7)), int(rng.integers(129, 521))
    engine = int(rng.integers(0, 3)) if ENGINES else 2
    lin, dist = (bool(rng.integers(0, 2)), bool(rng.integers(0, 2))) if FLAGS else (True, True)
    ops.set_gemm_engine(engine)
    sd = S.make_state_dict(E, TE, H, U, Orel, T, seed=1000 + case)
    enc_dep, enc_tag = S.make_encoder_outputs(B, L, E, seed=2000 + case)
    lens1 = sorted((int(x) for x in rng.integers(2, L + 1, B)), reverse=True) if ragged else [L] * B
    lens1[0] = L
    lens = [l - 1 for l in lens1]
    heads = TaskHeads(HeadWeights.from_state_dict(sd, dev, linearization=lin, distance=dist))
    join = not (STREAM and case % 2)
    out = heads.annotate_batch(enc_dep.to(dev), lens1 if case % 3 else torch.tensor(lens1, dtype=torch.int32, device=dev),
                               enc_tag.to(dev), lens if case % 3 else torch.tensor(lens, dtype=torch.int32, device=dev), join=join)
    torch.cuda.synchronize()
    graph_ok = True
    if GRAPH and case % 3 == 0:
        cs = heads.capture(B, L, enc_dim=E)
        enc2_dep, enc2_tag = S.make_encoder_outputs(B, L, E, seed=7000 + case)
        for ed_, et_ in ((enc_dep, enc_tag), (enc2_dep, enc2_tag)):
            cs.enc_dep.copy_(ed_); cs.enc_tag.copy_(et_)
            cs.lens_dep.copy_(torch.tensor(lens1, dtype=torch.int32)); cs.lens_tag.copy_(torch.tensor(lens, dtype=torch.int32))
            gout = {k_: v_.clone() for k_, v_ in cs.replay().items()}
            torch.cuda.synchronize()
            eo = heads.annotate_batch(ed_.to(dev), torch.tensor(lens1, dtype=torch.int32, device=dev), et_.to(dev),
                                      torch.tensor(lens, dtype=torch.int32, device=dev))
            torch.cuda.synchronize()
            graph_ok = graph_ok and all(torch.equal(gout[k_], eo[k_]) for k_ in ("pos", "ner", "heads", "rels", "mst_status"))
        del cs
    od = O.dep_heads(enc_dep, lens1, sd, linearization=lin, distance=dist)
    ot = O.tagger_heads(enc_tag, lens, sd)
    wh, ws = C.mst_batch(od["arc_logp"].numpy(), lens1)
    gh, gs = C.mst_batch(out["arc_logp"].cpu().numpy(), lens1)  # the decoder alone, on the GPU's own scores: bit-exact
    dec_ok = np.array_equal(out["heads"].cpu().numpy(), gh) and np.array_equal(out["mst_status"].cpu().numpy(), gs)
    rel = np.zeros((B, L), np.int64)
    for b in range(B):
        for i in range(1, lens1[b]):
            rel[b, i] = int(od["label_argmax"][b, i, wh[b, i]])
    wt, _ = C.viterbi_batch(ot["ner_logits"].numpy(), sd["crit_ner._transitions"].numpy(), lens)
    d_heads = int((out["heads"].cpu().numpy() != wh).sum())
    d_rels = int((out["rels"].cpu().numpy() != rel).sum())
    d_ner = int((out["ner"].cpu().numpy() != wt).sum())
    if d_ner:  # ReLU'd emissions tie often: a differing path that scores the oracle's maximum under the ORACLE's emissions is a tie
        em, tr_ = ot["ner_logits"].numpy().astype(np.float64), sd["crit_ner._transitions"].numpy().astype(np.float64)
        gt = out["ner"].cpu().numpy()
        for b in range(B):
            if (gt[b] != wt[b]).any():
                sc = lambda t: sum(em[b, i, t[i]] for i in range(lens[b])) + sum(tr_[t[i - 1], t[i]] for i in range(1, lens[b]))
                if abs(sc(gt[b]) - sc(wt[b])) <= 1e-5 * max(1.0, abs(sc(wt[b]))):
                    d_ner -= int((gt[b] != wt[b]).sum())
    d_pos = int((out["pos"].cpu().numpy() != ot["pos_pred"].numpy()).sum())
    # the reference-shaped / two-launch forms of the same heads (an embedding matrix passed in): scores against the oracle
    w = heads.w
    ld_t = torch.tensor(lens, dtype=torch.int32, device=dev)
    tg = heads.tagger(enc_tag.to(dev), lens)
    _, ner2, log2 = ops.tagger_head(enc_tag.to(dev), ld_t, w.tag_Whn, w.tag_bhn, w.upos_clf_w, w.upos_clf_b, w.emb_ner,
                                    w.ner_w_emb, w.ner_b)
    dp = heads.parser(enc_dep.to(dev), lens1, want_label_scores=True)  # the dense label tensor (B x L x O x L in memory)
    live_t = torch.arange(L - 1)[None, :] < torch.tensor(lens)[:, None]
    live_d = torch.arange(L)[None, :] < torch.tensor(lens1)[:, None]
    e_ner = max(float((tg["ner_logits"].cpu() - ot["ner_logits"])[live_t].abs().max()) if live_t.any() else 0.0,
                float((ner2.cpu() - ot["ner_logits"])[live_t].abs().max()) if live_t.any() else 0.0)
    fin = torch.isfinite(od["arc_logp"]) & live_d[:, :, None]
    e_arc2 = float((dp["arc_logp"].cpu() - od["arc_logp"])[fin].abs().max())
    live_c = (live_d[:, :, None] & live_d[:, None, :])  # live (dependent, head) cells
    ls_g, ls_o = dp["label_scores"].permute(0, 1, 3, 2).cpu(), od["label_scores"]
    e_lab = float((ls_g - ls_o)[live_c].abs().max()) / (float(ls_o[live_c].abs().max()) + 1e-12)
    am_g, am_o = dp["label_argmax"].cpu(), od["label_argmax"]
    top = ls_o.max(-1)[0]
    at = torch.gather(ls_o, -1, am_g.long().unsqueeze(-1)).squeeze(-1)
    lab_flip_bad = int((((am_g != am_o) & live_c) & ((top - at) > 1e-4 * top.abs().clamp_min(1.0))).sum())
    scores_ok = e_ner < 2e-5 and e_arc2 < 2e-3 and e_lab < 1e-3 and lab_flip_bad == 0
    arc_err = float((out["arc_logp"].cpu() - od["arc_logp"])[fin].abs().max())  # (padded dependents are not scored)
    flag = "" if dec_ok and scores_ok and graph_ok and not (d_heads or d_rels or d_ner or d_pos) else "  <-- look"
    bad += (not dec_ok) + (not scores_ok) + (not graph_ok) + d_heads + d_rels + d_ner + d_pos
    print(f"case {case:3d} eng={engine} lin={int(lin)} dist={int(dist)} join={int(join)} E={E} TE={TE} H={H} U={U} O={Orel} T={T} B={B} L={L} ragged={int(ragged)}: decoder_exact={dec_ok} graph_equals_eager={graph_ok} "
          f"diff heads {d_heads} rels {d_rels} ner {d_ner} pos {d_pos}; max |arc_logp - oracle| {arc_err:.2e}; two-launch forms: ner {e_ner:.1e} arc {e_arc2:.1e} dense labels {e_lab:.1e} (arg-max off a near-tie: {lab_flip_bad}){flag}", flush=True)
print("total differing ids (and inexact decodes):", bad)
