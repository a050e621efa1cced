"""BASELINE config 4 on one GPU: a corpus of distinct synthetic batches walked in the streaming form, sharded by batch.
The counts a rank layout accumulates must equal the sum of the per-batch counts, whichever way the batches are dealt
(phonlp_amd/corpus.py; the reference analogue is the batch loop of annotate_model.py:237-293 followed by its scorers)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_corpus_walk_counts_add_up():
    assert torch.cuda.is_available()
    from phonlp_amd import HeadWeights, TaskHeads, ops
    from phonlp_amd import synthetic as S
    from phonlp_amd.corpus import NER_VOCAB, SyntheticCorpus, walk
    from phonlp_amd.sharding import eval_counts, tag_tables
    dev = torch.device("cuda:0")
    E, TE, H, U, O, T = 128, 20, 48, 11, 13, 9
    heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(E, TE, H, U, O, T, seed=3), dev))
    B, L, n = 24, 40, 66 * 24 + 7  # 67 distinct batches, the last one short
    for ragged in (False, True):
        corpus = SyntheticCorpus(n, B, L, E, U, O, T, seed=99, ragged=ragged)
        assert corpus.n_batches == 67 and corpus.batch_size(66) == 7
        # reference: every batch on its own, joined form, counted by the torch restatement of the count kernel
        want = torch.zeros(8, dtype=torch.int64, device=dev)
        want_ent = torch.zeros(3, dtype=torch.int64, device=dev)
        tk, tt = (t.to(dev) for t in tag_tables(NER_VOCAB[:T]))
        for i in range(corpus.n_batches):
            bt = corpus.batch(i, dev)
            # lengths as device tensors: the padded path (the walk below hands ragged batches over as host lists: packed)
            out = heads.annotate_batch(bt["enc_dep"], bt["lens_dep"], bt["enc_tag"], bt["lens_tag"])
            torch.cuda.synchronize()
            want += eval_counts(out["pos"], out["ner"], out["heads"], out["rels"], bt["gold_pos"], bt["gold_ner"],
                                bt["gold_head"], bt["gold_rel"], bt["lens_tag"], bt["lens_dep"], out["mst_status"])
            ops.ner_entity_counts_into(want_ent, out["ner"], bt["gold_ner"], bt["lens_tag"], tk, tt)
        torch.cuda.synchronize()
        assert int(want[0]) == n and int(want[6]) == 0
        # one rank, streaming
        c1, e1, s1 = walk(heads, corpus, 0, 1, dev)
        torch.cuda.synchronize()
        assert s1 == n and c1.tolist() == want.tolist() and e1.tolist() == want_ent.tolist()
        # four ranks' shards one after the other: what the all-reduce adds up
        tot, tot_e, ns = torch.zeros_like(want), torch.zeros_like(want_ent), 0
        for r in range(4):
            c, e, s = walk(heads, corpus, r, 4, dev)
            torch.cuda.synchronize()
            tot += c
            tot_e += e
            ns += s
        assert ns == n and tot.tolist() == want.tolist() and tot_e.tolist() == want_ent.tolist()


@pytest.mark.parametrize("ragged", [False, True], ids=["full", "ragged-packed"])
def test_corpus_walk_full_head_sizes(ragged):
    """The same property at the sizes BASELINE config 4 names (PhoBERT-base heads, batches of 256 sentences of 128
    words): 9 distinct batches, the last one short; per batch the joined padded step is compared ID FOR ID with the C
    oracle's decoders on the GPU's own scores, and the streaming walk (ragged: the packed path) over 1 and 2 rank
    layouts must tally exactly the counts of those per-batch outputs."""
    import numpy as np
    from oracle import c_oracle as C
    from phonlp_amd import HeadWeights, TaskHeads, ops
    from phonlp_amd import synthetic as S
    from phonlp_amd.corpus import NER_VOCAB, SyntheticCorpus, walk
    from phonlp_amd.sharding import eval_counts, tag_tables
    dev = torch.device("cuda:0")
    E, TE, H, U, O, T = 768, 100, 400, 24, 33, 9
    sd = S.make_state_dict(E, TE, H, U, O, T, seed=1234)
    heads = TaskHeads(HeadWeights.from_state_dict(sd, dev))
    B, L, n = 256, 128, 8 * 256 + 40
    corpus = SyntheticCorpus(n, B, L, E, U, O, T, seed=4242, ragged=ragged)
    assert corpus.n_batches == 9 and corpus.batch_size(8) == 40
    want = torch.zeros(8, dtype=torch.int64, device=dev)
    want_ent = torch.zeros(3, dtype=torch.int64, device=dev)
    tk, tt = (t.to(dev) for t in tag_tables(NER_VOCAB[:T]))
    trans = sd["crit_ner._transitions"].numpy()
    for i in range(corpus.n_batches):
        bt = corpus.batch(i, dev)
        out = heads.annotate_batch(bt["enc_dep"], bt["lens_dep"], bt["enc_tag"], bt["lens_tag"])
        torch.cuda.synchronize()
        if i in (0, 8):  # decoders of two batches against the C oracle on the GPU's own scores
            dep = heads.parser(bt["enc_dep"], bt["lens_dep"], want_label_argmax=False)
            tag = heads.tagger(bt["enc_tag"], bt["lens_tag"])
            ld, lt = bt["lens_dep"].cpu().tolist(), bt["lens_tag"].cpu().tolist()
            wh, ws = C.mst_batch(dep["arc_logp"].cpu().numpy(), ld)
            assert not ws.any() and np.array_equal(out["heads"].cpu().numpy(), wh)
            wt, _ = C.viterbi_batch(tag["ner_logits"].cpu().numpy(), trans, lt)
            assert np.array_equal(out["ner"].cpu().numpy(), wt)
        want += eval_counts(out["pos"], out["ner"], out["heads"], out["rels"], bt["gold_pos"], bt["gold_ner"],
                            bt["gold_head"], bt["gold_rel"], bt["lens_tag"], bt["lens_dep"], out["mst_status"])
        ops.ner_entity_counts_into(want_ent, out["ner"], bt["gold_ner"], bt["lens_tag"], tk, tt)
    torch.cuda.synchronize()
    assert int(want[0]) == n and int(want[6]) == 0
    c1, e1, s1 = walk(heads, corpus, 0, 1, dev)
    torch.cuda.synchronize()
    assert s1 == n and c1.tolist() == want.tolist() and e1.tolist() == want_ent.tolist()
    tot, tot_e, ns = torch.zeros_like(want), torch.zeros_like(want_ent), 0
    for r in range(2):
        c, e, s = walk(heads, corpus, r, 2, dev)
        torch.cuda.synchronize()
        tot += c
        tot_e += e
        ns += s
    assert ns == n and tot.tolist() == want.tolist() and tot_e.tolist() == want_ent.tolist()
