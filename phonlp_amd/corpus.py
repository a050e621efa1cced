"""BASELINE config 4: a corpus of synthetic sentences walked in batches, sharded by batch over the ranks.

The reference walks its corpus one batch after the other (annotate_model.py:237-293: per batch the two forwards, the
decoders, then the outputs are appended); its evaluation scripts then count agreements with the gold annotation over
the whole corpus (models/pos/scorer.py:29-36, models/ner/scorer.py:13-57, utils/conll18_ud_eval.py:531-532).  Here a
rank takes batches rank, rank + world, ... (sharding.shard_batches), every batch goes through TaskHeads.annotate_batch
in its streaming form (the next batch's GEMMs start while this batch's MST is still decoding) and the counts ADD into
device vectors; the only collective is the caller's one all-reduce of those vectors at the end.

Batches are generated ON THE DEVICE from a per-batch seed (seed + batch index), so a corpus of any size needs no
storage, every batch is distinct, and any rank layout sees the same sentences: the counts of a sharded walk must add
up to the counts of the single-rank walk (tests/test_gpu_corpus.py).
"""
from typing import Dict, Optional

import torch

from . import ops
from .sharding import shard_batches, tag_tables

NER_VOCAB = ["<PAD>", "<UNK>", "<EMPTY>", "<ROOT>", "O", "B-PER", "I-PER", "B-LOC", "I-LOC", "B-ORG", "I-ORG", "B-MISC",
             "I-MISC", "E-PER", "S-PER", "E-LOC"]


class SyntheticCorpus:
    """n_sentences sentences of L words (parser length, ROOT included) in batches of B; the last batch may be short."""

    def __init__(self, n_sentences: int, B: int, L: int, E: int, U: int, O: int, T: int, seed: int = 1234,
                 ragged: bool = False):
        self.n, self.B, self.L, self.E, self.U, self.O, self.T = n_sentences, B, L, E, U, O, T
        self.seed, self.ragged = seed, ragged
        self.n_batches = (n_sentences + B - 1) // B

    def batch_size(self, i: int) -> int:
        return min(self.B, self.n - i * self.B)

    def batch(self, i: int, device) -> Dict[str, torch.Tensor]:
        """Encoder rows, sentence lengths and gold annotation of batch i, drawn on `device` from seed + i."""
        b, L = self.batch_size(i), self.L
        g = torch.Generator(device=device).manual_seed(self.seed + i)
        out = {
            "enc_dep": torch.randn(b, L, self.E, generator=g, device=device),
            "enc_tag": torch.randn(b, L - 1, self.E, generator=g, device=device),
        }
        if self.ragged:
            # drawn on the HOST (a few hundred integers): the reference's loader knows every sentence's length on the
            # host too (annotate_model.py:440-467), and host-known lengths are what lets annotate_batch pack the live words
            gh = torch.Generator().manual_seed(self.seed + i)
            lens_h = torch.sort(torch.randint(max(2, L // 4), L + 1, (b,), generator=gh, dtype=torch.int32),
                                descending=True)[0]
            lens = lens_h.to(device)
            out["lens_dep_host"], out["lens_tag_host"] = lens_h.tolist(), (lens_h - 1).tolist()
        else:
            lens = torch.full((b,), L, dtype=torch.int32, device=device)
        out["lens_dep"], out["lens_tag"] = lens, lens - 1
        ri = lambda hi, w: torch.randint(0, hi, (b, w), generator=g, device=device, dtype=torch.int32)
        out["gold_pos"], out["gold_ner"] = ri(self.U, L - 1), ri(self.T, L - 1)
        out["gold_head"], out["gold_rel"] = ri(L, L), ri(self.O, L)
        return out


def walk(heads, corpus: SyntheticCorpus, rank: int = 0, world: int = 1, device=None, streaming: bool = True,
         counts: Optional[torch.Tensor] = None, counts_ent: Optional[torch.Tensor] = None, first: int = 0,
         last: Optional[int] = None):
    """This rank's share of the corpus (batches first + rank, first + rank + world, ... below `last`) through the head
    path.  Returns (counts int64[8], counts_ent int64[3], sentences walked); the count vectors are on the device, ADDED
    to (pass them in to keep accumulating), and complete once the device is synchronised."""
    dev = device or torch.device("cuda", torch.cuda.current_device())
    if counts is None:
        counts = torch.zeros(8, dtype=torch.int64, device=dev)
    if counts_ent is None:
        counts_ent = torch.zeros(3, dtype=torch.int64, device=dev)
    tag_kind, tag_type = (t.to(dev) for t in tag_tables(NER_VOCAB[:corpus.T]))
    last = corpus.n_batches if last is None else min(last, corpus.n_batches)
    s_cnt = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    sentences = 0
    for i in shard_batches(last - first, rank, world):
        bt = corpus.batch(first + i, dev)
        # ragged batches hand over their lengths as host lists (the packed path needs them to size its launches)
        out = heads.annotate_batch(bt["enc_dep"], bt.get("lens_dep_host", bt["lens_dep"]), bt["enc_tag"],
                                   bt.get("lens_tag_host", bt["lens_tag"]), join=not streaming)
        with torch.cuda.stream(out["tag_stream"]):
            for k in ("gold_ner", "lens_tag"):
                bt[k].record_stream(out["tag_stream"])
            ops.ner_entity_counts_into(counts_ent, out["ner"], bt["gold_ner"], bt["lens_tag"], tag_kind, tag_type)
        s_cnt.wait_event(out["done"])
        s_cnt.wait_event(out["tag_done"])
        with torch.cuda.stream(s_cnt):
            for k in ("pos", "ner", "heads", "rels", "mst_status"):
                out[k].record_stream(s_cnt)
            for k in ("gold_pos", "gold_ner", "gold_head", "gold_rel", "lens_tag", "lens_dep"):
                bt[k].record_stream(s_cnt)
            ops.eval_counts_into(counts, out["pos"], out["ner"], out["heads"], out["rels"], bt["gold_pos"],
                                 bt["gold_ner"], bt["gold_head"], bt["gold_rel"], bt["lens_tag"], bt["lens_dep"],
                                 out["mst_status"])
        sentences += corpus.batch_size(first + i)
    main.wait_stream(s_cnt)
    if heads._side is not None:
        for s in heads._side:
            main.wait_stream(s)
    return counts, counts_ent, sentences
