"""Sentence-level sharding and the eval-count vector (the only cross-GPU quantity).

Batches of length-sorted sentences are dealt round-robin to ranks, which keeps the length distribution of every
rank equal; weights are replicated; no collective touches the data path.  The counts mirror what the
reference's scorers accumulate: POS accuracy (ref models/pos/scorer.py:29-36), UAS/LAS
(utils/conll18_ud_eval.py:531-532) and, for NER, tag-level agreement plus the entity-level counts of
models/ner/scorer.py (ops.ner_entity_counts; tag_tables() turns the tag vocabulary into the two small integer
tables the kernel replays the BIOES decoder with).
"""
import torch

COUNT_NAMES = ("sentences", "words", "pos_correct", "ner_correct", "uas_correct", "las_correct", "mst_failures", "_")


TAG_KINDS = {"B-": 1, "I-": 2, "E-": 3, "S-": 4}


def tag_tables(tag_strings):
    """Tag vocabulary (id -> string) -> (kind, type) int32 tensors for ops.ner_entity_counts: kind 0 'O', 1-4 for the
    B-/I-/E-/S- prefixes, 5 for strings ref ner/utils.py:85-128 decode_from_bioes ignores (<PAD>, <UNK>, ...);
    type = dense id of the string after the prefix (-1 where there is none)."""
    kinds, types, ids = [], [], {}
    for t in tag_strings:
        if t is None or t == "O":
            kinds.append(0); types.append(-1)
        elif t[:2] in TAG_KINDS:
            kinds.append(TAG_KINDS[t[:2]]); types.append(ids.setdefault(t[2:], len(ids)))
        else:
            kinds.append(5); types.append(-1)
    return torch.tensor(kinds, dtype=torch.int32), torch.tensor(types, dtype=torch.int32)


def prf(correct: int, guessed: int, gold: int):
    """Micro precision / recall / F1 exactly as ref models/ner/scorer.py:52-60 derives them from the counts."""
    p = correct / guessed if guessed > 0 else 0.0
    r = correct / gold if gold > 0 else 0.0
    return p, r, (2.0 * p * r / (p + r) if p + r > 0 else 0.0)


def shard_batches(n_batches: int, rank: int, world: int):
    return range(rank, n_batches, world)


def eval_counts(pos, ner, heads, rels, gold_pos, gold_ner, gold_head, gold_rel, lens_tag, lens_dep, mst_status=None):
    """All [B, L] int tensors (tagger tensors may be one column shorter than parser ones); lens_* int [B].
    Parser rows start with ROOT (column 0 is skipped).  Returns int64[8] on the inputs' device."""
    dev = pos.device
    Lt, Ld = pos.shape[1], heads.shape[1]
    mt = torch.arange(Lt, device=dev).unsqueeze(0) < lens_tag.to(dev).long().unsqueeze(1)
    col = torch.arange(Ld, device=dev).unsqueeze(0)
    md = (col >= 1) & (col < lens_dep.to(dev).long().unsqueeze(1))
    uas = (heads == gold_head) & md
    out = torch.zeros(8, dtype=torch.int64, device=dev)
    out[0] = pos.shape[0]
    out[1] = mt.sum()
    out[2] = ((pos == gold_pos) & mt).sum()
    out[3] = ((ner == gold_ner) & mt).sum()
    out[4] = uas.sum()
    out[5] = (uas & (rels == gold_rel)).sum()
    if mst_status is not None:
        out[6] = (mst_status != 0).sum()
    return out
