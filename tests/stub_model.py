"""A stand-in for the reference's `annotate_model.JointModel`, for testing `phonlp_amd.compat.patch` on a box
where /root/reference does not exist.  It owns an encoder (`.phobert`), the head parameters under the
reference's names, toy vocab/tokenizer and the host-side batching helpers with the reference's semantics
(annotate_model.py:382-467: [ROOT] prefix on the parser path, descending-length sort, pad ids) -- written
here from that description, not copied."""
import numpy as np
import torch
import torch.nn as nn


class ToyVocab:
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
    def __init__(self, vocab_size):
        self.vocab_size = vocab_size

    def encode(self, w):
        h = sum((k + 1) * ord(ch) for k, ch in enumerate(w))
        pieces = [3 + h % (self.vocab_size - 3)]
        if len(w) > 6:
            pieces.append(3 + (h * 7 + 1) % (self.vocab_size - 3))
        return [0] + pieces + [2]


def _sort_desc(fields, lens):
    order = sorted(range(len(lens)), key=lambda i: (lens[i], i), reverse=True)
    return [[f[i] for i in order] for f in fields], order


def _pad(rows, pad_id=0):
    out = torch.full((len(rows), max(len(r) for r in rows)), pad_id, dtype=torch.long)
    for i, r in enumerate(rows):
        out[i, :len(r)] = torch.tensor(r, dtype=torch.long)
    return out


class StubJointModel(nn.Module):
    def __init__(self, z, hidden=32, n_upos=6, n_rel=7, n_ner=5):
        super().__init__()
        from transformers import AutoModel, RobertaConfig
        cfg = RobertaConfig(vocab_size=200, hidden_size=hidden, num_hidden_layers=2, num_attention_heads=4,
                            intermediate_size=64, max_position_embeddings=258, pad_token_id=1,
                            output_hidden_states=True)
        self.phobert = AutoModel.from_config(cfg)
        enc = {k[len("enc_phobert."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("enc_phobert.")}
        missing, unexpected = self.phobert.load_state_dict(enc, strict=False)
        assert not [m for m in missing if "position_ids" not in m], missing
        self._heads = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p_")}
        self.args = {"linearization": True, "distance": True}
        self.vocab = {
            "upos": ToyVocab([f"P{k}" for k in range(n_upos - 4)]),
            "deprel": ToyVocab([f"rel{k}" for k in range(n_rel - 4)]),
            "ner_tag": ToyVocab([f"N{k}" for k in range(n_ner - 4)]),
            "word": ToyVocab([]), "char": ToyVocab([]),
        }
        self.tokenizer = ToyTokenizer(200)
        self.eval()

    def state_dict(self, *a, **k):
        sd = {"phobert." + n: v for n, v in self.phobert.state_dict().items()}
        sd.update(self._heads)
        return sd

    def _process(self, batch_text, with_root):
        out = []
        for sent in batch_text:
            ids, first = [0], []
            words = ([("[ROOT]", 3)] if with_root else []) + [(w, None) for w in sent]
            for w, _ in words:
                first.append(len(ids))
                ids += self.tokenizer.encode(w)[1:-1]
            ids.append(2)
            wid = ([3] if with_root else []) + self.vocab["word"].map(sent)
            chars = ([[3]] if with_root else []) + [self.vocab["char"].map(list(w)) for w in sent]
            out.append([ids, first, wid, chars])
        return out

    def process_data_tagger(self, batch_text):
        return self._process(batch_text, False)

    def process_data_parser(self, batch_text):
        return self._process(batch_text, True)

    def chunk_batches(self, data, batch_size):
        (data,), self.data_orig_idx = _sort_desc([data], [len(x[2]) for x in data])
        return [data[i:i + batch_size] for i in range(0, len(data), batch_size)]

    def get_batch(self, key, chunks):
        batch = list(zip(*chunks[key]))
        fields, orig_idx = _sort_desc(batch, [len(x) for x in batch[2]])
        tokens = _pad(fields[0], pad_id=1)
        first = _pad(fields[1])
        words = _pad(fields[2])
        return tokens, first, words.eq(0), sum(len(s) for s in fields[3]), orig_idx, [len(x) for x in fields[1]]


class TableEncoder(nn.Module):
    """Encoder stand-in for the full-size fixture (BASELINE config 1): last hidden state = token table row + position
    table row, both drawn from a seeded CPU generator -- the same rows in the authoring container (plugged into the
    reference's JointModel by tests/golden/make_golden.py) and on the GPU box, so the fixture stores no encoder."""

    def __init__(self, vocab_size=2000, hidden=768, max_pos=512, seed=4321):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.tok = nn.Parameter(torch.randn(vocab_size, hidden, generator=g), requires_grad=False)
        self.pos = nn.Parameter(torch.randn(max_pos, hidden, generator=g) * 0.5, requires_grad=False)

    def forward(self, tokens, *a, **k):
        h = self.tok[tokens] + self.pos[: tokens.shape[1]].unsqueeze(0)
        return (h, None, (h,))  # the reference reads output[2][-1] (annotate_model.py:119,159)


CONFIG1 = dict(E=768, TE=100, H=400, U=24, O=30, T=12, vocab_size=2000, head_seed=2024, enc_seed=4321)


def config1_vocab():
    c = CONFIG1
    return {
        "upos": ToyVocab([f"P{k}" for k in range(c["U"] - 4)]),
        "deprel": ToyVocab([f"rel{k}" for k in range(c["O"] - 4)]),
        "ner_tag": ToyVocab([f"N{k}" for k in range(c["T"] - 4)]),
        "word": ToyVocab([]), "char": ToyVocab([]),
    }


def config1_head_state():
    from phonlp_amd import synthetic as S
    c = CONFIG1
    return S.make_state_dict(c["E"], c["TE"], c["H"], c["U"], c["O"], c["T"], seed=c["head_seed"])


class StubJointModelFull(StubJointModel):
    """The stand-in at PhoBERT-base head sizes with sample_data's label-set sizes (E=768, H=400,
    TE=100, U=24, O=30, T=12) and the table encoder: what tests/golden/config1_sample.npz was recorded with."""

    def __init__(self):
        nn.Module.__init__(self)
        c = CONFIG1
        self.phobert = TableEncoder(c["vocab_size"], c["E"], seed=c["enc_seed"])
        self._heads = config1_head_state()
        self.args = {"linearization": True, "distance": True}
        self.vocab = config1_vocab()
        self.tokenizer = ToyTokenizer(c["vocab_size"])
        self.eval()
