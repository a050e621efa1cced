"""BASELINE config 1 on the HIP path: annotate() of the 100 sentences of the reference's sample_data/input.txt at
PhoBERT-base head sizes (E=768, H=400, TE=100; label sets U=24, O=30, T=12), against the output files the reference's
own annotate_model.JointModel wrote for the same seeded head weights and table encoder
(tests/golden/config1_sample.npz, made by tests/golden/make_golden.py::gen_config1).  Bar: every POS tag, NER tag,
head and relation equal to the reference's, at every batch size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    assert torch.cuda.is_available()
    from phonlp_amd.compat import patch
    from tests.stub_model import StubJointModelFull
    return patch(StubJointModelFull())


@pytest.mark.parametrize("batch_size", [1, 8, 32, 100])
def test_config1_annotate_matches_reference(model, golden, tmp_path, batch_size):
    z = golden("config1_sample.npz")
    fin, fout = tmp_path / "in.txt", tmp_path / "out.txt"
    fin.write_text("\n".join(z["sent_text"].tolist()) + "\n")
    model.annotate(input_file=str(fin), output_file=str(fout), batch_size=batch_size)
    got, want = fout.read_text().splitlines(), str(z[f"out_bs{batch_size}"]).splitlines()
    assert len(got) == len(want) == 2562
    diff = {"pos": 0, "ner": 0, "head": 0, "rel": 0}
    words = 0
    for g, w in zip(got, want):
        if not w:
            assert not g
            continue
        words += 1
        gc, wc = g.split("\t"), w.split("\t")
        assert gc[:2] == wc[:2]
        for k, name in ((2, "pos"), (3, "ner"), (4, "head"), (5, "rel")):
            diff[name] += gc[k] != wc[k]
    print(f"config 1, batch_size {batch_size}: {words} words, tokens differing from the reference: {diff}")
    assert not any(diff.values()), diff


def test_config1_text_api(model, golden):
    """annotate(text=...) returns (data, pos, ner, dep) like the reference (annotate_model.py:218-229,295)."""
    z = golden("config1_sample.npz")
    sent = z["sent_text"].tolist()[3]
    data, pos, ner, dep = model.annotate(text=sent)
    want = [ln.split("\t") for ln in str(z["out_bs1"]).split("\n\n")[3].splitlines()]
    assert data == [sent.split(" ")]
    assert [p[0] for p in pos[0]] == [w[2] for w in want]
    assert ner[0] == [w[3] for w in want]
    assert [d[0] for d in dep[0]] == [w[4] for w in want] and [d[1] for d in dep[0]] == [w[5] for w in want]


def test_config1_mst_fast_path_rate(model, golden):
    """How many of config 1's 100 real sentences the round-based MST hands to the reference-order kernel on the scores
    the heads give them (the rate was only known on synthetic log-probabilities).  The gate is a margin, not
    exactness, so ordinary float scores of any dynamic range qualify; only (near-)ties fall back."""
    from phonlp_amd import ops
    from phonlp_amd.compat import refresh_heads
    z = golden("config1_sample.npz")
    sents = [t.split(" ") for t in z["sent_text"].tolist()]
    data = model.chunk_batches(model.process_data_parser(sents), 100)
    tokens, first_subword, _, _, _, sentlens = model.get_batch(0, data)
    arc, _ = model.dep_forward(tokens, first_subword, sentlens)
    dev = torch.device("cuda:0")
    heads, status, nfb = ops.mst_one_root_counted(torch.from_numpy(arc).to(dev),
                                                  torch.tensor(list(sentlens), dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    assert (status == 0).all()
    print(f"config 1: {int(nfb)} of {len(sents)} sentences took the reference-order fallback")
    assert 0 <= int(nfb) <= 5
