"""Extra pin, authoring container only: random-case comparison of the oracle with the reference's
own decoders loaded from /root/reference (skipped on the GPU box, where it does not exist)."""
import importlib.util
import os
import sys

import numpy as np
import pytest

from oracle import c_oracle as C
from oracle import heads_oracle as O

REF = "/root/reference/phonlp/models/common"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def _load(fn):
    spec = importlib.util.spec_from_file_location("ref_" + fn[:-3], os.path.join(REF, fn))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_mst_random_vs_reference():
    sys.setrecursionlimit(20000)
    cle = _load("chuliu_edmonds.py")
    rng = np.random.default_rng(11)
    for trial in range(400):
        l = int(rng.integers(2, 60))
        if trial % 4 == 0:
            m = rng.integers(-2, 1, (l, l)).astype(np.float32)
        else:
            m = (rng.standard_normal((l, l)) * (1 + trial % 3)).astype(np.float32)
            m -= np.log(np.exp(m).sum(1, keepdims=True))
        want = cle.chuliu_edmonds_one_root(m.copy()).tolist()
        assert O.chuliu_edmonds_one_root(m).tolist() == want, trial
        got, rc = C.mst_one_root(m)
        assert rc == 0 and got.tolist() == want, trial


@pytest.mark.parametrize("kind", ["confident", "wide", "nearties"])
def test_mst_hard_families_vs_reference(kind):
    """The families the GPU decoders' equivalence gate is tested on (tests/mst_families.py): the oracle they are compared
    with must itself follow the reference there -- scores of any dynamic range, exact ties next to one-ulp gaps."""
    from tests.mst_families import make
    sys.setrecursionlimit(20000)
    cle = _load("chuliu_edmonds.py")
    rng = np.random.default_rng({"confident": 21, "wide": 22, "nearties": 23}[kind])
    for trial in range(120):
        l = int(rng.integers(2, 70))
        m = make(kind, l, rng)
        with np.errstate(invalid="ignore"):  # -inf - -inf inside the reference's contraction
            want = cle.chuliu_edmonds_one_root(m.copy()).tolist()
        got, rc = C.mst_one_root(m)
        assert rc == 0 and got.tolist() == want, (kind, trial)
        if trial % 4 == 0:
            assert O.chuliu_edmonds_one_root(m).tolist() == want, (kind, trial)


def test_viterbi_random_vs_reference():
    crf = _load("crf.py")
    rng = np.random.default_rng(12)
    for trial in range(100):
        l, T = int(rng.integers(1, 80)), int(rng.integers(2, 20))
        s = np.maximum(rng.standard_normal((l, T)), 0).astype(np.float32)
        tr = (rng.standard_normal((T, T)) * 0.1 * (trial % 2)).astype(np.float32)
        want, best = crf.viterbi_decode(s, tr)
        got, gb = O.viterbi_decode(s, tr)
        assert got == [int(w) for w in want] and gb == best
        tags, cb = C.viterbi_batch(s[None], tr, [l])
        assert tags[0, :l].tolist() == got and cb[0] == best
