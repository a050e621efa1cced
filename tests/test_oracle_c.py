"""Pins the plain-C oracle (oracle/decoders.c) to the reference-generated fixtures and to the
numpy restatement on extra random cases."""
import numpy as np
import pytest

from oracle import c_oracle as C
from oracle import heads_oracle as O


@pytest.fixture(scope="module", autouse=True)
def _build():
    C.build()


def test_c_viterbi_golden(golden):
    z = golden("viterbi.npz")
    for c in range(int(z["n"])):
        s = z[f"v{c}_scores"]
        tags, best = C.viterbi_batch(s[None], z[f"v{c}_trans"], [s.shape[0]])
        assert tags[0, :s.shape[0]].tolist() == z[f"v{c}_tags"].tolist(), f"case {c}"
        assert best[0] == z[f"v{c}_best"]


def test_c_crf_golden(golden):
    z = golden("crf.npz")
    for c in range(int(z["n"])):
        got = C.crf_log_norm(z[f"c{c}_inputs"], z[f"c{c}_masks"], z[f"c{c}_trans"])
        np.testing.assert_allclose(got, z[f"c{c}_log_norm"], rtol=1e-5, atol=1e-5)


def test_c_mst_golden(golden):
    z = golden("mst.npz")
    po = ph = 0
    for k, l in enumerate(z["sizes"]):
        m = z["flat"][po:po + l * l].reshape(l, l)
        got, rc = C.mst_one_root(m)
        assert rc == 0
        assert got.tolist() == z["heads"][ph:ph + l].tolist(), f"case {k} kind {z['kinds'][k]} l={l}"
        po += l * l
        ph += l


def test_c_mst_vs_numpy_random():
    rng = np.random.default_rng(7)
    for trial in range(300):
        l = int(rng.integers(2, 70))
        kind = trial % 3
        if kind == 0:
            m = rng.standard_normal((l, l)) * 2
        elif kind == 1:
            m = rng.integers(-2, 1, (l, l)).astype(np.float64)
        else:
            m = rng.standard_normal((l, l))
            m[:, 0] += 2
        m = m.astype(np.float32)
        got, rc = C.mst_one_root(m)
        assert rc == 0
        assert got.tolist() == O.chuliu_edmonds_one_root(m).tolist(), f"trial {trial} l={l}"


def test_c_mst_batch_ragged():
    rng = np.random.default_rng(8)
    B, L = 6, 20
    arc = rng.standard_normal((B, L, L)).astype(np.float32)
    lens = [20, 17, 9, 4, 2, 1]
    heads, status = C.mst_batch(arc, lens)
    assert status.tolist() == [0] * (B - 1) + [1]  # l=1: the reference's broken-assert case
    for b, l in enumerate(lens[:-1]):
        assert heads[b, :l].tolist() == O.chuliu_edmonds_one_root(arc[b, :l, :l]).tolist()
