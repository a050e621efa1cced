"""Score-matrix families that stress the MST decoders' equivalence gate (csrc/mst_fast.hip, header): shared by the CPU
pin of the oracle against the reference (test_oracle_vs_reference.py) and the GPU parity test (test_gpu_parity.py)."""
import numpy as np


def log_softmax_rows(m):
    m = m.astype(np.float64)
    m = m - m.max(1, keepdims=True)
    return (m - np.log(np.exp(m).sum(1, keepdims=True))).astype(np.float32)


def make(kind, l, rng):
    """-> float32 [l, l] arc scores (rows = dependents), like a parser's log-probabilities unless said otherwise."""
    if kind == "confident":
        # a trained parser: one head per row near-certain (log-probabilities from -1e-9 to -100 in one row)
        return log_softmax_rows(rng.standard_normal((l, l)) * 2 * rng.choice([8.0, 30.0]))
    if kind == "wide":
        # no common grid at all: magnitudes from 1e-12 to 1e6, denormals and a few -inf ("no edge") entries
        m = -np.exp(rng.standard_normal((l, l)) * 9).astype(np.float32)
        m[rng.random((l, l)) < 0.02] = -np.inf
        m[rng.random((l, l)) < 0.01] = -1e-41  # denormal
        m[:, 0] = np.where(np.isinf(m[:, 0]), -1.0, m[:, 0])  # keep every word attachable to ROOT
        return m
    if kind == "nearties":
        # coarse grid (many exact ties) with one-ulp and few-ulp neighbours sprinkled in: gaps of 0, 1e-7, 1e-6 ...
        m = log_softmax_rows(np.round(rng.standard_normal((l, l)) * 2, 1))
        for _ in range(3 * l):
            i, j, k = rng.integers(0, l, 3)
            step = int(rng.integers(0, 4))
            v = m[i, k]
            for _ in range(step):
                v = np.nextafter(v, np.float32(0), dtype=np.float32)
            m[i, j] = v
        return m
    raise ValueError(kind)


KINDS = ("confident", "wide", "nearties")
