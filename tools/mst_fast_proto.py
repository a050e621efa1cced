#!/usr/bin/env python3
"""CPU prototype of the round-based ("all cycles of a round at once") Chu-Liu-Edmonds decode that
phonlp_amd/csrc/mst_fast.hip runs, checked here against the reference-order oracle (oracle/decoders.c).

Why it may replace the reference's one-cycle-per-level recursion (models/common/chuliu_edmonds.py:52-129):
 (G) if every finite entry of the edited matrix is a multiple of 2^-q and 2*n*max|entry|*2^q < 2^53, every value the
     reference forms in float64 (entries are sums of <= n original entries) is exact, i.e. the reference runs exact
     Chu-Liu-Edmonds with some tie-breaking and returns A maximum arborescence;
 (U) if every selection this decode makes has a strictly unique maximum, THE maximum arborescence is unique (exchange
     argument, by induction over the contractions), so any correct variant returns the same tree.
 Sentences failing (G) or (U) are decoded by the reference-order kernel instead.  Exact arithmetic = int64 fixed point.

    python tools/mst_fast_proto.py [n_cases]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

NEG = np.iinfo(np.int64).min // 4
MAX_ROUNDS = 32


def grid_check(arc):
    """-> (ok, q): fixed-point exponent q such that every finite entry * 2^q is an integer, and (G) holds."""
    a = np.asarray(arc, np.float32)
    n = a.shape[0]
    m = a.copy()
    np.fill_diagonal(m, 0)
    m[0] = 0
    if np.isnan(m).any() or np.isposinf(m).any():
        return False, 0
    fin = np.isfinite(m) & (m != 0)
    if not fin.any():
        return True, 0
    bits = m[fin].view(np.uint32)
    e = ((bits >> 23) & 0xFF).astype(np.int64)
    if (e == 0).any():
        return False, 0  # denormals: left to the reference-order path
    man = ((bits & 0x7FFFFF) | 0x800000).astype(np.int64)
    tz = np.zeros_like(e)
    for k in range(24):
        tz = np.where((tz == k) & (((man >> k) & 1) == 0), k + 1, tz)
    lo = int((e - 127 - 23 + tz).min())  # lowest set bit over all entries
    hi = int((e - 127).max()) + 1  # |entry| < 2^hi
    q = max(0, -lo)
    nbits = int(np.ceil(np.log2(max(n, 2)))) + 1
    return bool(hi + q + nbits <= 52), q


def decode(arc, q, root=-1, stats=None):
    """One decode, every cycle of a round contracted together.  -> (tree, score) or None when a selection is tied."""
    n = arc.shape[0]
    W = np.full((n, n), NEG, np.int64)
    a = np.asarray(arc, np.float64)
    fin = np.isfinite(a)
    W[fin] = np.round(a[fin] * (2.0 ** q)).astype(np.int64)
    np.fill_diagonal(W, NEG)
    W[0] = NEG
    if root >= 1:
        W[1:, 0] = NEG
        W[root] = NEG
        W[root, 0] = 0
    comp = np.arange(n)
    off = np.zeros(n, np.int64)
    best_j = np.zeros(n, np.int64)
    best_v = np.full(n, NEG, np.int64)

    def scan_row(i):
        row = np.where(comp == comp[i], NEG, W[i])
        j = int(row.argmax())
        v = row[j]
        if v == NEG or (row == v).sum() != 1:
            return False
        best_j[i], best_v[i] = j, v
        return True

    for i in range(1, n):
        if not scan_row(i):
            return None
    sel_i = np.arange(n)   # per component (indexed by its representative node): the selected in-edge sel_i -> sel_j
    sel_j = best_j.copy()
    sel_v = best_v + off
    alive = np.ones(n, bool)
    history = []
    nb = int(np.ceil(np.log2(max(n, 2)))) + 1
    for rnd in range(MAX_ROUNDS + 1):
        hc = np.zeros(n, np.int64)
        for X in range(1, n):
            if alive[X]:
                hc[X] = comp[sel_j[X]]
        ptr = hc.copy()
        for _k in range(nb):
            ptr = ptr[ptr]
        oncyc = np.zeros(n, bool)
        for X in range(1, n):
            if alive[X] and ptr[X] != 0:
                oncyc[ptr[X]] = True
        if not oncyc.any():
            break
        if rnd == MAX_ROUNDS:
            return None
        history.append((comp.copy(), sel_i.copy(), sel_j.copy()))
        label = np.full(n, -1)
        for X in range(1, n):
            if oncyc[X] and label[X] < 0:
                mem = [X]
                y = hc[X]
                while y != X:
                    mem.append(y)
                    y = hc[y]
                rep = min(mem)
                total = sum(int(sel_v[m]) for m in mem)
                for m in mem:
                    label[m] = rep
                    off[comp == m] += total - sel_v[m]
        merged_rows = [i for i in range(1, n) if oncyc[comp[i]]]
        for X in range(1, n):
            if oncyc[X] and label[X] != X:
                alive[X] = False
        for i in merged_rows:
            comp[i] = label[comp[i]]
        for i in merged_rows:
            if comp[best_j[i]] == comp[i] and not scan_row(i):
                return None
        for X in range(1, n):
            if oncyc[X] and alive[X]:
                rows = np.where(comp == X)[0]
                vals = best_v[rows] + off[rows]
                k = int(vals.argmax())
                if (vals == vals[k]).sum() != 1:
                    return None
                sel_i[X], sel_j[X], sel_v[X] = rows[k], best_j[rows[k]], vals[k]
    if stats is not None:
        stats.append(len(history))
    # expansion: the real in-edge of every component, rounds undone in reverse
    real_i, real_j = sel_i.copy(), sel_j.copy()
    comp_after = comp
    for comp_s, sel_i_s, sel_j_s in reversed(history):
        ni, nj = sel_i_s.copy(), sel_j_s.copy()
        for X in range(1, n):
            if comp_s[X] != X:
                continue  # not a representative in this state
            Y = comp_after[X]
            if comp_s[real_i[Y]] == X:
                ni[X], nj[X] = real_i[Y], real_j[Y]
        real_i, real_j, comp_after = ni, nj, comp_s
    tree = np.zeros(n, np.int64)
    tree[1:] = real_j[1:]
    assert (real_i[1:] == np.arange(1, n)).all()
    score = int(sum(W[i, tree[i]] for i in range(1, n)))
    return tree, score


def mst_one_root_fast(arc, stats=None):
    """-> heads (int64[n]) or None when the sentence must go to the reference-order path."""
    n = arc.shape[0]
    ok, q = grid_check(arc)
    if not ok or n < 2:
        return None
    r = decode(arc, q, -1, stats)
    if r is None:
        return None
    tree, _ = r
    roots = [i for i in range(1, n) if tree[i] == 0]
    if len(roots) == 1:
        return tree
    best, best_tree = None, None
    for root in roots:  # ref :154-161, ascending, strict '>'
        r = decode(arc, q, root, stats)
        if r is None:
            return None
        t, s = r
        s += int(round(float(arc[root, 0]) * 2.0 ** q))
        if best is None or s > best:
            best, best_tree = s, t
    return best_tree


def main():
    from oracle import c_oracle as C
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(1)
    n_fast = n_fall = 0
    rounds = []
    for case in range(n_cases):
        l = int(rng.integers(2, 70))
        kind = case % 6
        if kind == 0:
            m = rng.standard_normal((l, l)) * 2
        elif kind == 1:  # cycle-heavy
            m = rng.standard_normal((l, l))
            for k in range(1, l - 1, 2):
                m[k, k + 1] += 4
                m[k + 1, k] += 4
        elif kind == 2:  # rooty
            m = rng.standard_normal((l, l)) * 1.5
            m[:, 0] += 3
        elif kind == 3:  # ties
            m = rng.integers(-3, 1, (l, l)).astype(np.float64)
        elif kind == 4:  # coarse grid: some ties, exact
            m = np.round(rng.standard_normal((l, l)) * 8) / 4
        else:  # confident rows: near-zero log-probs
            m = rng.standard_normal((l, l)) * 6
        if kind not in (3, 4):
            m = m - m.max(axis=1, keepdims=True)
            m = m - np.log(np.exp(m).sum(axis=1, keepdims=True))
        arc = m.astype(np.float32)
        want, rc = C.mst_one_root(arc)
        got = mst_one_root_fast(arc, rounds)
        if got is None:
            n_fall += 1
            continue
        n_fast += 1
        if rc != 0 or not np.array_equal(got, want):
            print("MISMATCH case", case, "l", l, "kind", kind, got, want, rc)
            sys.exit(1)
    print(f"{n_cases} cases: {n_fast} fast (all equal to the reference-order oracle), {n_fall} fell back; "
          f"rounds per decode: mean {np.mean(rounds):.2f}, max {max(rounds)}")


if __name__ == "__main__":
    main()
