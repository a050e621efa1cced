"""GPU parity tests: the HIP path (through the C-ABI, via the torch custom ops) against the CPU oracle,
the reference-generated golden fixtures, and size-independent properties at the headline sizes.

Bars (BASELINE.json north_star): decoded heads / relations / POS / NER tags bit-exact; arc and label
scores and the CRF log-partition within 1e-3 relative in fp32 (the tolerances below are tighter).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import c_oracle as C  # noqa: E402
from oracle import heads_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from phonlp_amd import _lib
    _lib.lib()  # fail loudly if the HIP library is missing
    C.build()
    return torch.device("cuda:0")


@pytest.fixture(params=[0, 1, 2], ids=["f32mfma", "bf16x6", "f16x3"])
def engine(request, dev):
    """Runs a GEMM-bearing test once per matrix engine of the library (include/phonlp_heads.h)."""
    from phonlp_amd import ops
    before = ops.get_gemm_engine()
    ops.set_gemm_engine(request.param)
    yield request.param
    ops.set_gemm_engine(before)


def _params(z, prefix="p_"):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def _close(got, want, rtol=2e-4, atol=2e-5):
    got, want = torch.as_tensor(got).float().cpu(), torch.as_tensor(want).float().cpu()
    fin = torch.isfinite(want)
    assert torch.equal(fin, torch.isfinite(got)), "non-finite pattern differs"
    torch.testing.assert_close(got[fin], want[fin], rtol=rtol, atol=atol)


# ------------------------------------------------------------------------------------------ dense layer
@pytest.mark.parametrize("M,N,K,relu,masked", [
    (300, 70, 868, True, False), (129, 9, 868, True, True), (64, 416, 416, False, False),
    (1000, 868, 868, False, True), (257, 24, 400, False, True), (1500, 300, 868, True, True), (1027, 129, 20, False, False),
    (2048, 128, 4, False, False), (128, 128, 32, True, False), (5, 3, 4, False, False),
    # narrow layers over many rows (the one-wavefront 32-row tiles)
    (4100, 24, 400, False, True), (5000, 40, 868, True, True), (4096, 9, 20, True, False), (4099, 64, 36, False, False),
])
def test_linear_vs_torch_fp32(dev, engine, M, N, K, relu, masked):
    from phonlp_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    L = 50
    B = (M + L - 1) // L
    lens = torch.randint(1, L + 1, (B,), generator=g).to(torch.int32)
    ref = x @ w.t() + b
    if relu:
        ref = torch.relu(ref)
    if masked:
        rows = torch.arange(M)
        ref = ref * ((rows % L) < lens[rows // L].long()).unsqueeze(1)
    got = ops.linear(x.to(dev), w.to(dev), b.to(dev), relu, lens.to(dev) if masked else None, L if masked else 0)
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-4, atol=1e-4)


def _linear_split(ops, x, w, bias=None, relu=False, lens=None, L=0, pad_bias=None):
    """Engine 2 proper: both operands as split images (include/phonlp_heads.h phonlp_split_rows / _linear_split)."""
    xp, xi = ops.split_rows(x)
    wp, wi = ops.split_rows(w)
    return ops.linear_split(xp, xi, wp, wi, bias, x.shape[1], relu, lens, L, pad_bias, 0)


@pytest.mark.parametrize("M,N,K,relu,masked", [
    (300, 70, 868, True, False), (129, 9, 868, True, True), (64, 416, 416, False, False),
    (1000, 868, 868, False, True), (257, 24, 400, False, True), (1500, 300, 868, True, True), (1027, 129, 20, False, False),
    (2048, 128, 4, False, False), (128, 128, 32, True, False), (5, 3, 4, False, False), (700, 413, 408, True, True),
])
def test_linear_split_vs_torch_fp32(dev, M, N, K, relu, masked):
    """The f16x3 dense layer on split images against torch fp32, same cases as the fp32-operand entry point; with
    `masked` also the folded-form padding rows (relu(pad_bias))."""
    from phonlp_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    pb = torch.randn(N, generator=g)
    L = 50
    B = (M + L - 1) // L
    lens = torch.randint(1, L + 1, (B,), generator=g).to(torch.int32)
    ref = x @ w.t() + b
    if relu:
        ref = torch.relu(ref)
    if masked:
        rows = torch.arange(M)
        live = ((rows % L) < lens[rows // L].long()).unsqueeze(1)
        ref = torch.where(live, ref, (torch.relu(pb) if relu else pb).expand_as(ref))
    got = _linear_split(ops, x.to(dev), w.to(dev), b.to(dev), relu, lens.to(dev) if masked else None, L if masked else 0,
                        pb.to(dev) if masked else None)
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-4, atol=1e-4)


def test_split_rows_image(dev):
    """The split image itself: h + l reproduces the scaled row to 2^-23 relative for elements within 2^16 of the row
    maximum, zero rows and zero padding stay zero, inv_scale is the exact inverse power of two, strided input."""
    from phonlp_amd import ops
    g = torch.Generator().manual_seed(5)
    rows, K = 37, 100
    x = torch.randn(rows, 2 * K, generator=g) * torch.exp(4 * torch.randn(rows, 1, generator=g))
    x[3] = 0
    x[5, :K] *= 1e-30
    xs = x[:, :K]
    planes, inv = ops.split_rows(x.to(dev)[:, :K])
    K16 = (K + 15) // 16 * 16
    img = planes.cpu().view(torch.float16).view(rows, K16 // 8, 2, 8).double()
    h, l = img[:, :, 0, :].reshape(rows, K16), img[:, :, 1, :].reshape(rows, K16)
    assert (h[:, K:] == 0).all() and (l[:, K:] == 0).all()
    inv = inv.cpu().double()
    assert torch.equal(torch.log2(inv), torch.log2(inv).round())
    rec = (h + l)[:, :K] * inv.unsqueeze(1)
    mx = xs.abs().max(1, keepdim=True).values.double().clamp_min(1e-300)
    assert ((h.abs().max(1).values < 2.0 ** 15).all())
    big = xs.abs().double() >= mx * 2.0 ** -15
    assert ((rec - xs.double()).abs()[big] <= xs.abs().double()[big] * 2.0 ** -23).all()
    assert ((rec - xs.double()).abs() <= torch.maximum(xs.abs().double() * 2.0 ** -23, mx * 2.0 ** -38)).all()
    assert (rec[3] == 0).all()


@pytest.mark.parametrize("M,N,K", [(512, 640, 401), (256, 868, 868), (384, 256, 4096), (1300, 650, 401), (1024, 256, 2048)])
def test_engines_against_fp64(dev, M, N, K):
    """All matrix engines are fp32-grade: error against an fp64 product, in units of sum_k |a||b| (the scale an
    fp32 dot product's rounding error lives on), stays below 1e-6, and the split engines are no worse than
    the f32 MFMA chain.  Rows carry widely different magnitudes (exp(2*N(0,1))) to exercise the split."""
    from phonlp_amd import ops
    g = torch.Generator().manual_seed(K)
    Kp = (K + 3) // 4 * 4
    x = torch.zeros(M, Kp); x[:, :K] = torch.randn(M, K, generator=g) * torch.exp(2 * torch.randn(M, 1, generator=g))
    w = torch.zeros(N, Kp); w[:, :K] = torch.randn(N, K, generator=g) * torch.exp(torch.randn(N, 1, generator=g))
    ref = x.double() @ w.double().t()
    scale = x.double().abs() @ w.double().abs().t()
    errs = []
    before = ops.get_gemm_engine()
    try:
        for e in (0, 1):
            ops.set_gemm_engine(e)
            y = ops.linear(x.to(dev), w.to(dev), None, False, None, 0).cpu().double()
            errs.append(((y - ref).abs() / scale).max().item())
        y = _linear_split(ops, x.to(dev), w.to(dev)).cpu().double()
        errs.append(((y - ref).abs() / scale).max().item())
    finally:
        ops.set_gemm_engine(before)
    assert max(errs) < 1e-6, errs
    assert errs[1] <= 1.5 * errs[0] + 1e-8 and errs[2] <= 1.5 * errs[0] + 1e-8, errs


# ------------------------------------------------------------------------------------------ Viterbi
def test_viterbi_golden_bit_exact(dev, golden):
    from phonlp_amd import ops
    z = golden("viterbi.npz")
    for c in range(int(z["n"])):
        s, tr = z[f"v{c}_scores"], z[f"v{c}_trans"]
        l, T = s.shape
        logits = torch.zeros(2, l + 3, T)  # padded batch: second sentence is shorter
        logits[0, :l] = torch.from_numpy(s)
        logits[1, :max(l - 1, 1)] = torch.from_numpy(s[:max(l - 1, 1)])
        lens = torch.tensor([l, max(l - 1, 1)], dtype=torch.int32)
        tags, best = ops.crf_viterbi(logits.to(dev), torch.from_numpy(tr).to(dev), lens.to(dev))
        assert tags[0, :l].cpu().tolist() == z[f"v{c}_tags"].tolist(), f"case {c}"
        assert tags[0, l:].abs().sum().item() == 0
        assert np.float32(best[0].item()) == z[f"v{c}_best"], f"case {c}"
        want1, b1 = O.viterbi_decode(s[:max(l - 1, 1)], tr)
        assert tags[1, :max(l - 1, 1)].cpu().tolist() == want1 and np.float32(best[1].item()) == np.float32(b1)


@pytest.mark.parametrize("B,L,T,mode", [(256, 128, 9, "relu"), (64, 127, 12, "relu_zero"), (33, 300, 40, "rand"),
                                        (7, 1, 9, "rand"), (256, 127, 9, "ints")])
def test_viterbi_random_vs_c_oracle(dev, B, L, T, mode):
    from phonlp_amd import ops
    rng = np.random.default_rng(B + L + T)
    x = rng.standard_normal((B, L, T)).astype(np.float32)
    tr = (rng.standard_normal((T, T)) * 0.1).astype(np.float32)
    if mode.startswith("relu"):
        x = np.maximum(x, 0)
    if mode == "relu_zero":
        tr[:] = 0
    if mode == "ints":
        x = rng.integers(-2, 3, (B, L, T)).astype(np.float32)
        tr = rng.integers(-1, 2, (T, T)).astype(np.float32)
    lens = np.sort(rng.integers(1, L + 1, B))[::-1].astype(np.int32).copy()
    lens[0] = L
    want_tags, want_best = C.viterbi_batch(x, tr, lens)
    tags, best = ops.crf_viterbi(torch.from_numpy(x).to(dev), torch.from_numpy(tr).to(dev), torch.from_numpy(lens).to(dev))
    assert np.array_equal(tags.cpu().numpy(), want_tags)
    assert np.array_equal(best.cpu().numpy(), want_best)  # single fp32 adds: bit-exact


# ------------------------------------------------------------------------------------------ CRF forward
def test_crf_log_partition_golden(dev, golden):
    from phonlp_amd import ops
    z = golden("crf.npz")
    for c in range(int(z["n"])):
        x, m, t, tr = (torch.from_numpy(z[f"c{c}_{k}"]) for k in ("inputs", "masks", "tags", "trans"))
        ln, un, bi = ops.crf_log_partition(x.to(dev), tr.to(dev), m.to(torch.uint8).to(dev), None,
                                           t.to(torch.int32).to(dev))
        torch.testing.assert_close(ln.cpu(), torch.from_numpy(z[f"c{c}_log_norm"]), rtol=1e-5, atol=1e-5)
        loss = (-(un + bi - ln)).sum().item() / x.shape[0]
        assert abs(loss - float(z[f"c{c}_loss"])) <= 1e-4 * max(1.0, abs(float(z[f"c{c}_loss"])))


def test_crf_log_partition_random(dev):
    from phonlp_amd import ops
    rng = np.random.default_rng(5)
    B, L, T = 256, 127, 9
    x = np.maximum(rng.standard_normal((B, L, T)), 0).astype(np.float32)
    tr = (rng.standard_normal((T, T)) * 0.1).astype(np.float32)
    lens = np.sort(rng.integers(1, L + 1, B))[::-1].astype(np.int32).copy()
    pad = (np.arange(L)[None, :] >= lens[:, None])
    want = C.crf_log_norm(x, pad, tr)
    ln, _, _ = ops.crf_log_partition(torch.from_numpy(x).to(dev), torch.from_numpy(tr).to(dev), None,
                                     torch.from_numpy(lens).to(dev), None)
    np.testing.assert_allclose(ln.cpu().numpy(), want, rtol=1e-5, atol=1e-4)
    ln2, _, _ = ops.crf_log_partition(torch.from_numpy(x).to(dev), torch.from_numpy(tr).to(dev),
                                      torch.from_numpy(pad.astype(np.uint8)).to(dev), None, None)
    assert torch.equal(ln, ln2)  # mask form == sentlens form


# ------------------------------------------------------------------------------------------ MST
def _run_mst(dev, mats, L):
    from phonlp_amd import ops
    B = len(mats)
    rng = np.random.default_rng(99)
    arc = rng.standard_normal((B, L, L)).astype(np.float32)  # garbage outside [:l,:l] must be ignored
    lens = np.zeros(B, np.int32)
    for b, m in enumerate(mats):
        l = m.shape[0]
        arc[b, :l, :l] = m
        lens[b] = l
    heads, status, nfb = ops.mst_one_root_counted(torch.from_numpy(arc).to(dev), torch.from_numpy(lens).to(dev))
    torch.cuda.synchronize()
    _run_mst.fallbacks = int(nfb.item())
    return heads.cpu().numpy(), status.cpu().numpy()


@pytest.fixture(params=["rounds", "reference-order"])
def mst_path(request):
    """Both MST decoders: the round-based kernel (with its per-sentence fallback) and the reference-order kernel alone."""
    from phonlp_amd import ops
    was = ops.get_mst_fast_path()
    ops.set_mst_fast_path(request.param == "rounds")
    yield request.param
    ops.set_mst_fast_path(was)


def test_mst_golden_bit_exact(dev, golden, mst_path):
    z = golden("mst.npz")
    mats, want = [], []
    po = ph = 0
    for l in z["sizes"]:
        mats.append(z["flat"][po:po + l * l].reshape(l, l))
        want.append(z["heads"][ph:ph + l])
        po += l * l
        ph += l
    heads, status = _run_mst(dev, mats, 128)
    assert (status == 0).all()
    for k, (m, w) in enumerate(zip(mats, want)):
        l = m.shape[0]
        assert heads[k, :l].tolist() == w.tolist(), f"case {k} kind {z['kinds'][k]} l={l}"
        assert not heads[k, l:].any()


@pytest.mark.parametrize("B,L,kind", [(256, 128, "randn"), (128, 128, "rooty"), (128, 100, "ints"), (64, 128, "pairs"),
                                      # one word per lane / odd widths / tiny sentences on the round-based kernel
                                      (64, 43, "randn"), (40, 64, "pairs"), (33, 17, "rooty"), (50, 5, "randn"), (70, 97, "randn"),
                                      (24, 160, "randn"), (8, 300, "randn"), (6, 512, "randn"),
                                      # the two-phase form of the long-sentence kernel: many candidate roots per sentence
                                      # (more than its four slots: round-robin), all-ties integer matrices, pair cycles
                                      (12, 160, "rooty"), (6, 200, "ints"), (9, 170, "pairs")])
def test_mst_random_vs_c_oracle(dev, B, L, kind, mst_path):
    rng = np.random.default_rng(B * 3 + L)
    mats = []
    for b in range(B):
        l = L if b % 3 == 0 else int(rng.integers(2, L + 1))
        if kind == "ints":
            m = rng.integers(-3, 1, (l, l)).astype(np.float32)
        else:
            m = rng.standard_normal((l, l)) * 2
            if kind == "rooty":
                m[:, 0] += 3
            if kind == "pairs":
                for k in range(1, l - 1, 2):
                    m[k, k + 1] += 4
                    m[k + 1, k] += 4
            m = m - np.log(np.exp(m).sum(1, keepdims=True))
            m = m.astype(np.float32)
        mats.append(m)
    heads, status = _run_mst(dev, mats, L)
    assert (status == 0).all()
    for b, m in enumerate(mats):
        want, rc = C.mst_one_root(m)
        assert rc == 0
        assert heads[b, :m.shape[0]].tolist() == want.tolist(), f"sentence {b} l={m.shape[0]}"
    if mst_path == "rounds" and (L <= 128 or L >= 160):
        # (L <= 128: csrc/mst_fast.hip; L >= 160: csrc/mst_round.hip, the matrix left in global memory)
        # float log-probabilities take the round-based kernel (a few sentences may still fall back: a sentence of 2-3
        # words whose scores tie, or sums that are not provably exact); all-ties integer matrices never do
        nfb = _run_mst.fallbacks
        assert (nfb >= B // 2) if kind == "ints" else (nfb <= B // 8), f"{nfb} of {B} sentences fell back"


@pytest.mark.parametrize("B,L,kind", [(192, 128, "confident"), (96, 128, "wide"), (128, 128, "nearties"), (64, 60, "nearties"),
                                      (16, 200, "confident"), (12, 300, "nearties"), (8, 260, "wide")])
def test_mst_margin_families_vs_c_oracle(dev, B, L, kind, mst_path):
    """The round-based decoders' gate is a MARGIN (csrc/mst_fast.hip header), not exactness: a trained parser's near-certain
    arcs (log-probabilities from -1e-9 to -100 in a row) must take the fast path, and matrices full of exact ties and
    one-ulp gaps must come out as the reference's tree whichever decoder ends up with them."""
    from tests.mst_families import make
    rng = np.random.default_rng(B * 7 + L)
    mats = [make(kind, L if b % 3 == 0 else int(rng.integers(2, L + 1)), rng) for b in range(B)]
    heads, status = _run_mst(dev, mats, L)
    assert (status == 0).all()
    for b, m in enumerate(mats):
        want, rc = C.mst_one_root(m)
        assert rc == 0
        assert heads[b, :m.shape[0]].tolist() == want.tolist(), f"{kind} sentence {b} l={m.shape[0]}"
    if mst_path == "rounds":
        nfb = _run_mst.fallbacks
        print(f"{kind} B={B} L={L}: {nfb} fallbacks")
        # confident scores take the fast path; "wide" ones (max |entry| ~ 1e6 beside 1e-12: the margin scales with the
        # largest entry) and tie-ridden ones mostly end up with the reference-order kernel -- parity holds either way
        if kind == "confident":
            assert nfb <= B // 8, f"{nfb} of {B} sentences fell back"


def test_mst_root_only_sentence_reports_no_tree(dev):
    heads, status = _run_mst(dev, [np.zeros((1, 1), np.float32), np.zeros((2, 2), np.float32)], 8)
    assert status.tolist() == [1, 0]  # the reference asserts on l == 1 (chuliu_edmonds.py:162-168)
    assert heads[1, :2].tolist() == [0, 0]


# ------------------------------------------------------------------------------------------ biaffine fixtures
def test_deep_biaffine_golden(dev, engine, golden):
    """DeepBiaffineScorer fixtures (reference output, incl. the flat-reshape quirk and the ignored bias)."""
    from phonlp_amd import ops
    z = golden("biaffine.npz")
    for c in range(int(z["n_db"])):
        p = _params(z, f"db{c}_p_")
        h = torch.from_numpy(z[f"db{c}_h"])
        B, L, D = h.shape
        Dp = (D + 3) // 4 * 4  # the kernels need K % 4 == 0: zero-pad features and weight columns
        pad = lambda t: torch.nn.functional.pad(t, (0, Dp - D))
        W12, b12, Wbt = ops.pack_scorer(pad(p["W1.weight"]).to(dev), p["W1.bias"].to(dev), pad(p["W2.weight"]).to(dev),
                                        p["W2.bias"].to(dev), p["scorer.W_bilin.weight"].contiguous().to(dev))
        H = p["W1.weight"].shape[0]
        O = p["scorer.W_bilin.weight"].shape[2]
        ref = torch.from_numpy(z[f"db{c}_out"])  # B x L x L x O
        _, scores = ops.deep_biaffine_label(pad(h).contiguous().to(dev), W12, b12, Wbt, H, O, True)
        _close(scores.permute(0, 1, 3, 2), ref, rtol=1e-4, atol=1e-5)
        if O == 1:
            _, raw = ops.deep_biaffine_arc(pad(h).contiguous().to(dev), W12, b12, Wbt, H, 1, False, False, True)
            _close(raw[0], ref.squeeze(3), rtol=1e-4, atol=1e-5)


def test_pairwise_bilinear_golden_on_gpu(dev, engine, golden):
    """The PairwiseBilinear fixtures (reference output for raw x1 != x2, O up to 33, non-zero bias that the reference
    never adds, biaffine.py:21-37) through phonlp_deep_biaffine_label.  The entry point takes ONE hidden tensor and
    ReLU'd first layers, so a signed pair (x1, x2) goes in as four non-negative problems -- hidden = [x1+- | x2+-], W1 /
    W2 selecting the halves, zero biases -- combined by bilinearity; the weight is laid out so that the pack's flat
    reshape (the reference's quirk) lands on the fixture's effective weight."""
    from phonlp_amd import ops
    z = golden("biaffine.npz")
    for c in range(int(z["n_pb"])):
        x1, x2 = torch.from_numpy(z[f"pb{c}_x1"]), torch.from_numpy(z[f"pb{c}_x2"])
        w, ref = torch.from_numpy(z[f"pb{c}_weight"]), torch.from_numpy(z[f"pb{c}_out"])  # ref: B x L1 x L2 x O
        assert float(np.abs(z[f"pb{c}_bias"]).max()) > 0  # the fixture's bias is non-zero and must not show up
        B, L1, D1 = x1.shape
        _, L2, D2 = x2.shape
        Ow = w.shape[2]
        H, L = max(D1, D2), max(L1, L2)
        eff = w.reshape(D1, Ow, D2)  # what the reference's view() makes of the stored (D1, D2, O) weight
        big = torch.zeros(H + 1, Ow, H + 1)
        big[:D1, :, :D2] = eff
        wbig = big.reshape(H + 1, H + 1, Ow).contiguous()  # flat memory = [d1][o][d2]: the pack reads it back that way
        D = 2 * H
        Dp = (D + 3) // 4 * 4
        W1 = torch.zeros(H, Dp); W1[:, :H] = torch.eye(H)
        W2 = torch.zeros(H, Dp); W2[:, H:2 * H] = torch.eye(H)
        W12, b12, Wbt = ops.pack_scorer(W1.to(dev), torch.zeros(H).to(dev), W2.to(dev), torch.zeros(H).to(dev), wbig.to(dev))
        total = torch.zeros(B, L, Ow, L)
        for s1 in (1.0, -1.0):
            for s2 in (1.0, -1.0):
                hid = torch.zeros(B, L, Dp)
                hid[:, :L1, :D1] = torch.relu(s1 * x1)
                hid[:, :L2, H:H + D2] = torch.relu(s2 * x2)
                _, sc = ops.deep_biaffine_label(hid.contiguous().to(dev), W12, b12, Wbt, H, Ow, True)
                total += s1 * s2 * sc.cpu()
        got = total.permute(0, 1, 3, 2)[:, :L1, :L2]
        _close(got, ref, rtol=2e-4, atol=2e-4 * float(ref.abs().max()))


# ------------------------------------------------------------------------------------------ whole head path
@pytest.mark.parametrize("name", ["joint_small.npz", "joint_zero_init.npz"])
def test_joint_fixture_end_to_end(dev, engine, golden, name):
    """The reference's own dep_forward / tagger_forward / decoders on a ragged batch (fixture made by running
    annotate_model.JointModel) against the HIP path fed the same encoder rows."""
    from phonlp_amd import HeadWeights, TaskHeads
    z = golden(name)
    heads = TaskHeads(HeadWeights.from_state_dict(_params(z), dev))
    lens1, lens = z["dep_sentlens"].tolist(), z["tag_sentlens"].tolist()
    dep = heads.parser(torch.from_numpy(z["dep_enc"]).to(dev), lens1)
    tag = heads.tagger(torch.from_numpy(z["tag_enc"]).to(dev), lens)
    _close(dep["arc_logp"], z["dep_arc_logp"], rtol=1e-4, atol=2e-5)
    assert np.array_equal(dep["label_argmax"].cpu().numpy().astype(np.int64), z["dep_label_argmax"])
    assert np.array_equal(tag["pos_pred"].cpu().numpy().astype(np.int64), z["tag_pos_pred"])
    _close(tag["ner_logits"], z["tag_ner_logits"], rtol=1e-4, atol=1e-5)
    # (a) tie semantics: decoders fed the REFERENCE's scores must reproduce the reference's decoded output
    #     bit for bit (the zero-init fixture is all ties: heads come from the distance prior alone)
    ref_arc = torch.from_numpy(z["dep_arc_logp"]).to(dev)
    h_ref, st_ref = heads.mst(ref_arc, lens1)
    assert (st_ref == 0).all()
    got_heads = np.concatenate([h_ref[i, 1:l].cpu().numpy() for i, l in enumerate(lens1)])
    assert got_heads.tolist() == z["dec_heads"].tolist()
    from phonlp_amd import ops
    rels_ref = ops.gather_labels(torch.from_numpy(z["dep_label_argmax"]).to(torch.int32).to(dev), h_ref,
                                 torch.tensor(lens1, dtype=torch.int32, device=dev))
    got_rels = np.concatenate([rels_ref[i, 1:l].cpu().numpy() for i, l in enumerate(lens1)])
    assert got_rels.tolist() == z["dec_rels"].tolist()
    ner_ref, _ = heads.viterbi(torch.from_numpy(z["tag_ner_logits"]).to(dev), lens)
    got_ner = np.concatenate([ner_ref[i, :l].cpu().numpy() for i, l in enumerate(lens)])
    assert got_ner.tolist() == z["dec_ner"].tolist()
    # (b) end to end from the GPU's own scores.  Exact for the non-degenerate model; the zero-init model has
    #     exact ties ACROSS rows on the CPU that fp32 rounding of the row normalisers may break either way.
    out = heads.annotate_batch(torch.from_numpy(z["dep_enc"]).to(dev), lens1, torch.from_numpy(z["tag_enc"]).to(dev), lens)
    assert (out["mst_status"] == 0).all()
    if name == "joint_small.npz":
        got_heads = np.concatenate([out["heads"][i, 1:l].cpu().numpy() for i, l in enumerate(lens1)])
        got_rels = np.concatenate([out["rels"][i, 1:l].cpu().numpy() for i, l in enumerate(lens1)])
        got_ner = np.concatenate([out["ner"][i, :l].cpu().numpy() for i, l in enumerate(lens)])
        assert got_heads.tolist() == z["dec_heads"].tolist()
        assert got_rels.tolist() == z["dec_rels"].tolist()
        assert got_ner.tolist() == z["dec_ner"].tolist()
    # CRF loss as predict_ner evaluates it
    ln, un, bi = heads.crf_log_partition(torch.from_numpy(z["tag_ner_logits"]).to(dev),
                                         pad_mask=torch.from_numpy(z["tag_words_mask"]),
                                         gold=torch.from_numpy(z["crf_gold"]))
    loss = (-(un + bi - ln)).sum().item() / len(lens)
    assert abs(loss - float(z["crf_loss"])) <= 1e-4 * max(1.0, abs(float(z["crf_loss"])))


def _near_tie_ok(got_idx, want_idx, scores, tol):
    """Arg-max parity up to fp32 rounding: wherever the indices differ, the oracle's score at the GPU's
    index must be within `tol` of the oracle's maximum (a genuine near-tie)."""
    diff = got_idx != want_idx
    if not diff.any():
        return 0
    top = scores.max(-1)[0]
    at = torch.gather(scores, -1, got_idx.long().unsqueeze(-1)).squeeze(-1)
    assert bool(((top - at)[diff] <= tol).all()), "arg-max differs where the oracle has a clear winner"
    return int(diff.sum())


@pytest.mark.parametrize("B,L,ragged", [(32, 64, False), (24, 128, True), (3, 200, True)])
def test_synthetic_heads_vs_oracle(dev, engine, B, L, ragged):
    """BASELINE config 2 (B=32, L=64) and ragged / long variants with the 768/400/33/9 head sizes."""
    from phonlp_amd import HeadWeights, TaskHeads
    from phonlp_amd import synthetic as S
    sd = S.make_state_dict()
    enc_dep, enc_tag = S.make_encoder_outputs(B, L)
    lens1 = S.make_sentlens(B, L, ragged)
    lens = [l - 1 for l in lens1]
    heads = TaskHeads(HeadWeights.from_state_dict(sd, dev))
    dep = heads.parser(enc_dep.to(dev), lens1, want_raw=True, want_label_scores=True)
    tag = heads.tagger(enc_tag.to(dev), lens)
    od = O.dep_heads(enc_dep, lens1, sd)
    ot = O.tagger_heads(enc_tag, lens, sd)
    _close(dep["hidden_out"], od["hidden_out"], rtol=1e-4, atol=1e-5)
    _close(dep["raw"][0], od["unlabeled_raw"], rtol=1e-3, atol=1e-4)
    _close(dep["raw"][1], od["lin_raw"], rtol=1e-3, atol=1e-4)
    _close(dep["raw"][2], od["dist_raw"], rtol=1e-3, atol=1e-4)
    _close(dep["arc_logp"], od["arc_logp"], rtol=1e-3, atol=1e-4)
    _close(dep["label_scores"].permute(0, 1, 3, 2), od["label_scores"], rtol=1e-3, atol=1e-4)
    flips = _near_tie_ok(dep["label_argmax"].cpu(), od["label_argmax"], od["label_scores"], 1e-4)
    assert flips == 0  # near-ties would be tolerated by _near_tie_ok; none occurs on these inputs (printed below)
    _close(tag["ner_logits"], ot["ner_logits"], rtol=1e-4, atol=1e-5)
    _near_tie_ok(tag["pos_pred"].cpu(), ot["pos_pred"], ot["upos_logits"], 1e-5)
    # decoders are bit-exact GIVEN the same scores: feed the GPU decoders the GPU scores and compare with
    # the C oracle on those same scores
    ner, best = heads.viterbi(tag["ner_logits"], lens)
    wt, wb = C.viterbi_batch(tag["ner_logits"].cpu().numpy(), sd["crit_ner._transitions"].numpy(), lens)
    assert np.array_equal(ner.cpu().numpy(), wt) and np.array_equal(best.cpu().numpy(), wb)
    h, st = heads.mst(dep["arc_logp"], lens1)
    wh, ws = C.mst_batch(dep["arc_logp"].cpu().numpy(), lens1)
    assert np.array_equal(st.cpu().numpy(), ws) and np.array_equal(h.cpu().numpy(), wh)
    # and end-to-end: trees decoded from GPU scores vs trees decoded from oracle scores
    oh, _ = C.mst_batch(od["arc_logp"].numpy(), lens1)
    nh = int((h.cpu().numpy() != oh).sum())
    print(f"synthetic heads vs oracle, engine {engine}, B={B} L={L} ragged={ragged}: label arg-max flips at near-ties "
          f"{flips} of {B * L * L} dense cells; heads differing from the oracle-score trees {nh} of {sum(lens1) - B} words")
    assert nh == 0  # bit-exact end to end on these inputs; the measured count is printed (DESIGN.md section 2 quotes it)


@pytest.mark.parametrize("B,L,lens1", [(1, 2, [2]), (1, 3, [3]), (2, 5, [5, 2]), (5, 9, [9, 7, 4, 2, 2]),
                                       (3, 64, [64, 33, 2]), (1, 128, [128])])
def test_tiny_and_ragged_batches_end_to_end(dev, engine, B, L, lens1):
    """The smallest inputs annotate() can produce (one sentence of one word: ROOT + 1 node on the parser path, one
    token on the tagger path) and ragged batches whose tail sentences have one word, through annotate_batch (three
    streams, folded dep_hid, labels at the decoded heads) against the oracle chain end to end."""
    from phonlp_amd import HeadWeights, TaskHeads
    from phonlp_amd import synthetic as S
    sd = S.make_state_dict()
    enc_dep, enc_tag = S.make_encoder_outputs(B, L, seed=77 + B + L)
    lens = [l - 1 for l in lens1]
    heads = TaskHeads(HeadWeights.from_state_dict(sd, dev))
    out = heads.annotate_batch(enc_dep.to(dev), lens1, enc_tag.to(dev), lens)
    torch.cuda.synchronize()
    od = O.dep_heads(enc_dep, lens1, sd)
    ot = O.tagger_heads(enc_tag, lens, sd)
    wh, ws = C.mst_batch(od["arc_logp"].numpy(), lens1)
    assert not ws.any() and (out["mst_status"] == 0).all()
    assert np.array_equal(out["heads"].cpu().numpy(), wh)
    rel = np.zeros((B, L), np.int64)
    for b in range(B):
        for i in range(1, lens1[b]):
            rel[b, i] = int(od["label_argmax"][b, i, wh[b, i]])
    assert np.array_equal(out["rels"].cpu().numpy(), rel)
    wt, _ = C.viterbi_batch(ot["ner_logits"].numpy(), sd["crit_ner._transitions"].numpy(), lens)
    assert np.array_equal(out["ner"].cpu().numpy(), wt)
    assert np.array_equal(out["pos"].cpu().numpy(), ot["pos_pred"].numpy())
    # padding: zeros everywhere past the sentence
    for b in range(B):
        assert (out["heads"][b, lens1[b]:] == 0).all() and (out["rels"][b, lens1[b]:] == 0).all()
        assert (out["ner"][b, lens[b]:] == 0).all() and (out["pos"][b, lens[b]:] == 0).all()


@pytest.mark.parametrize("lin,dist", [(False, False), (True, False), (False, True)])
def test_scorer_flags(dev, engine, lin, dist):
    """args["linearization"] / args["distance"] off (ref annotate_model.py:77-94, 192-208): arc scores against the
    oracle with the same flags, through parser() (layer by layer) and annotate_batch (folded weights, engine paths)."""
    from phonlp_amd import HeadWeights, TaskHeads
    from phonlp_amd import synthetic as S
    B, L = 12, 40
    sd = S.make_state_dict(64, 12, 48, 7, 6, 5, seed=11)
    enc_dep, enc_tag = S.make_encoder_outputs(B, L, 64, seed=5)
    lens1 = S.make_sentlens(B, L, True)
    lens = [l - 1 for l in lens1]
    heads = TaskHeads(HeadWeights.from_state_dict(sd, dev, linearization=lin, distance=dist))
    dep = heads.parser(enc_dep.to(dev), lens1)
    od = O.dep_heads(enc_dep, lens1, sd, linearization=lin, distance=dist)
    _close(dep["arc_logp"], od["arc_logp"], rtol=1e-3, atol=1e-4)
    out = heads.annotate_batch(enc_dep.to(dev), lens1, enc_tag.to(dev), lens)
    wh, ws = C.mst_batch(od["arc_logp"].numpy(), lens1)
    assert not ws.any() and (out["mst_status"] == 0).all()
    assert np.array_equal(out["heads"].cpu().numpy(), wh)
    rel = np.zeros((B, L), np.int64)
    for b in range(B):
        for i in range(1, lens1[b]):
            rel[b, i] = int(od["label_argmax"][b, i, wh[b, i]])
    assert np.array_equal(out["rels"].cpu().numpy(), rel)


HEADLINE_RATES = {}  # (engine, ragged) -> measured rates, printed by the test (DESIGN.md quotes them)


@pytest.mark.parametrize("ragged", [False, True], ids=["full", "ragged"])
def test_headline_batch_properties(dev, engine, ragged):
    """B=256, L=128 (BASELINE config 3; `ragged`: lengths uniform in [32, 128], sorted descending): sentences are
    independent, so any sub-batch must reproduce the big batch bit-for-bit; decoded trees must be single-rooted and
    acyclic; decoders match the C oracle on the GPU's scores; and for 16 of the 256 sentences the arc log-probs, the
    winning label scores, the NER emissions and every decoded id are compared with the CPU oracle (sub-batch identity
    makes 16 sentences stand for the batch)."""
    from phonlp_amd import HeadWeights, TaskHeads, ops
    from phonlp_amd import synthetic as S
    B, L = 256, 128
    sd = S.make_state_dict()
    enc_dep, enc_tag = S.make_encoder_outputs(B, L)
    lens1 = S.make_sentlens(B, L, ragged)
    lens = [l - 1 for l in lens1]
    heads = TaskHeads(HeadWeights.from_state_dict(sd, dev))
    big = heads.annotate_batch(enc_dep.to(dev), lens1, enc_tag.to(dev), lens)
    dep = heads.parser(enc_dep.to(dev), lens1)
    sub = slice(40, 56)
    # sub-batch identity at the first, a middle and the last sixteen sentences (ragged: the longest, middling and
    # shortest ones) -- what lets sixteen sentences stand for the batch below
    for sl_ in (slice(0, 16), sub, slice(B - 16, B)):
        small = heads.annotate_batch(enc_dep[sl_].to(dev), lens1[sl_], enc_tag[sl_].to(dev), lens[sl_])
        for k in ("pos", "ner", "heads", "rels"):
            w_ = small[k].shape[1]  # a sub-batch of short sentences is as wide as the batch (same tensors, sliced)
            assert torch.equal(big[k][sl_][:, :w_], small[k]), (k, sl_)
    assert (big["mst_status"] == 0).all()
    h = big["heads"].cpu().numpy()
    for b in range(B):
        assert (h[b, 1:lens1[b]] == 0).sum() == 1  # exactly one child of ROOT
        assert not h[b, lens1[b]:].any()
        for i in range(1, lens1[b]):  # every node reaches ROOT
            v, steps = i, 0
            while v != 0 and steps <= L:
                v, steps = h[b, v], steps + 1
            assert v == 0
    wh, ws = C.mst_batch(dep["arc_logp"].cpu().numpy(), lens1)
    assert np.array_equal(h, wh)
    rel_want = np.take_along_axis(dep["label_argmax"].cpu().numpy(), h[:, :, None].astype(np.int64), 2)[:, :, 0]
    rel_want[:, 0] = 0
    for b in range(B):
        rel_want[b, lens1[b]:] = 0
    assert np.array_equal(big["rels"].cpu().numpy(), rel_want)
    # ---- 16 sentences against the CPU oracle: scores within the north star's 1e-3 relative (tighter here), ids exact
    sl1, sl = lens1[sub], lens[sub]
    od = O.dep_heads(enc_dep[sub], sl1, sd, keep_label_scores=True)
    ot = O.tagger_heads(enc_tag[sub], sl, sd)
    _close(dep["arc_logp"][sub], od["arc_logp"], rtol=2e-4, atol=2e-5)
    tag = heads.tagger(enc_tag[sub].to(dev), sl)
    _close(tag["ner_logits"], ot["ner_logits"], rtol=2e-4, atol=2e-5)
    assert torch.equal(tag["pos_pred"].cpu().long(), ot["pos_pred"])
    XF, T = ops.label_project(dep["hidden_out"][sub].contiguous(), heads.w.lab_W12, heads.w.lab_b12, heads.w.lab_Wbt,
                              heads.w.H, heads.w.O)
    hs = big["heads"][sub].contiguous()
    _, best = ops.label_at_heads(T, XF, hs, torch.tensor(sl1, dtype=torch.int32, device=dev), heads.w.H, heads.w.O)
    lab = od["label_scores"]  # [b, i, j, o]
    n_words = n_head = n_rel = n_ner = 0
    want_h, want_st = C.mst_batch(od["arc_logp"].numpy(), sl1)
    want_t, _ = C.viterbi_batch(ot["ner_logits"].numpy(), sd["crit_ner._transitions"].numpy(), sl)
    assert not want_st.any()
    hg, rg, ng = hs.cpu().numpy(), big["rels"][sub].cpu().numpy(), big["ner"][sub].cpu().numpy()
    for b in range(16):
        for i in range(1, sl1[b]):
            n_words += 1
            row = lab[b, i, hg[b, i]]
            assert abs(float(best.view(16, L)[b, i]) - float(row.max())) <= 2e-4 * max(1.0, abs(float(row.max())))
            n_head += int(hg[b, i] != want_h[b, i])
            if hg[b, i] == want_h[b, i]:
                n_rel += int(rg[b, i] != int(row.argmax()))
        n_ner += int((ng[b, :sl[b]] != want_t[b, :sl[b]]).sum())
    HEADLINE_RATES[(engine, ragged)] = (n_words, n_head, n_rel, n_ner)
    print(f"headline parity vs oracle, engine {engine}, ragged {ragged}: {n_words} words; differing heads {n_head}, "
          f"relations {n_rel}, NER tags {n_ner}")
    assert n_head == 0 and n_rel == 0 and n_ner == 0


def test_long_sentence_batch_properties(dev):
    """BASELINE config 5 (B=64, L=512): the whole path at four times the headline length (block-synchronous MST with
    the float64 matrix in the workspace, multi-tile label stage B, engine-2 split images).  Sub-batch == big batch bit
    for bit; heads == the C oracle's on the GPU's arc scores (sample of sentences); relations == the dense label
    scorer's arg-max at those heads except at provable near-ties."""
    from phonlp_amd import HeadWeights, TaskHeads, ops
    from phonlp_amd import synthetic as S
    B, L = 64, 512
    sd = S.make_state_dict()
    enc_dep, enc_tag = S.make_encoder_outputs(B, L)
    lens1 = S.make_sentlens(B, L, True)
    lens1[0] = L
    lens = [n - 1 for n in lens1]
    heads = TaskHeads(HeadWeights.from_state_dict(sd, dev))
    big = heads.annotate_batch(enc_dep.to(dev), lens1, enc_tag.to(dev), lens)
    assert (big["mst_status"] == 0).all()
    sub = slice(10, 18)
    small = heads.annotate_batch(enc_dep[sub].to(dev), lens1[sub], enc_tag[sub].to(dev), lens[sub])
    for k in ("pos", "ner", "heads", "rels"):
        assert torch.equal(big[k][sub], small[k]), k
    pick = [0, 1, 7, 20, 33, 63]
    dep = heads.parser(enc_dep[pick].to(dev), [lens1[i] for i in pick], want_label_scores=True)
    h = big["heads"].cpu().numpy()[pick]
    wh, ws = C.mst_batch(dep["arc_logp"].cpu().numpy(), [lens1[i] for i in pick])
    assert not ws.any() and np.array_equal(h, wh)
    rels = big["rels"].cpu()[pick]
    scores = dep["label_scores"].cpu()  # [b, i, o, j]
    for bi, b in enumerate(pick):
        for i in range(1, lens1[b]):
            sc = scores[bi, i, :, h[bi, i]]
            assert sc[rels[bi, i]] >= sc.max() - 1e-4 * max(1.0, float(sc.abs().max())), (b, i)
        assert (rels[bi, lens1[b]:] == 0).all() and rels[bi, 0] == 0


# ------------------------------------------------------------------------------------------ glue kernels (section 8f)
@pytest.mark.parametrize("B,L,H,O", [(6, 37, 24, 5), (16, 128, 400, 33)])
def test_label_at_heads_matches_dense_label_scorer(dev, engine, B, L, H, O):
    """rels from the at-heads contraction == gather of the dense scorer's arg-max at the same heads; where the two
    disagree the dense scores of the two candidates must be tied to fp32 rounding.  Winning scores agree with the
    dense tensor to 2e-4 relative (the north-star bar is 1e-3)."""
    from phonlp_amd import HeadWeights, ops
    from phonlp_amd import synthetic as S
    sd = S.make_state_dict(64, 12, H, 7, O, 5, seed=L)
    w = HeadWeights.from_state_dict(sd, dev)
    g = torch.Generator().manual_seed(B * L)
    hidden = torch.randn(B, L, 64 + 12, generator=g).to(dev)
    lens = torch.randint(2, L + 1, (B,), generator=g).to(torch.int32)
    lens[0] = L
    heads = torch.stack([torch.randint(0, int(n), (L,), generator=g) for n in lens]).to(torch.int32)
    lens_d, heads_d = lens.to(dev), heads.to(dev)
    dense_arg, scores = ops.deep_biaffine_label(hidden, w.lab_W12, w.lab_b12, w.lab_Wbt, H, O, True)  # [B,L,O,L]
    want = ops.gather_labels(dense_arg, heads_d, lens_d).cpu()
    XF, T = ops.label_project(hidden, w.lab_W12, w.lab_b12, w.lab_Wbt, H, O)
    got, best = ops.label_at_heads(T, XF, heads_d, lens_d, H, O)
    got, best, scores = got.cpu(), best.cpu(), scores.cpu()
    for b in range(B):
        for i in range(L):
            if not (1 <= i < int(lens[b])):
                assert got[b, i] == 0
                continue
            col = scores[b, i, :, int(heads[b, i])]
            tol = 2e-4 * float(col.abs().max()) + 1e-6
            assert abs(float(best[b, i]) - float(col.max())) <= tol
            if got[b, i] != want[b, i]:
                assert abs(float(col[got[b, i]]) - float(col[want[b, i]])) <= tol, (b, i)


@pytest.mark.parametrize("B,L,H,O", [(5, 37, 24, 5), (8, 128, 400, 33), (3, 130, 400, 7),
                                     # fused form: K16 = 64 / 96 (two / three k32 pairs of the wide kernel), K16 = 48 (not a
                                     # multiple of 32: the 256x128 kernel), several column tiles, relations not a multiple of 4
                                     (4, 90, 56, 6), (6, 70, 88, 9), (3, 50, 40, 4)])
def test_label_rows4_pair_matches_row_major(dev, B, L, H, O):
    """Engine 2's stage A in the 'rows interleaved by 4' layout is the row-major stage A bit for bit (same loop,
    different store), and label_at_heads_rows4 returns what label_at_heads returns on the same T: winning scores to
    fp32 rounding (another summation order), relations equal except where the two best scores are that close."""
    from phonlp_amd import HeadWeights, ops
    from phonlp_amd import synthetic as S
    sd = S.make_state_dict(64, 12, H, 7, O, 5, seed=L)
    w = HeadWeights.from_state_dict(sd, dev)
    g = torch.Generator().manual_seed(B * L + 1)
    hidden = torch.randn(B, L, 64 + 12, generator=g).to(dev)
    lens = torch.randint(2, L + 1, (B,), generator=g).to(torch.int32)
    lens[0] = L
    heads = torch.stack([torch.randint(0, int(n), (L,), generator=g) for n in lens]).to(torch.int32)
    lens_d, heads_d = lens.to(dev), heads.to(dev)
    KP = ops.kp(H)
    XF, _ = ops.label_project(hidden, w.lab_W12, w.lab_b12, w.lab_Wbt, H, O)
    x1p, x1i = ops.split_rows(XF[:, :KP])
    wp, wi = w.lab_Wbt_s
    T = ops.linear_split(x1p, x1i, wp, wi, None, KP, False, None, 0, None, 1)
    T4 = ops.label_stage_a_split(x1p, x1i, wp, wi, B, L, H, O)
    M = B * L
    M4 = (M + 3) // 4 * 4
    back = T4.view(M4 // 4, O * KP, 4).permute(0, 2, 1).reshape(M4, O * KP)[:M]
    assert torch.equal(back, T)
    # X1's ones column taken out of the product and added per column in the epilogue (one k16 stage less when
    # H % 16 == 0): the same sums in another order
    T4o = ops.label_stage_a_split(x1p, x1i, wp, wi, B, L, H, O, w.lab_ones_col)
    torch.testing.assert_close(T4o, T4, rtol=1e-5, atol=1e-5 * float(T4.abs().max()))
    if H % 16:
        assert torch.equal(T4o, T4)
    r0, b0 = ops.label_at_heads(T, XF, heads_d, lens_d, H, O)
    r4, b4 = ops.label_at_heads_rows4(T4, XF, heads_d, lens_d, H, O)
    # ... and the fused form (stage A + contraction in one kernel, T never stored) gives the same again
    rf, bf = ops.label_at_heads_fused(x1p, x1i, w.lab_Wbtc_s[0], w.lab_Wbtc_s[1], XF, heads_d, lens_d, H, O)
    for rx, bx in ((r4, b4), (rf, bf)):
        torch.testing.assert_close(bx, b0, rtol=1e-5, atol=1e-5)
        diff = (r0 != rx)
        if diff.any():  # only provable near-ties may differ: recompute both candidates' scores in float64
            Tm = T.view(B, L, O, KP).double()
            x2 = XF[:, KP:].view(B, L, KP).double()
            for b, i in diff.nonzero().tolist():
                sc = Tm[b, i] @ x2[b, heads[b, i]]
                assert abs(sc[r0[b, i]] - sc[rx[b, i]]) <= 1e-5 * sc.abs().max()


@pytest.mark.parametrize("B,L,E,TE,H,U,emb_scale", [(7, 33, 64, 12, 24, 7, 1.0), (7, 33, 64, 12, 24, 7, 4.0),
                                                     (9, 21, 64, 12, 24, 7, 50.0), (5, 19, 24, 8, 16, 7, 1.0),
                                                     (16, 128, 768, 100, 400, 24, 1.0), (4, 21, 64, 4, 24, 17, 1.0)])
def test_pos_head_split_image(dev, B, L, E, TE, H, U, emb_scale):
    """pos_head_split = pos_head followed by split_rows(feat), bit for bit on the live rows (padded rows: zero planes),
    with or without the fp32 feat output.  Under engine 2 the enc columns of the image are written before the
    embedding exists; emb_scale > 1 makes the embedding exceed the enc part's exponent on some / all rows, which
    takes the kernel's redo-the-row path."""
    from phonlp_amd import HeadWeights, ops
    from phonlp_amd import synthetic as S
    sd = S.make_state_dict(E, TE, H, U, 5, 5, seed=B)
    sd["upos_emb_matrix_dep"] = sd["upos_emb_matrix_dep"] * emb_scale
    w = HeadWeights.from_state_dict(sd, dev)
    g = torch.Generator().manual_seed(L)
    enc = torch.randn(B, L, E, generator=g).to(dev)
    lens = torch.randint(1, L + 1, (B,), generator=g).to(torch.int32)
    lens[0] = L
    lens_d = lens.to(dev)
    feat0, pos0, log0 = ops.pos_head(enc, lens_d, w.upos_hid_w, w.upos_hid_b, w.upos_clf_w, w.upos_clf_b, w.emb_dep)
    # the soft POS embedding itself against a torch product (fewer embedding columns than tags -- the last case -- once left
    # the tail kernel's cross-lane broadcasts reading inactive lanes: found by tests/fuzz_step.py)
    want_emb = torch.softmax(log0.double(), -1) @ w.emb_dep.double()
    live3 = (torch.arange(L, device=dev)[None, :] < lens_d[:, None])
    _close(feat0[..., E:][live3], want_emb.float()[live3], rtol=1e-5, atol=1e-6)
    p0, i0 = ops.split_rows(feat0.view(B * L, E + TE))
    live = (torch.arange(L)[None, :] < lens[:, None]).reshape(-1)
    for want_feat in (True, False):
        p1, i1, pos1, log1, feat1 = ops.pos_head_split(enc, lens_d, w.upos_hid_w, w.upos_hid_b, w.upos_clf_w,
                                                       w.upos_clf_b, w.emb_dep, want_feat)
        assert torch.equal(pos1, pos0) and torch.equal(log1, log0)
        assert torch.equal(p1.cpu()[live], p0.cpu()[live]) and torch.equal(i1.cpu()[live], i0.cpu()[live])
        assert (p1.cpu()[~live] == 0).all()
        if want_feat:
            assert torch.equal(feat1, feat0)


def _image_rows(planes, inv, K):
    """Split image -> the fp64 rows it stands for ((h + l) * inv_scale), first K columns."""
    rows = planes.shape[0]
    K16 = (K + 15) // 16 * 16
    img = planes.cpu().view(torch.float16).view(rows, K16 // 8, 2, 8).double()
    rec = (img[:, :, 0, :] + img[:, :, 1, :]).reshape(rows, K16) * inv.cpu().double().unsqueeze(1)
    assert (rec[:, K:] == 0).all()
    return rec[:, :K]


@pytest.mark.parametrize("B,L,E,H,U,enc_scale", [(7, 33, 64, 24, 7, 1.0), (9, 21, 64, 24, 7, 1e-3), (5, 19, 24, 16, 7, 1.0),
                                                  (3, 9, 20, 8, 3, 1.0), (4, 37, 64, 44, 32, 1.0), (6, 50, 64, 100, 17, 0.05),
                                                  (16, 128, 768, 400, 24, 1.0), (1, 32, 768, 400, 24, 1.0),
                                                  (2, 5, 8, 4, 1, 1.0), (3, 70, 16, 12, 5, 1e-6)])
def test_pos_head_probs_fused(dev, engine, B, L, E, H, U, enc_scale):
    """pos_head_split(emb=None) -- upos_clf inside the tail kernel, the tag probabilities where the embedding was -- against
    the two-launch form with emb = eye(U, Up): logits to fp32 rounding (another summation order), POS ids equal except
    provable near-ties, the image's rows equal to 1e-6 of the row maximum, padded rows zero planes and zero ids.
    enc_scale << 1 takes the redo-the-row path (the probabilities exceed the enc part's exponent), E % 8 != 0 the path
    without a presplit image, want_feat the fp32 rows; and a sentence's rows do not depend on the batch around it."""
    from phonlp_amd import HeadWeights, ops
    from phonlp_amd import synthetic as S
    TE0 = 12
    sd = S.make_state_dict(E, TE0, H, U, 5, 5, seed=B + U)
    w = HeadWeights.from_state_dict(sd, dev)
    g = torch.Generator().manual_seed(L)
    enc = (torch.randn(B, L, E, generator=g) * enc_scale).to(dev)
    lens = torch.randint(1, L + 1, (B,), generator=g).to(torch.int32)
    lens[0] = L
    lens_d = lens.to(dev)
    Up = ops.probs_width(U)
    eye = torch.eye(U, Up, device=dev)
    live = (torch.arange(L)[None, :] < lens[:, None]).reshape(-1)
    p0, i0, pos0, log0, feat0 = ops.pos_head_split(enc, lens_d, w.upos_hid_w, w.upos_hid_b, w.upos_clf_w, w.upos_clf_b, eye, True)
    rows0 = _image_rows(p0, i0, E + Up)
    for want_feat in (False, True):
        p1, i1, pos1, log1, feat1 = ops.pos_head_split(enc, lens_d, w.upos_hid_w, w.upos_hid_b, w.upos_clf_w, w.upos_clf_b,
                                                       None, want_feat)
        _close(log1, log0, rtol=1e-5, atol=2e-6)
        _near_tie_ok(pos1.cpu(), pos0.cpu(), log0.cpu(), 1e-5)
        assert (pos1.cpu().reshape(-1)[~live] == 0).all() and (log1.cpu().reshape(B * L, U)[~live] == 0).all()
        assert (p1.cpu()[~live] == 0).all() and (i1.cpu()[~live] == 1).all()
        rows1 = _image_rows(p1, i1, E + Up)
        tol = 1e-6 * rows0.abs().max(1, keepdim=True).values.clamp_min(1e-30)
        assert ((rows1 - rows0).abs()[live] <= tol[live]).all()
        if want_feat:
            _close(feat1, feat0, rtol=1e-5, atol=1e-6)
            # the image is split_rows of the fp32 rows, bit for bit
            pf, iv = ops.split_rows(feat1.view(B * L, E + Up))
            assert torch.equal(pf.cpu()[live], p1.cpu()[live]) and torch.equal(iv.cpu()[live], i1.cpu()[live])
    # the fp32 entry point (phonlp_pos_head) in the same form: the rows the split form wrote beside its image
    feat3, pos3, log3 = ops.pos_head(enc, lens_d, w.upos_hid_w, w.upos_hid_b, w.upos_clf_w, w.upos_clf_b, None)
    if ops.get_gemm_engine() == 2 and E % 8 == 0:  # (other engines: upos_hid itself runs on another kernel in the two entry points)
        assert torch.equal(feat3, feat1) and torch.equal(pos3, pos1) and torch.equal(log3, log1)
    else:
        _close(feat3, feat1, rtol=1e-5, atol=2e-6)
    # sub-batch identity: the first two sentences alone give the same bits
    nb = min(B, 2)
    p2, i2, pos2, log2, _ = ops.pos_head_split(enc[:nb].contiguous(), lens_d[:nb].contiguous(), w.upos_hid_w, w.upos_hid_b,
                                               w.upos_clf_w, w.upos_clf_b, None, False)
    assert torch.equal(p2, p1[:nb * L]) and torch.equal(i2, i1[:nb * L]) and torch.equal(pos2, pos1[:nb])
    assert torch.equal(log2, log1[:nb])


def test_probs_forms_refuse_what_they_cannot_do(dev):
    """The emb=None forms hold a token's tags in one 32 x 32 accumulator tile: more than 32 tags are refused with an error
    (the caller then passes an embedding matrix, the two-launch form), never computed wrongly."""
    from phonlp_amd import HeadWeights, ops
    from phonlp_amd import synthetic as S
    E, TE, H, U, T = 64, 12, 24, 40, 5
    w = HeadWeights.from_state_dict(S.make_state_dict(E, TE, H, U, 5, T, seed=1), dev)
    enc = torch.randn(2, 9, E, device=dev)
    lens = torch.tensor([9, 4], dtype=torch.int32, device=dev)
    with pytest.raises(Exception, match="emb == NULL"):
        ops.pos_head_split(enc, lens, w.upos_hid_w, w.upos_hid_b, w.upos_clf_w, w.upos_clf_b, None, False)
    with pytest.raises(Exception, match="emb == NULL"):
        ops.tagger_head(enc, lens, w.tag_Whn, w.tag_bhn, w.upos_clf_w, w.upos_clf_b, None, w.ner_w_tags, w.ner_b)
    # and the whole step still runs at that size (annotate_batch picks the two-launch forms)
    from phonlp_amd import TaskHeads
    out = TaskHeads(w).annotate_batch(torch.randn(2, 10, E, device=dev), [10, 5], enc, [9, 4])
    torch.cuda.synchronize()
    assert int((out["mst_status"] != 0).sum()) == 0


@pytest.mark.parametrize("B,L,E,TE,H,U,T", [(7, 33, 64, 12, 24, 7, 5), (5, 19, 24, 8, 16, 7, 3), (2, 4, 20, 4, 8, 3, 70),
                                             (16, 127, 768, 100, 400, 24, 9), (6, 40, 64, 4, 24, 17, 3)])
def test_fused_tagger_head(dev, engine, B, L, E, TE, H, U, T):
    """ops.tagger_head (upos_hid and the enc columns of ner_tag_clf in one GEMM, embedding columns added in the tail,
    feat never formed) against pos_head + ner_emission and against the oracle (ref annotate_model.py:132-150):
    POS ids equal, emissions to fp32 rounding, padded rows exact zeros."""
    from phonlp_amd import HeadWeights, TaskHeads, ops
    from phonlp_amd import synthetic as S
    sd = S.make_state_dict(E, TE, H, U, 5, T, seed=B + T)
    w = HeadWeights.from_state_dict(sd, dev)
    g = torch.Generator().manual_seed(L)
    enc = torch.randn(B, L, E, generator=g)
    lens = torch.randint(1, L + 1, (B,), generator=g).sort(descending=True)[0].to(torch.int32)
    lens[0] = L
    lens_d = lens.to(dev)
    pos, ner, logits = ops.tagger_head(enc.to(dev), lens_d, w.tag_Whn, w.tag_bhn, w.upos_clf_w, w.upos_clf_b, w.emb_ner,
                                       w.ner_w_emb, w.ner_b)
    ref = TaskHeads(w).tagger(enc.to(dev), lens_d)
    assert torch.equal(pos, ref["pos_pred"])
    _close(logits, ref["upos_logits"], rtol=1e-5, atol=1e-6)
    _close(ner, ref["ner_logits"], rtol=1e-5, atol=2e-6)
    ot = O.tagger_heads(enc, lens.tolist(), sd)
    _close(ner, ot["ner_logits"], rtol=1e-4, atol=1e-5)
    _near_tie_ok(pos.cpu(), ot["pos_pred"], ot["upos_logits"], 1e-5)
    live = torch.arange(L)[None, :] < lens[:, None]
    assert (ner.cpu()[~live] == 0).all() and (pos.cpu()[~live] == 0).all()
    if U <= 32:
        # emb=None: upos_emb_matrix_ner folded into ner_tag_clf's embedding columns at load, upos_clf inside the tail kernel
        # (its 400-term sums in another order than the dense layer's); and a sentence's rows do not depend on the batch
        pos2, ner2, logits2 = ops.tagger_head(enc.to(dev), lens_d, w.tag_Whn, w.tag_bhn, w.upos_clf_w, w.upos_clf_b, None,
                                              w.ner_w_tags, w.ner_b)
        _near_tie_ok(pos2.cpu(), ref["pos_pred"].cpu(), ref["upos_logits"].cpu(), 1e-5)
        _close(logits2, ref["upos_logits"], rtol=1e-5, atol=2e-6)
        _close(ner2, ref["ner_logits"], rtol=1e-5, atol=4e-6)
        _close(ner2, ot["ner_logits"], rtol=1e-4, atol=1e-5)
        assert (ner2.cpu()[~live] == 0).all() and (pos2.cpu()[~live] == 0).all() and (logits2.cpu()[~live] == 0).all()
        nb = min(B, 2)
        pos3, ner3, logits3 = ops.tagger_head(enc[:nb].to(dev).contiguous(), lens_d[:nb].contiguous(), w.tag_Whn, w.tag_bhn,
                                              w.upos_clf_w, w.upos_clf_b, None, w.ner_w_tags, w.ner_b)
        assert torch.equal(pos3, pos2[:nb]) and torch.equal(ner3, ner2[:nb]) and torch.equal(logits3, logits2[:nb])


def test_streaming_batches_match_joined(dev):
    """annotate_batch(join=False) on back-to-back batches (tails on the finishing stream, next batch's GEMMs not
    waiting for the MST) gives exactly the outputs of the joined form."""
    from phonlp_amd import HeadWeights, TaskHeads
    from phonlp_amd import synthetic as S
    heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(64, 12, 48, 7, 6, 5, seed=3), dev))
    batches = []
    for k in range(4):
        B, L = 24 + 8 * k, 40 + 7 * k
        ed, et = S.make_encoder_outputs(B, L, 64, seed=50 + k)
        lens1 = S.make_sentlens(B, L, ragged=True, seed=k)
        batches.append((ed.to(dev), lens1, et.to(dev), [n - 1 for n in lens1]))
    want = [heads.annotate_batch(*b) for b in batches]
    torch.cuda.synchronize()
    got = [heads.annotate_batch(*b, join=False) for b in batches]  # enqueued back to back, no host sync between
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        for k in ("pos", "ner", "heads", "rels", "mst_status"):
            assert torch.equal(w[k].cpu(), g[k].cpu()), k


@pytest.mark.parametrize("B,S,L,E,idx_dtype", [(5, 40, 19, 32, torch.int64), (64, 200, 128, 768, torch.int32)])
def test_gather_first_subword(dev, B, S, L, E, idx_dtype):
    from phonlp_amd import ops
    g = torch.Generator().manual_seed(B + S)
    enc = torch.randn(B, S, E, generator=g)
    idx = torch.randint(0, S, (B, L), generator=g).to(idx_dtype)
    lens = torch.randint(1, L + 1, (B,), generator=g).to(torch.int32)
    want = torch.stack([enc[b].index_select(0, idx[b].long()) for b in range(B)])  # ref annotate_model.py:120-123
    got = ops.gather_first_subword(enc.to(dev), idx.to(dev), None)
    assert torch.equal(got.cpu(), want)  # a copy: bit-exact
    got = ops.gather_first_subword(enc.to(dev), idx.to(dev), lens.to(dev))
    mask = (torch.arange(L).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(-1)
    assert torch.equal(got.cpu(), want * mask)


def test_eval_counts_vs_python(dev):
    from phonlp_amd import ops
    from phonlp_amd.sharding import eval_counts
    g = torch.Generator().manual_seed(3)
    B, Ld = 37, 50
    Lt = Ld - 1
    ri = lambda hi, shape: torch.randint(0, hi, shape, generator=g, dtype=torch.int32)
    pos, ner, gpos, gner = ri(4, (B, Lt)), ri(3, (B, Lt)), ri(4, (B, Lt)), ri(3, (B, Lt))
    heads, rels, ghead, grel = ri(5, (B, Ld)), ri(3, (B, Ld)), ri(5, (B, Ld)), ri(3, (B, Ld))
    lens_dep = torch.randint(2, Ld + 1, (B,), generator=g, dtype=torch.int32)
    lens_tag = lens_dep - 1
    status = (torch.rand(B, generator=g) < 0.1).to(torch.int32)
    want = eval_counts(pos, ner, heads, rels, gpos, gner, ghead, grel, lens_tag, lens_dep, status)
    d = lambda t: t.to(dev)
    got = ops.eval_counts(d(pos), d(ner), d(heads), d(rels), d(gpos), d(gner), d(ghead), d(grel), d(lens_tag),
                          d(lens_dep), d(status))
    assert got.cpu().tolist() == want.tolist()


def test_ner_entity_counts(dev, golden):
    """Entity-level NER counts kernel: exact against the reference's score_by_entity fixtures, and against the
    oracle on a headline-sized random batch."""
    from phonlp_amd import ops
    from phonlp_amd.sharding import tag_tables
    z = golden("ner_entities.npz")
    vocab = [str(v) for v in z["vocab"]]
    kind, typ = (t.to(dev) for t in tag_tables(vocab))
    for c in range(int(z["n_cases"])):
        got = ops.ner_entity_counts(torch.from_numpy(z[f"c{c}_pred"]).to(dev), torch.from_numpy(z[f"c{c}_gold"]).to(dev),
                                    torch.from_numpy(z[f"c{c}_lens"]).to(dev), kind, typ)
        assert got.cpu().tolist() == z[f"c{c}_counts"].tolist(), c
    g = torch.Generator().manual_seed(5)
    B, L = 256, 127
    gold = torch.where(torch.rand(B, L, generator=g) < 0.6, torch.tensor(4), torch.randint(0, len(vocab), (B, L), generator=g))
    pred = torch.where(torch.rand(B, L, generator=g) < 0.2, torch.randint(0, len(vocab), (B, L), generator=g), gold)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    got = ops.ner_entity_counts(pred.to(torch.int32).to(dev), gold.to(torch.int32).to(dev), lens.to(torch.int32).to(dev),
                                kind, typ).cpu().tolist()
    ps = [[vocab[t] for t in pred[b, :lens[b]].tolist()] for b in range(B)]
    gs = [[vocab[t] for t in gold[b, :lens[b]].tolist()] for b in range(B)]
    assert got == list(O.ner_entity_counts(ps, gs))
    # long sentences (the kernel walks them 64 positions at a time, entities and ignored tags across the chunk borders),
    # every tag equally likely, lengths on and around the borders
    for seed, (B, L) in enumerate([(40, 300), (9, 64), (9, 65), (5, 129)]):
        g = torch.Generator().manual_seed(100 + seed)
        gold = torch.randint(0, len(vocab), (B, L), generator=g)
        pred = torch.where(torch.rand(B, L, generator=g) < 0.3, torch.randint(0, len(vocab), (B, L), generator=g), gold)
        lens = torch.randint(1, L + 1, (B,), generator=g)
        lens[0] = L
        lens[1] = min(L, 64)
        got = ops.ner_entity_counts(pred.to(torch.int32).to(dev), gold.to(torch.int32).to(dev), lens.to(torch.int32).to(dev),
                                    kind, typ).cpu().tolist()
        ps = [[vocab[t] for t in pred[b, :lens[b]].tolist()] for b in range(B)]
        gs = [[vocab[t] for t in gold[b, :lens[b]].tolist()] for b in range(B)]
        assert got == list(O.ner_entity_counts(ps, gs)), (B, L)


def test_folded_dep_hid_matches_unfolded(dev, engine):
    """dep_hid folded into the scorers' first layer (annotate_batch's form) against the layer-by-layer form
    (parser()): arc log-probs to 2e-4, identical non-finite pattern, and the padded rows of the projected features
    bit-equal (they are relu(original bias) in both forms, whatever the padded input rows hold)."""
    from phonlp_amd import HeadWeights, ops
    from phonlp_amd import synthetic as S
    B, L = 12, 70
    w = HeadWeights.from_state_dict(S.make_state_dict(seed=11), dev)
    lens = torch.tensor(S.make_sentlens(B, L, ragged=True, seed=4), dtype=torch.int32, device=dev)
    feat = torch.randn(B, L, w.dep_hid_w.shape[1], generator=torch.Generator().manual_seed(2)).to(dev)
    hidden = ops.dep_hidden(feat, lens, w.dep_hid_w, w.dep_hid_b)
    garbage = feat.clone()
    for b in range(B):
        garbage[b, int(lens[b]):] = 1e3  # padded input rows must not matter in the folded form
    arc_u, _ = ops.deep_biaffine_arc(hidden, w.arc_W12, w.arc_b12, w.arc_Wbt, w.H, w.n_arc_scorers, True, True, False)
    arc_f, _ = ops.deep_biaffine_arc(garbage, w.arc_W12f, w.arc_b12f, w.arc_Wbt, w.H, w.n_arc_scorers, True, True, False,
                                     lens, w.arc_b12)
    _close(arc_f, arc_u, rtol=2e-4, atol=2e-4)
    XFu, Tu = ops.label_project(hidden, w.lab_W12, w.lab_b12, w.lab_Wbt, w.H, w.O)
    XFf, Tf = ops.label_project(garbage, w.lab_W12f, w.lab_b12f, w.lab_Wbt, w.H, w.O, lens, w.lab_b12)
    XFu, XFf = XFu.view(B, L, -1).cpu(), XFf.view(B, L, -1).cpu()
    for b in range(B):
        n = int(lens[b])
        assert torch.equal(XFu[b, n:], XFf[b, n:])
        torch.testing.assert_close(XFf[b, :n], XFu[b, :n], rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(Tf.cpu(), Tu.cpu(), rtol=1e-3, atol=1e-4)
