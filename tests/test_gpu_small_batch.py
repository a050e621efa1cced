"""Small batches (BASELINE config 2, the reference's default batch_size=1) run on kernels of their own -- the dense layers on
64 x 64 tiles with a deep ring (linear_f16x3_small_kernel), the arc scorers as a workgroup per (sentence, scorer) with a
combine pass (phonlp_arc_scores_split_ws) -- which must give the BITS of the forms the large batches use: a sentence's
scores do not depend on the batch it is decoded in."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("ragged", [False, True], ids=["full", "ragged"])
@pytest.mark.parametrize("L", [32, 64, 100, 128])
def test_small_batches_give_the_big_batch_bits(dev, L, ragged):
    from phonlp_amd import HeadWeights, TaskHeads
    from phonlp_amd._lib import lib
    from phonlp_amd import synthetic as S
    B = 200  # 3 B > 512: the arc kernel's one-workgroup-per-sentence form; B * L words: the large dense-layer kernels
    assert lib().phonlp_arc_scores_split_ws_bytes(B, L, 3) == 0
    assert lib().phonlp_arc_scores_split_ws_bytes(32, L, 3) == 2 * 32 * L * L * 4
    sd = S.make_state_dict()
    enc_dep, enc_tag = S.make_encoder_outputs(B, L)
    lens1 = S.make_sentlens(B, L, ragged)
    lens = [n - 1 for n in lens1]
    heads = TaskHeads(HeadWeights.from_state_dict(sd, dev))
    big = heads.annotate_batch(enc_dep.to(dev), lens1, enc_tag.to(dev), lens)
    for sl_ in (slice(0, 1), slice(B - 1, B), slice(7, 12), slice(100, 132)):
        n1, n = lens1[sl_], lens[sl_]
        small = heads.annotate_batch(enc_dep[sl_].to(dev), n1, enc_tag[sl_].to(dev), n)
        w_ = small["heads"].shape[1]
        for k in ("pos", "ner", "heads", "rels"):
            assert torch.equal(big[k][sl_][:, :small[k].shape[1]], small[k]), (k, sl_)
        for bi, b in enumerate(range(*sl_.indices(B))):  # live dependents' rows, all head columns of the (narrower) sub-batch
            got = small["arc_logp"][bi, :n1[bi], :w_]
            want = big["arc_logp"][b, :n1[bi], :w_]
            if not ragged:
                assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (sl_, b)
            else:  # the big ragged batch runs PACKED (padded head columns read one pad row), the sub-batch may not: last bits
                fin = torch.isfinite(want)
                assert torch.equal(fin, torch.isfinite(got))
                assert (got[fin] - want[fin]).abs().max() <= 2e-5


def test_arc_workspace_form_matches_plain_entry(dev):
    """phonlp_arc_scores_split_ws with and without its workspace on the same image: same bits (three scorers, with and without the
    linearization / distance terms)."""
    from phonlp_amd import HeadWeights, ops
    from phonlp_amd._lib import lib
    from phonlp_amd import synthetic as S
    from phonlp_amd.ops import _p, _stream, check
    w = HeadWeights.from_state_dict(S.make_state_dict(), dev)
    for B, L in ((1, 32), (3, 77), (32, 64), (9, 128)):
        g = torch.Generator(device=dev); g.manual_seed(B * 131 + L)
        feat = torch.relu(torch.randn(B * L, 792, device=dev, generator=g))
        fp, fi = ops.split_rows(feat)
        lens = torch.full((B,), L, dtype=torch.int32, device=dev)
        xfp, xfi = ops.linear_split_image(fp, fi, w.arc_W12q_s[0], w.arc_W12q_s[1], w.arc_b12q, 792, True, lens, L,
                                          w.arc_b12q_pad, w.arc_bound[0], w.arc_bound[1], 3)
        for use_lin, use_dist in ((True, True), (True, False), (False, False)):
            outs = []
            for with_ws in (False, True):
                out = torch.full((B, L, L), float("nan"), device=dev)
                nb = int(lib().phonlp_arc_scores_split_ws_bytes(B, L, 3)) if with_ws else 0
                assert not with_ws or nb > 0
                ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
                check(lib().phonlp_arc_scores_split_ws(_p(xfp), _p(xfi), xfp.shape[0], B, L, w.H, 3, int(use_lin), int(use_dist),
                                                       _p(w.arc_Wbq_s[0]), _p(w.arc_Wbq_s[1]), None, None, None, 0.0, _p(out),
                                                       _p(ws) if with_ws else None, nb, _stream()), "arc_scores_split_ws")
                outs.append(out)
            torch.cuda.synchronize()
            assert torch.isfinite(outs[0][:, :, :]).any()
            assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)), (B, L, use_lin, use_dist)
    # packed rows (ragged batch: off / sentlens / the pad row)
    for B, L in ((5, 64), (32, 96), (12, 128)):
        g = torch.Generator(device=dev); g.manual_seed(B * 17 + L)
        n = torch.randint(2, L + 1, (B,), generator=torch.Generator().manual_seed(B + L)).to(torch.int32)
        n[0] = L
        rows = int(n.sum())
        off = (torch.cumsum(n, 0) - n).to(torch.int32).to(dev)
        feat = torch.relu(torch.randn(rows, 792, device=dev, generator=g))
        fp, fi = ops.split_rows(feat)
        xfp, xfi = ops.linear_split_image(fp, fi, w.arc_W12q_s[0], w.arc_W12q_s[1], w.arc_b12q, 792, True, None, 0, None,
                                          w.arc_bound[0], w.arc_bound[1], 3)
        outs = []
        for with_ws in (False, True):
            out = torch.zeros((B, L, L), device=dev)
            nb = int(lib().phonlp_arc_scores_split_ws_bytes(B, L, 3)) if with_ws else 0
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
            check(lib().phonlp_arc_scores_split_ws(_p(xfp), _p(xfi), rows, B, L, w.H, 3, 1, 1, _p(w.arc_Wbq_s[0]),
                                                   _p(w.arc_Wbq_s[1]), _p(off), _p(n.to(dev)), _p(w.arc_pad_img[0]),
                                                   float(w.arc_pad_img[1]), _p(out), _p(ws) if with_ws else None, nb, _stream()),
                  "arc_scores_split_ws")
            outs.append(out)
        torch.cuda.synchronize()
        assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)), (B, L, "packed")
        assert (outs[0][0, :, 1] != 0).all()
    # a workspace that is too small is refused
    B, L = 4, 32
    rc = lib().phonlp_arc_scores_split_ws(_p(xfp), _p(xfi), xfp.shape[0], B, L, w.H, 3, 1, 1, _p(w.arc_Wbq_s[0]), _p(w.arc_Wbq_s[1]),
                                          None, None, None, 0.0, _p(out), _p(ws), 16, _stream())
    assert rc != 0


@pytest.mark.parametrize("N,K", [(17, 100), (401, 800), (808, 48), (64, 16), (130, 1000)])
def test_dense_layer_rows_do_not_depend_on_the_batch(dev, N, K):
    """Engine-2 dense layer (phonlp_linear_split): the rows of a product over a few words (64 x 64 tiles, deep ring) carry the bits
    of the same rows inside a product over many (256 x 128 / transposed 256 x 256 tiles), for odd N and K, with bias + ReLU, a
    padding mask and its pad value."""
    from phonlp_amd import ops
    big_m = 100352 if N < 100 else 40960  # > 1280 tiles of 64 x 64: the large kernels
    L = 64
    g = torch.Generator(device=dev); g.manual_seed(N * 1000 + K)
    K4 = (K + 3) // 4 * 4
    X = torch.randn(big_m, K4, device=dev, generator=g) * torch.rand(big_m, 1, device=dev, generator=g) * 4
    W = torch.randn(N, K4, device=dev, generator=g)
    if K4 != K:
        X[:, K:] = 0
        W[:, K:] = 0
    bias = torch.randn(N, device=dev, generator=g)
    padv = torch.randn(N, device=dev, generator=g)
    xp, xi = ops.split_rows(X)
    wp, wi = ops.split_rows(W)
    lens_big = torch.randint(1, L + 1, (big_m // L,), device=dev, generator=g).to(torch.int32)
    for relu, masked in ((True, False), (False, True), (True, True)):
        sl = lens_big if masked else None
        y_big = ops.linear_split(xp, xi, wp, wi, bias, K4, relu, sl, L if masked else 0, padv if masked else None, 0)
        for m in (1, 5, 63, 64, 65, 200, 1024):
            rows = (m + L - 1) // L * L if masked else m  # whole sentences under a mask
            y = ops.linear_split(xp[:rows].contiguous(), xi[:rows].contiguous(), wp, wi, bias, K4, relu,
                                 lens_big[:rows // L].contiguous() if masked else None, L if masked else 0,
                                 padv if masked else None, 0)
            assert torch.equal(y.view(torch.int32), y_big[:rows].view(torch.int32)), (N, K, relu, masked, m)
