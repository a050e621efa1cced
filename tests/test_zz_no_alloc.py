"""phonlp_crf_nll_grad_static -- the entry point that allocates nothing (scratch in a file-scope device array) --
through the C-ABI against phonlp_crf_nll_grad, bit for bit, on two calls in a row, the second with a smaller input
(the shared array keeps the first call's contents)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(lib, check, B, L, T, seed):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    logits = torch.relu(torch.randn(B, L, T, generator=g)).to(dev)
    trans = (torch.randn(T, T, generator=g) * 0.3).to(dev)
    lens = torch.randint(1, L + 1, (B,), generator=g, dtype=torch.int32)
    lens[0] = L
    lens = lens.to(dev)
    gold = torch.randint(0, T, (B, L), generator=g, dtype=torch.int32).to(dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nb = lib.phonlp_crf_nll_grad_ws_bytes(B, T)
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
    d1, t1 = torch.full_like(logits, 7.0), torch.full_like(trans, 7.0)
    check(lib.phonlp_crf_nll_grad(p(logits), p(trans), None, p(lens), p(gold), B, L, T, 1.0 / B, p(d1), p(t1), p(ws), nb, st),
          "phonlp_crf_nll_grad")
    d2, t2 = torch.full_like(logits, -7.0), torch.full_like(trans, -7.0)
    check(lib.phonlp_crf_nll_grad_static(p(logits), p(trans), None, p(lens), p(gold), B, L, T, 1.0 / B, p(d2), p(t2), st),
          "phonlp_crf_nll_grad_static")
    torch.cuda.synchronize()
    assert torch.equal(d1, d2) and torch.equal(t1, t2), (B, L, T)
    assert torch.isfinite(d2).all() and float(t2.abs().sum()) > 0


def test_no_alloc_entry_point_matches_workspace_entry_point():
    assert torch.cuda.is_available()
    from phonlp_amd._lib import check, lib
    L_ = lib()
    assert L_.phonlp_crf_nll_grad_static_max_bytes() == 4 << 20
    _both(L_, check, 256, 127, 9, seed=1)    # headline tag set, register form (T <= 16)
    _both(L_, check, 37, 40, 9, seed=2)      # smaller, straight after: stale shares of the first call stay unread
    _both(L_, check, 64, 33, 24, seed=3)     # LDS form (T > 16)
    _both(L_, check, 5, 8, 24, seed=4)
    # beyond the static maximum: a clear error, nothing launched
    dev = torch.device("cuda:0")
    x = torch.zeros(2048, 2, 32, device=dev)
    tr = torch.zeros(32, 32, device=dev)
    gold = torch.zeros(2048, 2, dtype=torch.int32, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = L_.phonlp_crf_nll_grad_static(p(x), p(tr), None, None, p(gold), 2048, 2, 32, 1.0, p(x.clone()), p(tr.clone()), None)
    assert rc == -4 and b"exceeds the static scratch" in L_.phonlp_last_error()
