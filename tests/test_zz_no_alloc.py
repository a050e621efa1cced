"""phonlp_crf_nll_grad_static -- the entry point that allocates nothing (scratch in a file-scope device array) --
through the C-ABI against phonlp_crf_nll_grad, bit for bit, on two calls in a row, the second with a smaller input
(the shared array keeps the first call's contents)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(lib, check, B, L, T, seed, first_on_device=False):
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
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    check(lib.phonlp_crf_nll_grad_static(p(logits), p(trans), None, p(lens), p(gold), B, L, T, 1.0 / B, p(d2), p(t2), st),
          "phonlp_crf_nll_grad_static")
    torch.cuda.synchronize()
    if not first_on_device:  # the first call on a device looks the array's address up (which may load the module's
        # device variables); every later call must leave the device's free memory exactly as it found it
        assert torch.cuda.mem_get_info()[0] == free_before, "phonlp_crf_nll_grad_static changed the free device memory"
    assert torch.equal(d1, d2) and torch.equal(t1, t2), (B, L, T)
    assert torch.isfinite(d2).all() and float(t2.abs().sum()) > 0


def test_no_alloc_entry_point_matches_workspace_entry_point():
    assert torch.cuda.is_available()
    from phonlp_amd._lib import check, lib
    L_ = lib()
    assert L_.phonlp_crf_nll_grad_static_max_bytes() == 4 << 20
    _both(L_, check, 256, 127, 9, seed=1, first_on_device=True)    # headline tag set, register form (T <= 16)
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


def test_no_alloc_entry_point_under_graph_capture():
    """What the entry point is for: a caller under HIP graph capture.  After one call outside capture on this device
    (the address look-up: include/phonlp_heads.h), the call is captured and the replay's result equals the eager one."""
    assert torch.cuda.is_available()
    from phonlp_amd._lib import check, lib
    L_ = lib()
    dev = torch.device("cuda:0")
    B, L, T = 48, 30, 9
    g = torch.Generator().manual_seed(11)
    logits = torch.relu(torch.randn(B, L, T, generator=g)).to(dev)
    trans = (torch.randn(T, T, generator=g) * 0.3).to(dev)
    lens = torch.randint(1, L + 1, (B,), generator=g, dtype=torch.int32).to(dev)
    gold = torch.randint(0, T, (B, L), generator=g, dtype=torch.int32).to(dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    d_e, t_e = torch.zeros_like(logits), torch.zeros_like(trans)
    d_g, t_g = torch.zeros_like(logits), torch.zeros_like(trans)

    def call(d, t):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(L_.phonlp_crf_nll_grad_static(p(logits), p(trans), None, p(lens), p(gold), B, L, T, 1.0 / B, p(d), p(t), st),
              "phonlp_crf_nll_grad_static")

    call(d_e, t_e)  # eager, outside capture (also the per-device set-up if this test runs alone)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph, stream=s):
            call(d_g, t_g)
    torch.cuda.current_stream().wait_stream(s)
    d_g.zero_(); t_g.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(d_e, d_g) and torch.equal(t_e, t_g)
