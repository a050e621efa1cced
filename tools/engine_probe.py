"""Accuracy (vs an fp64 product) and speed of the two GEMM engines on the dense-layer kernel.
usage: python tools/engine_probe.py [reps]"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phonlp_amd._lib import lib, check
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr())
for (M, N, K, ldk) in [(32768, 13233, 401, 408), (32768, 401, 868, 868), (32768, 868, 868, 868), (4096, 4096, 4096, 4096)]:
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.zeros(M, ldk); x[:, :K] = torch.randn(M, K, generator=g) * torch.exp(2 * torch.randn(M, 1, generator=g))
    w = torch.zeros(N, ldk); w[:, :K] = torch.randn(N, K, generator=g) * 0.05
    x, w = x.to(dev), w.to(dev)
    y = torch.empty(M, N, device=dev)
    rows = torch.randint(0, M, (256,), generator=g).to(dev)
    ref = x[rows].double() @ w.double().t()
    scale = (x[rows].double().abs() @ w.double().abs().t())  # sum |a||b|: the natural error scale
    for eng in (0, 1):
        check(lib().phonlp_set_gemm_engine(eng), "engine")
        run = lambda: check(lib().phonlp_linear(P(x), ldk, P(w), ldk, None, P(y), N, M, N, ldk, 0, None, 0, None), "linear")
        run(); run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        err = (y[rows].double() - ref).abs()
        print(f"M={M} N={N} K={K} engine={eng}: {dt*1e3:.3f} ms {2.0*M*N*K/dt/1e12:.1f} TFLOP/s  "
              f"max|err|/sum|a||b| = {(err/scale).max().item():.3e}  rel-to-ref rms = {(err.pow(2).mean().sqrt()/ref.pow(2).mean().sqrt()).item():.3e}",
              flush=True)
