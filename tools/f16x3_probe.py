"""Accuracy (vs an fp64 product) and speed of the f16x3 split-image dense layer against engines 0/1.
usage: python tools/f16x3_probe.py [reps]"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("PHONLP_X3_ABLATION"):  # the ablation branches only exist in the -DPHONLP_ABLATION build (timing only)
    from phonlp_amd import _lib as _l
    _l.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libphonlp_heads_ablation.so")
from phonlp_amd import ops
from phonlp_amd._lib import lib, check
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr())
shapes = [(32768, 13464, 401, 408), (32768, 2448, 868, 868), (32768, 816, 868, 868), (4096, 4096, 4096, 4096), (1000, 333, 100, 100)]
if os.environ.get("PROBE_SHAPE"):  # "M N K ldk"
    shapes = [tuple(int(v) for v in os.environ["PROBE_SHAPE"].split())]
only_x3 = os.environ.get("PROBE_ONLY_X3") == "1"
for (M, N, K, ldk) in shapes:
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.zeros(M, ldk); x[:, :K] = torch.randn(M, K, generator=g) * torch.exp(2 * torch.randn(M, 1, generator=g))
    w = torch.zeros(N, ldk); w[:, :K] = torch.randn(N, K, generator=g) * 0.05
    x, w = x.to(dev), w.to(dev)
    rows = torch.randint(0, M, (256,), generator=g).to(dev)
    ref = x[rows].double() @ w.double().t()
    scale = (x[rows].double().abs() @ w.double().abs().t())
    def timeit(run):
        run(); run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): run()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    def report(name, y, dt):
        err = (y[rows].double() - ref).abs()
        print(f"M={M} N={N} K={K} {name}: {dt*1e3:.3f} ms {2.0*M*N*K/dt/1e12:.1f} TFLOP/s-eq  "
              f"max|err|/sum|a||b| = {(err/scale).max().item():.3e}  rms rel = {(err.pow(2).mean().sqrt()/ref.pow(2).mean().sqrt()).item():.3e}",
              flush=True)
    y = torch.empty(M, N, device=dev)
    for eng in (() if only_x3 else (0, 1)):
        ops.set_gemm_engine(eng)
        run = lambda: check(lib().phonlp_linear(P(x), ldk, P(w), ldk, None, P(y), N, M, N, ldk, 0, None, 0, None), "linear")
        dt = timeit(run)
        report(f"engine={eng}", y, dt)
    ops.set_gemm_engine(2)
    wp, wi = ops.split_rows(w)
    dts = timeit(lambda: ops.split_rows(x))
    xp, xi = ops.split_rows(x)
    holder = {}
    def run2(): holder["y"] = ops.linear_split(xp, xi, wp, wi, None, ldk, False, None, 0, None, 0)
    dt = timeit(run2)
    report(f"f16x3 (split of x: {dts*1e3:.3f} ms)", holder["y"], dt)
