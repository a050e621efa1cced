"""Runs one shape of the fp32 MFMA dense-layer kernel repeatedly (for rocprofv3 / event timing).
usage: python tools/gemm_probe.py M N K [reps] [tag]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
from phonlp_amd._lib import lib, check
M, N, K = (int(x) for x in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
tag = sys.argv[5] if len(sys.argv) > 5 else ""
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; y = torch.empty(M, N, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def run():
    check(lib().phonlp_linear(P(x), K, P(w), K, None, P(y), N, M, N, K, 0, None, 0, None), "linear")
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
for _ in range(reps): run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{tag} M={M} N={N} K={K}: {dt*1e3:.3f} ms  {2.0*M*N*K/dt/1e12:.1f} TFLOP/s", flush=True)
