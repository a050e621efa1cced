"""Times the three labels-at-heads style row streams alone on the chip (B=256, L=128, H=400, O=33), optionally from a
variant library:  PHONLP_PROBE_LIB=tools/probe/libphonlp_heads_u2.so python tools/at_heads_probe.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from phonlp_amd import _lib
if os.environ.get("PHONLP_PROBE_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["PHONLP_PROBE_LIB"])
from phonlp_amd import ops
from phonlp_amd.profile import KernelTimer

B, L, H, O = 256, 128, 400, 33
KP = ops.kp(H)
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
T = torch.randn(B * L, O * KP, generator=g).to(dev)
XF = torch.randn(B * L, 2 * KP, generator=g).to(dev)
heads = torch.randint(0, L, (B, L), generator=g).to(dev, torch.int32)
rel = torch.randint(0, O, (B, L), generator=g).to(dev, torch.int32)
lens = torch.full((B,), L, dtype=torch.int32, device=dev)
arc = torch.randn(B, L, L, generator=g).to(dev)
raw = torch.randn(3, B, L, L, generator=g).to(dev)
with KernelTimer() as kt:
    for _ in range(12):
        ops.label_at_heads(T, XF, heads, lens, H, O)
        ops.label_at_heads_rows4(T.view(-1), XF, heads, lens, H, O)
        ops.dep_loss(arc, raw, T, XF, heads, rel, lens, H, O, True, True, float(B * L))
    torch.cuda.synchronize()
gb = T.numel() * 4 / 1e9
print(os.environ.get("PHONLP_PROBE_LIB", "shipped library"))
for k, (n, mean, mn) in sorted(kt.stats.items()):
    if "sum" in k:
        continue
    print(f"  {k}: mean {mean * 1e3:.1f} us, min {mn * 1e3:.1f} us -> {gb / mn:.2f} TB/s over T at the minimum")
