"""Times the fused label stage A (labels-at-heads epilogue) alone on the chip at the headline shape.
usage: python tools/fused_probe.py [reps]   (PHONLP_X3_ABLATION=1/2: timing-only variants of the epilogue)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("PHONLP_X3_ABLATION"):  # the ablation branches only exist in the -DPHONLP_ABLATION build
    from phonlp_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libphonlp_heads_ablation.so")
from phonlp_amd import HeadWeights, TaskHeads, ops, synthetic as S
from phonlp_amd.profile import KernelTimer
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
B, L, H, O = 256, 128, 400, 33
w = HeadWeights.from_state_dict(S.make_state_dict(), dev)
KP = ops.kp(H)
x1 = torch.relu(torch.randn(B * L, KP, device=dev)); x1[:, H] = 1.0; x1[:, H + 1:] = 0.0
x1p, x1i = ops.split_rows(x1)
xf = torch.relu(torch.randn(B * L, 2 * KP, device=dev))
i = torch.arange(L, device=dev).unsqueeze(0).expand(B, L)
hd = ((i + torch.randint(-3, 4, (B, L), device=dev)).clamp(0, L - 1)).to(torch.int32).contiguous()  # nearby heads, like decoded trees
ld = torch.full((B,), L, dtype=torch.int32, device=dev)
run = lambda: ops.label_at_heads_fused(x1p, x1i, w.lab_Wbtc_s[0], w.lab_Wbtc_s[1], xf, hd, ld, H, O)
for _ in range(3): run()
torch.cuda.synchronize()
with KernelTimer(max_samples=256) as kt:
    for _ in range(reps): run()
    torch.cuda.synchronize()
print({k: (v[0], round(v[1], 4), round(v[2], 4)) for k, v in kt.stats.items()}, "abl", os.environ.get("PHONLP_X3_ABLATION"))
