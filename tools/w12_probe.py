"""Times the scorers' first layers alone at the headline shape (M = 32768 words, K = 792): the arc layer with the image
epilogue (N = 2496), the label layer (N = 816) and upos_hid (N = 400).  python tools/w12_probe.py [lib.so]
Environment: PHONLP_X3_TR (temporary A/B of the transposed persistent form), PHONLP_X3_ABLATION with the ablation build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phonlp_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from phonlp_amd import HeadWeights, ops, synthetic as S
from phonlp_amd.profile import KernelTimer
dev = torch.device("cuda:0")
w = HeadWeights.from_state_dict(S.make_state_dict(), dev)
B, L = int(os.environ.get("W12_B", 256)), int(os.environ.get("W12_L", 128))
M = B * L
feat = torch.relu(torch.randn(M, 792, device=dev))
fp, fi = ops.split_rows(feat)
lens = torch.full((B,), L, dtype=torch.int32, device=dev)
runs = {
    "arc_w12(image)": lambda: ops.linear_split_image(fp, fi, w.arc_W12q_s[0], w.arc_W12q_s[1], w.arc_b12q, 792, True, lens, L,
                                                     w.arc_b12q_pad, w.arc_bound[0], w.arc_bound[1], 3),
    "label_w12(as kind 0)": lambda: ops.linear_split(fp, fi, w.lab_W12p_s[0], w.lab_W12p_s[1], w.lab_b12f, 792, True, lens, L, w.lab_b12, 0),
    "label_w12(kind 2)": lambda: ops.linear_split(fp, fi, w.lab_W12p_s[0], w.lab_W12p_s[1], w.lab_b12f, 792, True, lens, L, w.lab_b12, 2),
}
out = []
for name, run in runs.items():
    for _ in range(3): run()
    torch.cuda.synchronize()
    with KernelTimer(256) as kt:
        for _ in range(10): run()
        torch.cuda.synchronize()
    st = [v for k, v in kt.stats.items() if k.startswith("linear")]
    out.append(f"{name}: {st[0][1]*1e3:.1f} us (min {st[0][2]*1e3:.1f})")
print(f"TR={os.environ.get('PHONLP_X3_TR','-')} ABL={os.environ.get('PHONLP_X3_ABLATION','-')}", " | ".join(out), flush=True)
