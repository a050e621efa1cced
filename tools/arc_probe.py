"""Times the fused arc kernel alone (first-layer image -> arc log-probs): python tools/arc_probe.py [lib.so] ...
One line per library: B=256 and B=1 at L=128, H=400."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
libs = sys.argv[1:] or [None]
import subprocess
if len(libs) > 1:  # one process per library (the library is loaded once per process)
    for l in libs:
        subprocess.run([sys.executable, __file__, l], check=False)
    sys.exit(0)
from phonlp_amd import _lib
if libs[0]:
    _lib.LIB_PATH = os.path.abspath(libs[0])
from phonlp_amd import HeadWeights, ops, synthetic as S
from phonlp_amd.profile import KernelTimer
dev = torch.device("cuda:0")
w = HeadWeights.from_state_dict(S.make_state_dict(), dev)
res = []
shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ.get("ARC_SHAPES", "256x128,1x128").split(",")]
for B, L in shapes:
    M = B * L
    feat = torch.randn(M, 792, device=dev)
    fp, fi = ops.split_rows(feat)
    lens = torch.full((B,), L, dtype=torch.int32, device=dev)
    xfp, xfi = ops.linear_split_image(fp, fi, w.arc_W12q_s[0], w.arc_W12q_s[1], w.arc_b12q, 792, True, lens, L, w.arc_b12q_pad,
                                      w.arc_bound[0], w.arc_bound[1], 3)
    run = lambda: ops.arc_scores_split(xfp, xfi, B, L, w.H, 3, True, True, w.arc_Wbq_s[0], w.arc_Wbq_s[1], None, None, None, 0.0)
    for _ in range(3): run()
    torch.cuda.synchronize()
    with KernelTimer(256) as kt:
        for _ in range(10): run()
        torch.cuda.synchronize()
    res.append(f"B={B} L={L}: {kt.stats['arc_fused'][1]*1e3:.1f} us (min {kt.stats['arc_fused'][2]*1e3:.1f})")
    if os.environ.get("ARC_STAMPS"):  # a -DARC_STAMPS build: cycle stamps of block 0
        dbg = torch.zeros(16, dtype=torch.int64, device=dev)
        os.environ["PHONLP_ARC_DBG"] = str(dbg.data_ptr())
        run(); torch.cuda.synchronize()
        t = dbg.cpu().tolist()
        names = ["A0", "cvt0", "B0", "terms0", "A1", "cvt1", "B1", "terms1", "A2", "cvt2", "B2", "terms2", "softmax"]
        res.append("stamps(cycles): " + " ".join(f"{n}={t[k + 1] - t[k]}" for k, n in enumerate(names)) + f" total={t[13] - t[0]}")
print(os.path.basename(libs[0] or "default"), " | ".join(res), flush=True)
