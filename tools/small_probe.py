"""Times the first layers of the heads at small shapes and prints a checksum of their outputs' bits (two builds of the library
with different `small_max` in launch_linear_f16x3 must print the same checksums: the small kernel gives the 256 x 128
kernel's bits; tests/test_gpu_small_batch.py holds that property in the suite).
python tools/small_probe.py "B,L B,L ..." """
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phonlp_amd import HeadWeights, ops, synthetic as S
from phonlp_amd.profile import KernelTimer
dev = torch.device("cuda:0")
w = HeadWeights.from_state_dict(S.make_state_dict(), dev)
shapes = [tuple(int(v) for v in t.split(",")) for t in (sys.argv[1] if len(sys.argv) > 1 else "1,32 32,64 64,128").split()]
for B, L in shapes:
    M = B * L
    g = torch.Generator(device=dev); g.manual_seed(B * 1000 + L)
    feat = torch.relu(torch.randn(M, 792, device=dev, generator=g))
    fp, fi = ops.split_rows(feat)
    lens = torch.full((B,), L, dtype=torch.int32, device=dev); lens[0] = max(2, L - 3)
    runs = {
        "arc_w12": lambda: ops.linear_split_image(fp, fi, w.arc_W12q_s[0], w.arc_W12q_s[1], w.arc_b12q, 792, True, lens, L,
                                                  w.arc_b12q_pad, w.arc_bound[0], w.arc_bound[1], 3),
        "label_w12": lambda: ops.linear_split(fp, fi, w.lab_W12p_s[0], w.lab_W12p_s[1], w.lab_b12f, 792, True, lens, L, w.lab_b12, 2),
    }
    out = []
    for name, run in runs.items():
        for _ in range(3): y = run()
        torch.cuda.synchronize()
        with KernelTimer(256) as kt:
            for _ in range(10): y = run()
            torch.cuda.synchronize()
        st = [v for k, v in kt.stats.items() if k.startswith("linear")]
        ys = y if isinstance(y, (tuple, list)) else (y,)
        h = hashlib.md5(b"".join(t.contiguous().view(torch.uint8).cpu().numpy().tobytes() for t in ys)).hexdigest()[:10]
        out.append(f"{name}: {st[0][1]*1e3:.1f} us (min {st[0][2]*1e3:.1f}) {h}")
    print(f"B={B} L={L} M={M}", " | ".join(out), flush=True)
