"""Times the whole head path at an arbitrary (B, L): usage python tools/config_probe.py B L [reps] [ragged]
(ragged: lengths uniform in [L/4, L], sorted descending, given as host lists; timed packed and padded)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phonlp_amd import HeadWeights, TaskHeads, synthetic as S
from phonlp_amd.profile import KernelTimer
B, L = int(sys.argv[1]), int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(), dev))
enc_dep, enc_tag = S.make_encoder_outputs(B, L)
ragged = len(sys.argv) > 4 and sys.argv[4] == "ragged"
lens1 = S.make_sentlens(B, L, ragged=ragged); lens = [l - 1 for l in lens1]
ed, et = enc_dep.to(dev), enc_tag.to(dev)
res = {}
for pack in ((90, 0) if ragged else (0,)):
    heads.cfg["pack_pct"] = pack
    for _ in range(3): out = heads.annotate_batch(ed, lens1, et, lens)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = heads.annotate_batch(ed, lens1, et, lens)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    with KernelTimer(4096) as kt:
        for _ in range(2): out = heads.annotate_batch(ed, lens1, et, lens)
        torch.cuda.synchronize()
    res[pack] = dt
    print(f"B={B} L={L} ragged={ragged} live={sum(lens1)/(B*L):.3f} packed={bool(pack)}: {dt*1e3:.2f} ms/step  {B/dt:.0f} sent/s  "
          f"mst_fail={int((out['mst_status']!=0).sum())}")
    print({k: round(v[1], 3) for k, v in sorted(kt.stats.items())})
if ragged:
    print(f"packed vs padded: {res[0]/res[90]:.2f}x")
# timing only: parity at these sizes is covered by tests/test_gpu_parity.py (tools never touch oracle/)
