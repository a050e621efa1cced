"""Times the MST kernel alone on the bench's own arc scores. usage: python tools/mst_probe.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phonlp_amd import HeadWeights, TaskHeads, ops, synthetic as S
from oracle import c_oracle as C
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
B, L = 256, 128
heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(), dev))
enc_dep, _ = S.make_encoder_outputs(B, L)
lens = torch.tensor([L] * B, dtype=torch.int32, device=dev)
arc = heads.parser(enc_dep.to(dev), lens)["arc_logp"]
h, st = ops.mst_one_root(arc, lens)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    h, st = ops.mst_one_root(arc, lens)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
wh, ws = C.mst_batch(arc.cpu().numpy(), [L] * B)
ok = np.array_equal(h.cpu().numpy(), wh) and not st.any().item()
print(f"mst B={B} L={L}: {dt*1e3:.3f} ms  parity={'OK' if ok else 'MISMATCH'}", flush=True)
