"""How long the host takes to ENQUEUE one annotate_batch step vs how long the GPU takes to run it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phonlp_amd import HeadWeights, TaskHeads, ops, synthetic as S
dev = torch.device("cuda:0")
B, L = 256, 128
heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(), dev))
enc_dep, enc_tag = (t.to(dev) for t in S.make_encoder_outputs(B, L))
ld = torch.tensor([L] * B, dtype=torch.int32, device=dev)
lt = torch.tensor([L - 1] * B, dtype=torch.int32, device=dev)
for join in (True, False):
    for _ in range(3):
        heads.annotate_batch(enc_dep, ld, enc_tag, lt, join=join)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        heads.annotate_batch(enc_dep, ld, enc_tag, lt, join=join)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"join={join}: enqueue {1e3*(t1-t0)/n:.3f} ms/step, total {1e3*(t2-t0)/n:.3f} ms/step", flush=True)
