"""Host time to ENQUEUE one joined step against the time the GPU takes for it (is the loop launch-bound?): python tools/enq_probe.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from phonlp_amd import HeadWeights, TaskHeads, synthetic as S
B, L = 256, 128
dev = torch.device("cuda:0")
heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(), dev))
enc_dep, enc_tag = S.make_encoder_outputs(B, L)
ld = torch.tensor([L] * B, dtype=torch.int32, device=dev); lt = ld - 1
ed, et = enc_dep.to(dev), enc_tag.to(dev)
for _ in range(5): out = heads.annotate_batch(ed, ld, et, lt)
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for _ in range(n): out = heads.annotate_batch(ed, ld, et, lt)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/n:.3f} ms/step (host), total {1e3*(t2-t0)/n:.3f} ms/step")
