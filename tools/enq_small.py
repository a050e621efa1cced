"""Host time to enqueue one step at a small shape, through the torch custom ops: python tools/enq_small.py B L"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phonlp_amd import HeadWeights, TaskHeads, synthetic as S
B, L = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(), dev))
enc_dep, enc_tag = S.make_encoder_outputs(B, L)
ld = [L] * B; lt = [L - 1] * B
ed, et = enc_dep.to(dev), enc_tag.to(dev)
for _ in range(5): out = heads.annotate_batch(ed, ld, et, lt)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n): out = heads.annotate_batch(ed, ld, et, lt)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B} L={L}: enqueue {1e3*(t1-t0)/n:.3f} ms/step (host), total {1e3*(t2-t0)/n:.3f} ms/step", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50): out = heads.annotate_batch(ed, ld, et, lt)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
