"""How often the round-based MST hands a sentence to the reference-order kernel on fresh synthetic batches at the headline size,
and what that costs the batch: python tools/fb_rate_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phonlp_amd import HeadWeights, TaskHeads, ops, synthetic as S
dev = torch.device("cuda:0")
heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(), dev))
B, L, N = 256, 128, 60
lens = torch.tensor([L] * B, dtype=torch.int32, device=dev)
tot = 0; reasons = []
g = torch.Generator(device=dev).manual_seed(7)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
times = []
for k in range(N):
    enc = torch.randn(B, L, 768, device=dev, generator=g)
    arc = heads.parser(enc, lens)["arc_logp"]
    ops.mst_one_root(arc, lens); torch.cuda.synchronize()
    ev[0].record(); h, st, nfb = ops.mst_one_root_counted(arc, lens); ev[1].record(); torch.cuda.synchronize()
    n = int(nfb); tot += n
    times.append((n, ev[0].elapsed_time(ev[1])))
    if n: reasons += [int(v) >> 24 for v in ops.mst_one_root_counted.last_list[:n].tolist()]
t0 = [t for n, t in times if n == 0]; t1 = [t for n, t in times if n > 0]
print(f"{tot} fallbacks in {N * B} sentences ({N} batches, {len(t1)} with one or more); reasons {reasons}")
print(f"MST ms per batch: without fallback {sum(t0)/max(len(t0),1):.3f}, with {sum(t1)/max(len(t1),1):.3f}")
