"""HIP-graph capture of one annotate_batch step at a fixed shape: replay time vs eager launches.
usage: python tools/graph_probe.py B L [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phonlp_amd import HeadWeights, TaskHeads, synthetic as S
B, L = int(sys.argv[1]), int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(), dev))
enc_dep, enc_tag = S.make_encoder_outputs(B, L)
ld = torch.tensor([L] * B, dtype=torch.int32, device=dev); lt = ld - 1
ed, et = enc_dep.to(dev), enc_tag.to(dev)
for _ in range(3): out = heads.annotate_batch(ed, ld, et, lt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): out = heads.annotate_batch(ed, ld, et, lt)
torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / reps
want = {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): heads.annotate_batch(ed, ld, et, lt)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    gout = heads.annotate_batch(ed, ld, et, lt)
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): g.replay()
torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / reps
ok = all(torch.equal(want[k], gout[k]) for k in want)
print(f"B={B} L={L}: eager {eager*1e3:.3f} ms  graph {graph*1e3:.3f} ms  outputs equal: {ok}")
