"""One small-shape step under rocprofv3 --kernel-trace, eager or as a graph replay: python3 tools/small_trace.py B L eager|graph [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phonlp_amd import HeadWeights, TaskHeads, synthetic as S
B, L, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]; reps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
dev = torch.device("cuda:0")
heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(), dev))
enc_dep, enc_tag = S.make_encoder_outputs(B, L)
ld = torch.tensor([L] * B, dtype=torch.int32, device=dev); lt = ld - 1
ed, et = enc_dep.to(dev), enc_tag.to(dev)
for _ in range(3): heads.annotate_batch(ed, ld, et, lt)
torch.cuda.synchronize()
if mode == "graph":
    step = heads.capture(B, L)
    step.enc_dep.copy_(ed); step.enc_tag.copy_(et)
    run = step.replay
else:
    run = lambda: heads.annotate_batch(ed, ld, et, lt)
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): run()
torch.cuda.synchronize()
print(f"B={B} L={L} {mode}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")
