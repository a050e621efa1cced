"""parser_backward at the headline size, a few calls (run under rocprofv3 --kernel-trace --stats to see every kernel, torch's included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phonlp_amd import HeadWeights, TaskHeads, synthetic as S
B, L = 256, 128
dev = torch.device("cuda:0")
heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(), dev))
enc_dep, _ = S.make_encoder_outputs(B, L)
g = torch.Generator().manual_seed(1)
head = torch.randint(0, L - 1, (B, L), generator=g)
head = torch.where(head < torch.arange(L).view(1, L), head, head + 1).to(dev, torch.int32)
rel = torch.randint(0, heads.w.O, (B, L), generator=g).to(dev, torch.int32)
enc = enc_dep.to(dev); lens = [L] * B
for _ in range(2): heads.parser_backward(enc, lens, head, rel, B * L)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(n): heads.parser_backward(enc, lens, head, rel, B * L)
torch.cuda.synchronize()
print(f"parser_backward {(time.perf_counter() - t0) / n * 1e3:.2f} ms")
