"""MST alone over sentence lengths: ms per call and how many sentences the round-based decode handed to the reference-order kernel.
usage: python tools/mst_len_probe.py [B L]..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phonlp_amd import HeadWeights, TaskHeads, ops, synthetic as S
dev = torch.device("cuda:0")
heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(), dev))
args = [int(v) for v in sys.argv[1:]] or [256, 128, 128, 160, 128, 256, 64, 384, 64, 512]
for B, L in zip(args[::2], args[1::2]):
    enc_dep, _ = S.make_encoder_outputs(B, L)
    lens = torch.tensor([L] * B, dtype=torch.int32, device=dev)
    arc = heads.parser(enc_dep.to(dev), lens)["arc_logp"]
    for _ in range(2):
        h, st, nfb = ops.mst_one_root_counted(arc, lens)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        ops.mst_one_root(arc, lens)
    b.record()
    torch.cuda.synchronize()
    why = ""
    if int(nfb) > 0:  # reasons (include/phonlp_heads.h): 1 = sums not provably exact, 2 / 3 = a tie in a selection
        why = " reasons " + str([int(v) >> 24 for v in ops.mst_one_root_counted.last_list[:int(nfb)].tolist()])
    print(f"B={B} L={L}: {a.elapsed_time(b) / 5:.3f} ms per call, fallbacks {int(nfb)}, status!=0 {int((st != 0).sum())}{why}")
