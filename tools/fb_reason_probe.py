"""Which sentences of the bench's batches leave the round-based MST for the reference-order kernel, and why (reason
codes of mst_fast_pick_kernel: 1 = exactness (G), 2 = a tie / round limit in the unconstrained decode, 3 = in a candidate
decode); then the same for "confident" scores (log_softmax(k * logp): a trained model's near-certain arcs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phonlp_amd import HeadWeights, TaskHeads, ops, synthetic as S
B, L, E = 256, 128, 768
dev = torch.device("cuda:0")
heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(), dev))
ld = [L] * B; lt = [L - 1] * B
lens = torch.tensor(ld, dtype=torch.int32, device=dev)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
def timed(arc):
    for _ in range(2): ops.mst_one_root(arc, lens)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ops.mst_one_root(arc, lens)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 5 * 1e3
for i in range(nb):
    ed, et = (t.to(dev) for t in S.make_encoder_outputs(B, L, E, seed=1234 + 17 * i))
    out = heads.annotate_batch(ed, ld, et, lt)
    torch.cuda.synchronize()
    arc = out["arc_logp"]
    h, st, nfb = ops.mst_one_root_counted(arc, lens)
    torch.cuda.synchronize()
    lst = ops.mst_one_root_counted.last_list[:int(nfb)].cpu().tolist()
    print(f"batch {i}: fallbacks {int(nfb)} {[(v & 0xFFFFFF, v >> 24) for v in lst]}  mst alone {timed(arc):.3f} ms", flush=True)
    if i == 0:
        for k in (4.0, 16.0, 64.0):
            a2 = torch.log_softmax(arc * k, 2).contiguous()
            h, st, nfb = ops.mst_one_root_counted(a2, lens)
            torch.cuda.synchronize()
            lst = ops.mst_one_root_counted.last_list[:int(nfb)].cpu().tolist()
            reasons = [v >> 24 for v in lst]
            print(f"  confident x{k}: max prob median {float(a2.exp().amax(2).median()):.6f}  fallbacks {int(nfb)} of {B} "
                  f"(G {reasons.count(1)}, tie {reasons.count(2)}, cand {reasons.count(3)})  mst alone {timed(a2):.3f} ms", flush=True)
