"""Long sentences (BASELINE config 5): per batch the step time, the MST's time alone and how many sentences the
round-based decode hands to the reference-order kernels (and why).  usage: python tools/long_probe.py [B L nbatches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("PHONLP_LIB"):  # A/B of two builds of the library on one box
    from phonlp_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ["PHONLP_LIB"])
from phonlp_amd import HeadWeights, TaskHeads, ops, synthetic as S
B, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 512)
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda:0")
heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(), dev))
ld = [L] * B; lt = [L - 1] * B
lens = torch.tensor(ld, dtype=torch.int32, device=dev)
for i in range(nb):
    ed, et = (t.to(dev) for t in S.make_encoder_outputs(B, L, 768, seed=4321 + B + L + 17 * i))
    for _ in range(2): out = heads.annotate_batch(ed, ld, et, lt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): out = heads.annotate_batch(ed, ld, et, lt)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    arc = out["arc_logp"]
    h, st, nfb = ops.mst_one_root_counted(arc, lens)
    torch.cuda.synchronize()
    lst = ops.mst_one_root_counted.last_list[:max(int(nfb), 0)].cpu().tolist()
    t0 = time.perf_counter()
    for _ in range(3): ops.mst_one_root(arc, lens)
    torch.cuda.synchronize(); mst = (time.perf_counter() - t0) / 3 * 1e3
    print(f"batch {i}: step {ms:.2f} ms  mst alone {mst:.2f} ms  fallbacks {int(nfb)} {[(v & 0xFFFFFF, v >> 24) for v in lst]}", flush=True)
