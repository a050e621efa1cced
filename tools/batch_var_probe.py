"""Per-batch variation of the step: the bench's resident batches one by one (kernel means over 3 steps each), with the
MST's fallback count and the ROOT-children statistics of the unconstrained trees."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phonlp_amd import HeadWeights, TaskHeads, ops, synthetic as S
from phonlp_amd.profile import KernelTimer
B, L, E = 256, 128, 768
dev = torch.device("cuda:0")
heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(), dev))
ld = [L] * B; lt = [L - 1] * B
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batches = [tuple(t.to(dev) for t in S.make_encoder_outputs(B, L, E, seed=1234 + 17 * i)) for i in range(nb)]
for i, (ed, et) in enumerate(batches):
    for _ in range(2): heads.annotate_batch(ed, ld, et, lt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): heads.annotate_batch(ed, ld, et, lt)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    with KernelTimer(4096) as kt:
        for _ in range(3): heads.annotate_batch(ed, ld, et, lt)
        torch.cuda.synchronize()
    dep = heads.parser(ed, ld, want_label_argmax=False)
    lens = torch.tensor(ld, dtype=torch.int32, device=dev)
    ops.set_mst_fast_path(True)
    h, st, nfb = ops.mst_one_root_counted(dep["arc_logp"], lens)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): ops.mst_one_root(dep["arc_logp"], lens)
    torch.cuda.synchronize(); mst_alone = (time.perf_counter() - t0) / 5 * 1e3
    k = {a: round(v[1], 3) for a, v in sorted(kt.stats.items())}
    print(f"batch {i}: step {ms:.3f} ms  mst in step {k.get('mst')}  mst alone {mst_alone:.3f}  fallbacks {int(nfb)}  "
          f"stage_a {k.get('linear:label_stage_a')} arc_w12 {k.get('linear:arc_w12')}", flush=True)
# rotating
for _ in range(4): [heads.annotate_batch(ed, ld, et, lt) for ed, et in batches]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): [heads.annotate_batch(ed, ld, et, lt) for ed, et in batches]
torch.cuda.synchronize(); print("rotating: %.3f ms/step" % ((time.perf_counter() - t0) / (5 * nb) * 1e3))
