"""Experiment: B=256 as one batch vs split into n sub-batches pushed through annotate_batch in its streaming form
(the MST of one sub-batch beside the GEMMs of the next).  usage: python tools/halfbatch_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phonlp_amd import HeadWeights, TaskHeads, synthetic as S
dev = torch.device("cuda:0")
B, L = 256, 128
heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(), dev))
ed, et = S.make_encoder_outputs(B, L)
ed, et = ed.to(dev), et.to(dev)
for nsplit in (1, 2, 4):
    b = B // nsplit
    parts = [(ed[i * b:(i + 1) * b].contiguous(), torch.full((b,), L, dtype=torch.int32, device=dev),
              et[i * b:(i + 1) * b].contiguous(), torch.full((b,), L - 1, dtype=torch.int32, device=dev)) for i in range(nsplit)]
    def step():
        outs = [heads.annotate_batch(*p, join=(nsplit == 1)) for p in parts]
        for o in outs:
            torch.cuda.current_stream().wait_stream(o["stream"])
            torch.cuda.current_stream().wait_stream(o["tag_stream"])
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    print(f"B=256 in {nsplit} sub-batch(es): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per 256 sentences", flush=True)
