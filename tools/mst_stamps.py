"""Diagnostic: per-phase cycle shares of the MST kernel (needs tools/libphonlp_heads_stamps.so, built with -DMST_STAMPS)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from phonlp_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "libphonlp_heads_stamps.so")
from phonlp_amd import HeadWeights, TaskHeads, ops, synthetic as S
dev = torch.device("cuda:0")
B, L = int(os.environ.get("STAMP_B", 256)), int(os.environ.get("STAMP_L", 128))
heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(), dev))
enc_dep, _ = S.make_encoder_outputs(B, L)
lens = torch.tensor([L] * B, dtype=torch.int32, device=dev)
arc = heads.parser(enc_dep.to(dev), lens)["arc_logp"]
fast = os.environ.get("STAMP_FAST", "1") == "1"
ops.set_mst_fast_path(fast)
dbg = torch.zeros(B * 5 if fast else B, 16, dtype=torch.int64, device=dev)
os.environ["PHONLP_MST_DBG"] = str(dbg.data_ptr())
ops.mst_one_root(arc, lens); torch.cuda.synchronize()
dbg.zero_()
ops.mst_one_root(arc, lens); torch.cuda.synchronize()
d = dbg.cpu().double()
if fast:  # round-based kernel: phase 1 = rows [0, B), phase 2 = rows [B, 5B) (one per candidate slot)
    names = ["load+exact", "argmax", "detect", "label+record", "relabel", "merge+in-edge", "new-state", "expand+tail"]
    if L > 128:  # csrc/mst_round.hip
        names = ["scan+exact", "-", "detect", "label+record", "relabel+list", "rescan", "in-edge", "expand"]
    for ph, (lo, hi) in (("phase 1", (0, B)), ("phase 2", (B, 5 * B))):
        x = d[lo:hi]
        x = x[x[:, 11] > 0]
        if not len(x):
            continue
        tot = x[:, :8].sum(1)
        print(f"{ph}: {len(x)} workgroups decoded; decodes/wg mean {x[:,11].mean():.2f} max {int(x[:,11].max())}; rounds/decode mean "
              f"{x[:,8].sum()/x[:,11].sum():.1f}; full-detect rounds {100*x[:,9].sum()/max(x[:,8].sum(),1):.0f}%; new components per round "
              f"{x[:,10].sum()/max(x[:,8].sum(),1):.2f}; cycles/wg mean {tot.mean():.0f} max {tot.max():.0f}")
        for k, n in enumerate(names):
            per = "per round" if 2 <= k <= 6 else "per decode"
            den = x[:, 8].sum() if 2 <= k <= 6 else x[:, 11].sum()
            if k == 0 and L > 128:
                print(f"  rows rescanned per round {x[:,9].sum()/max(x[:,8].sum(),1):.1f}")
            print(f"  {n:14s} share {100*x[:,k].sum()/tot.sum():5.1f}%   {per} {x[:,k].sum()/max(den,1):8.0f} cyc")
        for k in (12, 13, 14, 15):  # extra stamps of a diagnostic build, if any (cycles per round; not part of `tot`)
            if x[:, k].sum() > 0:
                print(f"  extra stamp {k}: per round {x[:,k].sum()/max(x[:,8].sum(),1):8.0f} cyc")
    sys.exit(0)
names = ["init", "argmax", "cycles", "lists", "edge+sum", "enter/leave+rec", "contract", "expand", "oneroot", "-", "-", "-", "rescan"]
tot = d[:, :9].sum(1) + d[:, 12]
print("rows rescanned per level: %.2f" % (d[:, 13].sum() / max(d[:, 10].sum(), 1)))
print("runs/sentence mean %.2f max %d; levels/sentence mean %.1f max %d" % (d[:, 11].mean(), d[:, 11].max(), d[:, 10].mean(), d[:, 10].max()))
print("cycles/sentence mean %.0f max %.0f" % (tot.mean(), tot.max()))
for k, n in enumerate(names):
    if n == "-": continue
    print(f"  {n:16s} share {100*d[:,k].sum()/tot.sum():5.1f}%   per level {d[:,k].sum()/max(d[:,10].sum(),1):8.0f} cyc")
