"""Times the training-side loss kernels at the headline size (B=256, L=128) with hipEvents:
    python tools/loss_probe.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phonlp_amd import HeadWeights, TaskHeads, ops
from phonlp_amd import synthetic as S

B, L = 256, 128
dev = torch.device("cuda:0")
sd = S.make_state_dict()
enc_dep, enc_tag = S.make_encoder_outputs(B, L)
lens1 = S.make_sentlens(B, L, False)
heads = TaskHeads(HeadWeights.from_state_dict(sd, dev))
w = heads.w
dep = heads.parser(enc_dep.to(dev), lens1, want_raw=True, want_label_argmax=False)
XF, T = ops.label_project(dep["hidden_out"], w.lab_W12, w.lab_b12, w.lab_Wbt, w.H, w.O)
g = torch.Generator().manual_seed(1)
head = torch.randint(0, L - 1, (B, L), generator=g)
head = torch.where(head < torch.arange(L).view(1, L), head, head + 1).to(dev, torch.int32)
rel = torch.randint(0, w.O, (B, L), generator=g).to(dev, torch.int32)
tag = heads.tagger(enc_tag.to(dev), [l - 1 for l in lens1])
gold = torch.randint(0, 24, (B, L - 1), generator=g).to(dev, torch.int32)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


t_dep = timed(lambda: ops.dep_loss(dep["arc_logp"], dep["raw"], T, XF, head, rel, dep["sentlens"], w.H, w.O, True, True, float(B * L)))
t_pos = timed(lambda: ops.pos_loss(tag["upos_logits"], gold, tag["sentlens"], 1))
nbytes = T.numel() * 4 + dep["arc_logp"].numel() * 4
t_dlg = timed(lambda: ops.dep_loss_grad(dep["arc_logp"], dep["raw"], T, XF, head, rel, dep["sentlens"], w.H, w.O, True, True, float(B * L)))
print(f"dep_loss_grad {t_dlg:.3f} ms ({(nbytes + 4.0 * B * L * L) / t_dlg / 1e6:.0f} GB/s over T + arc_logp + d_arc)")
print(f"dep_loss {t_dep:.3f} ms ({nbytes / t_dep / 1e6:.0f} GB/s over T + arc_logp = {nbytes / 1e9:.2f} GB), pos_loss {t_pos:.3f} ms")
from phonlp_amd.profile import KernelTimer
ner_gold = torch.randint(0, w.trans.shape[0], (B, L - 1), generator=g).to(dev, torch.int32)
with KernelTimer() as kt:
    for _ in range(10):
        ops.dep_loss(dep["arc_logp"], dep["raw"], T, XF, head, rel, dep["sentlens"], w.H, w.O, True, True, float(B * L))
        ops.pos_loss(tag["upos_logits"], gold, tag["sentlens"], 1)
        ops.crf_log_partition(tag["ner_logits"], w.trans, None, tag["sentlens"], ner_gold)
        ops.crf_nll_grad(tag["ner_logits"], w.trans, None, tag["sentlens"], ner_gold, 1.0 / B)
    torch.cuda.synchronize()
for k, (n, mean, mn) in sorted(kt.stats.items()):
    print(f"  {k}: {n} launches, mean {mean * 1e3:.1f} us, min {mn * 1e3:.1f} us")
# the backward through the scorers and dep_hid (TaskHeads.parser_backward) and its per-kernel split
enc_h = enc_dep.to(dev)
lens_l = [L] * B
t_bw = timed(lambda: heads.parser_backward(enc_h, lens_l, head, rel, B * L), n=5)
t_fw = timed(lambda: heads.parser(enc_h, lens_l, want_raw=True, want_label_argmax=False), n=5)
print(f"parser_backward {t_bw:.2f} ms (of which its own forward {t_fw:.2f} ms) at B={B} L={L}")
with KernelTimer(max_samples=4096) as kt:
    heads.parser_backward(enc_h, lens_l, head, rel, B * L)
    torch.cuda.synchronize()
tot = {}
for k, (n, mean, mn) in sorted(kt.stats.items()):
    if k.startswith("bw"):
        tot[k] = n * mean
print("  backward kernels (ms per call):", {k: round(v, 3) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})
