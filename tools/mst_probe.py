"""Times the MST kernel alone on the bench's own arc scores. usage: python tools/mst_probe.py [reps [B L]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phonlp_amd import HeadWeights, TaskHeads, ops, synthetic as S
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
B, L = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 128)
heads = TaskHeads(HeadWeights.from_state_dict(S.make_state_dict(), dev))
enc_dep, _ = S.make_encoder_outputs(B, L)
lens = torch.tensor([L] * B, dtype=torch.int32, device=dev)
arc = heads.parser(enc_dep.to(dev), lens, want_label_argmax=False)["arc_logp"]
ref = None
for fast in (True, False):
    ops.set_mst_fast_path(fast)
    h, st, nfb = ops.mst_one_root_counted(arc, lens)
    torch.cuda.synchronize()
    if fast and int(nfb) > 0:
        fl = ops.mst_one_root_counted.last_list[:int(nfb)].cpu().tolist()
        print("fallback sentences (sentence, reason):", [(v & 0xFFFFFF, v >> 24) for v in fl], flush=True)
        os.makedirs("gpurun_out", exist_ok=True)
        np.save("gpurun_out/mst_fallback_arc.npy", arc[[v & 0xFFFFFF for v in fl[:4]]].cpu().numpy())
    t0 = time.perf_counter()
    for _ in range(reps):
        h, st = ops.mst_one_root(arc, lens)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # timing only: parity of the decoder is the tests' job (tests/test_gpu_parity.py::test_mst_*), tools never touch oracle/
    same = "" if ref is None else f" same_as_rounds={bool((h == ref).all())}"
    ref = h
    print(f"mst B={B} L={L} {'rounds' if fast else 'reference-order'}: {dt*1e3:.3f} ms  failures={int(st.ne(0).sum())} "
          f"fallbacks={int(nfb)}{same}", flush=True)
ops.set_mst_fast_path(True)
