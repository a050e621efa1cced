#!/usr/bin/env python3
"""bench.py -- sentences/sec through the POS + NER + dependency heads on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole head path over one synthetic batch (BASELINE.json config 3: B=256
sentences, L=128 words incl. ROOT; encoder outputs already resident in HBM): POS head (both task passes),
dep_hid, four deep-biaffine scorers with the fused arc epilogue + log-softmax and label arg-max, NER
emissions + Viterbi, Chu-Liu-Edmonds single-root MST, relation lookup.  Output per sentence: POS ids, NER
tags, heads, relation ids, left on the device.  Sentences are independent, so ranks each run their own
batches (weak scaling, no data-path collective); the only collective is one all-reduce of the eval-count
vector (RCCL) inside the timed region.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (bilinear stage A of the label
scorer, an fp32 MFMA GEMM) timed live with HIP events on its launch stream; `cpu_baseline` is the CPU
oracle (kind "port": torch-CPU GEMMs + per-sentence numpy decoders, the reference's structure) timed on
this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, L, E, TE, H, U, O, T = 256, 128, 768, 100, 400, 24, 33, 9
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak (engine 0: v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 matrix peak; engine 1 issues 6 bf16 MFMA flops per fp32-equivalent flop
PEAK_HBM_GBS = 8000.0


def algorithmic_flops():
    """Per sentence of parser length L (2*m*n*k).  `total`: SURVEY.md section 8(d)'s formulas for the reference's path
    (incl. the dense L x L x O label stage B and dep_hid as a layer of its own).  `executed`: what the timed step
    actually multiplies -- dep_hid and the soft POS embedding are folded into the scorers' first layers (K = 768 + U),
    the label scorer is evaluated at the decoded heads only (L x O dot products instead of L x L x O)."""
    D, H1 = E + TE, H + 1
    per_scorer = lambda o: 4 * L * D * H + 2 * L * H1 * H1 * o + 2 * L * L * H1 * o
    biaffine = 3 * per_scorer(1) + per_scorer(O)
    dep_hid = 2 * L * D * D
    pos = 2 * L * (E * H + H * U + U * TE) + 2 * (L - 1) * (E * H + H * U + U * TE)
    ner = 2 * (L - 1) * D * T
    Kf = E + U
    executed = (4 * 4 * L * Kf * H                      # W1/W2 of the four scorers on [enc | tag probabilities]
                + 3 * (2 * L * H1 * H1 + 2 * L * L * H1)  # arc scorers: stage A + stage B
                + 2 * L * H1 * H1 * O + 2 * L * H1 * O    # label scorer: stage A + the contraction at the heads
                + 2 * L * (E * H + H * U) + 2 * (L - 1) * (E * (H + T) + H * U + U * TE + TE * T))  # POS / tagger heads
    return {"total": biaffine + dep_hid + pos + ner, "label_stage_a": 2 * L * H1 * H1 * O, "executed": executed}


def cpu_baseline(sd, enc_dep, enc_tag, lens_dep, lens_tag, n_sample=32):
    """The oracle on this host (kind "port"): GEMM heads through torch CPU (all intra-op threads), decoders per sentence
    in numpy exactly as the reference drives them (single thread).  One warm-up, then the minimum of three passes over a
    bounded sample."""
    from oracle import heads_oracle as Or
    ed, et = enc_dep[:n_sample], enc_tag[:n_sample]
    ld, lt = lens_dep[:n_sample], lens_tag[:n_sample]
    n_mst = min(n_sample, 12)
    trans = sd["crit_ner._transitions"].numpy()
    best = None
    with torch.no_grad():
        Or.dep_heads(ed[:4], ld[:4], sd, keep_label_scores=False)  # warm-up
        for _ in range(3):
            t0 = time.perf_counter()
            dep = Or.dep_heads(ed, ld, sd, keep_label_scores=False)
            tag = Or.tagger_heads(et, lt, sd)
            t_gemm = (time.perf_counter() - t0) / n_sample
            arc, logits = dep["arc_logp"].numpy(), tag["ner_logits"].numpy()
            t0 = time.perf_counter()
            for i in range(n_mst):
                Or.chuliu_edmonds_one_root(arc[i][:ld[i], :ld[i]])
            t_mst = (time.perf_counter() - t0) / n_mst
            t0 = time.perf_counter()
            for i in range(n_sample):
                Or.viterbi_decode(logits[i, :lt[i]], trans)
            t_vit = (time.perf_counter() - t0) / n_sample
            if best is None or t_gemm + t_mst + t_vit < sum(best):
                best = (t_gemm, t_mst, t_vit)
    t_gemm, t_mst, t_vit = best
    return {
        "value": 1.0 / sum(best), "unit": "sentences/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"min of 3 passes after a warm-up over {n_sample} of {B} sentences (L={L}) for the torch-CPU heads and "
                  f"Viterbi, {n_mst} for the numpy MST; per-sentence seconds: heads {t_gemm:.4f}, mst {t_mst:.4f}, "
                  f"viterbi {t_vit:.4f}; decoders single-threaded as in the reference",
        "note": "a port, not the reference: its iterative MST needs ~9 ms per 128-word sentence where the reference's own "
                "recursive chuliu_edmonds.py measured 75-120 ms (BASELINE.md section 2), so the reference itself is "
                "several times slower than this baseline; the ratio says nothing about kernel quality, roofline.frac does",
    }


def other_configs(heads, dev, reps):
    """The other BASELINE configs and the ragged / corpus forms of the headline, measured after the timed region with
    the same TaskHeads (untimed extras; rank 0 of a one-rank run): per config ms per joined step, sentences/s, MST
    status failures and the per-kernel means of two more steps with every launch bracketed by HIP events."""
    from phonlp_amd import synthetic as S
    from phonlp_amd.corpus import SyntheticCorpus, walk
    from phonlp_amd.profile import KernelTimer
    res = {}

    def run(name, Bc, Lc, ragged=False, pack_pct=None, note=None, graph=False):
        enc_dep_h, enc_tag_h = S.make_encoder_outputs(Bc, Lc, E, seed=4321 + Bc + Lc)
        lens1 = S.make_sentlens(Bc, Lc, ragged=ragged)  # host lists, as annotate() passes them (ragged batches pack)
        lens = [l - 1 for l in lens1]
        ed, et = enc_dep_h.to(dev), enc_tag_h.to(dev)
        keep = heads.cfg["pack_pct"]
        if pack_pct is not None:
            heads.cfg["pack_pct"] = pack_pct
        try:
            for _ in range(3):
                out = heads.annotate_batch(ed, lens1, et, lens)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                out = heads.annotate_batch(ed, lens1, et, lens)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            with KernelTimer(4096) as kt:
                for _ in range(2):
                    out = heads.annotate_batch(ed, lens1, et, lens)
                torch.cuda.synchronize()
        finally:
            heads.cfg["pack_pct"] = keep
        res[name] = {"B": Bc, "L": Lc, "live_fraction": round(sum(lens1) / (Bc * Lc), 3), "ms_per_step": round(ms, 4),
                     "value": round(Bc / (ms * 1e-3), 1), "unit": "sentences/s",
                     "mst_failures": int((out["mst_status"] != 0).sum()),
                     "kernels_ms": {k: round(v[1], 4) for k, v in sorted(kt.stats.items())}}
        if graph:  # the same step as one hipGraphLaunch (TaskHeads.capture): what a fixed-shape serving loop pays per batch
            step = heads.capture(Bc, Lc)
            step.enc_dep.copy_(ed)
            step.enc_tag.copy_(et)
            for _ in range(3):
                gout = step.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4 * reps):
                gout = step.replay()
            torch.cuda.synchronize()
            res[name]["graph_replay_ms"] = round((time.perf_counter() - t0) / (4 * reps) * 1e3, 4)
            res[name]["graph_equals_eager"] = bool(all(torch.equal(gout[k], out[k]) for k in ("pos", "ner", "heads", "rels")))
            del step
        if note:
            res[name]["note"] = note

    run("config2_B32_L64", 32, 64, note="BASELINE config 2 (launch-latency-bound)", graph=True)
    run("config5_B64_L512", 64, 512, note="BASELINE config 5 (long sentences: round-based MST on the matrix in global "
                                          "memory, unfused label path)")
    run("ragged_B256_L128_packed", B, L, ragged=True, pack_pct=90, note="lengths uniform in [32,128], live words packed")
    run("ragged_B256_L128_padded", B, L, ragged=True, pack_pct=0, note="the same batch at the padded shape")
    run("latency_B1_L32", 1, 32, note="the reference's default batch_size=1", graph=True)
    # a corpus walk over 64 DISTINCT batches drawn on the device (streaming form, counts tallied): BASELINE config 4 on one GPU
    corpus = SyntheticCorpus(64 * B, B, L, E, U, O, T, seed=991)
    walk(heads, corpus, 0, 1, dev, last=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cnt, cnt_ent, nsent = walk(heads, corpus, 0, 1, dev)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / corpus.n_batches * 1e3
    c = cnt.cpu().tolist()
    res["corpus_walk_64_batches"] = {"B": B, "L": L, "ms_per_step": round(ms, 4), "value": round(B / (ms * 1e-3), 1),
                                     "unit": "sentences/s", "sentences": c[0], "mst_failures": c[6],
                                     "note": "64 distinct batches generated on the device inside the timed loop "
                                             "(two randn of B x L x 768 per batch), streaming form, eval counts tallied"}
    return res


def launch_ranks(n, argv):
    """`bench.py --gpus N` with N > 1 and no external launcher: start N fresh rank processes of this script (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, rendezvous on 127.0.0.1), relay rank 0's JSON line and
    return non-zero if any rank does.  The parent makes no GPU call, before or after: the children are fresh processes
    (subprocess), never a re-exec of a process that has touched the GPU.  The reference has no counterpart (no
    parallelism); the north star asks for the number at 1/2/4/8 GPUs."""
    import socket
    import subprocess
    import threading
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))

    def relay():  # rank 0 prints the result line; the other ranks' stdout goes to stderr
        for line in procs[0].stdout:
            sys.stdout.write(line.decode(errors="replace"))
            sys.stdout.flush()

    th = threading.Thread(target=relay, daemon=True)
    th.start()
    rc = 0
    live = set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0:
                    rc = code if code > 0 else 128 - code
                    print(f"bench.py: rank {r} exited with {code}", file=sys.stderr, flush=True)
        time.sleep(0.05)
    for r in live:  # a rank failed: the others would wait in a collective forever -- end exactly the PIDs started here
        procs[r].terminate()
    for r in live:
        try:
            procs[r].wait(timeout=20)
        except subprocess.TimeoutExpired:
            procs[r].kill()
    th.join(timeout=5)
    return rc


def dry_run(args, rank, world):
    """PHONLP_BENCH_DRYRUN=1 (tests/test_bench_launch.py, CPU): the multi-rank SKELETON of this script -- rank start-up,
    rendezvous, batch sharding, the count all-reduce, MAX of the elapsed times, rank 0's line -- over gloo with NO head
    work at all: every rank only tallies the sentences of the batches it would have walked.  Never a reported number."""
    import torch.distributed as dist
    from phonlp_amd.sharding import shard_batches
    if os.environ.get("PHONLP_BENCH_DRYRUN_FAIL_RANK") == str(rank):  # the launcher's failure path under test
        sys.exit(3)
    if world > 1:
        dist.init_process_group(backend="gloo")
    counts = torch.zeros(11, dtype=torch.int64)
    t0 = time.perf_counter()
    if args.corpus > 0:
        n_batches = (args.corpus + B - 1) // B
        for i in shard_batches(n_batches, rank, world):
            counts[0] += min(B, args.corpus - i * B)
        steps, expect = n_batches, args.corpus
    else:
        counts[0] = B * args.steps
        steps, expect = args.steps, B * args.steps * world
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(counts)
        dist.all_reduce(t, dist.ReduceOp.MAX)
    if rank == 0:
        assert int(counts[0]) == expect, f"eval counts inconsistent: {counts.tolist()}"
        print(json.dumps({"metric": "dry run (no head work)", "value": None, "unit": "sentences/s", "n_gpus": world,
                          "steps": steps, "warmup": args.warmup, "data": "DRY RUN: launcher / rendezvous / reduction only",
                          "eval_counts": {"sentences": int(counts[0])}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--corpus", type=int, default=0, metavar="N",
                    help="BASELINE config 4 instead of the headline line: walk a corpus of N synthetic sentences "
                         "(L=128, batches of 256, distinct batches drawn on the device) sharded by batch over the ranks; "
                         "--steps/--warmup are ignored")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:  # no external launcher: this process only starts the ranks, it never touches a GPU
            sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")

    if os.environ.get("PHONLP_BENCH_DEBUG"):  # where is every rank after N seconds? (hang diagnosis)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PHONLP_BENCH_DEBUG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("PHONLP_BENCH_DRYRUN", "0") == "1":
        return dry_run(args, rank, world)
    # PHONLP_BENCH_REHEARSAL=1: rehearse the multi-rank code path on a one-GPU box (every rank on cuda:0, gloo instead
    # of RCCL, which refuses two ranks on one device); never used for reported numbers
    rehearsal = os.environ.get("PHONLP_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    from phonlp_amd import HeadWeights, TaskHeads
    from phonlp_amd import synthetic as S
    from phonlp_amd.profile import KernelTimer
    from phonlp_amd import ops, _lib

    sd = S.make_state_dict(E, TE, H, U, O, T, seed=1234)
    heads = TaskHeads(HeadWeights.from_state_dict(sd, dev))
    # every rank draws its own shard of the synthetic corpus (seed offset by rank): weak scaling.  N_RESIDENT distinct
    # batches are resident and the steps rotate over them: 4 x 200 MB of encoder rows do not fit the 256 MB Infinity
    # Cache, so a step reads its inputs from HBM as a corpus walk would (PHONLP_BENCH_BATCHES=1: one batch replayed, the
    # round-2 form, reported beside the headline as "single_batch_replay")
    N_RESIDENT = max(1, int(os.environ.get("PHONLP_BENCH_BATCHES", "4")))
    batches_h = [S.make_encoder_outputs(B, L, E, seed=1234 + 1000 * rank + 17 * i) for i in range(N_RESIDENT)]
    enc_dep_h, enc_tag_h = batches_h[0]
    lens_dep, lens_tag = [L] * B, [L - 1] * B
    batches = [(d.to(dev), t.to(dev)) for d, t in batches_h]
    enc_dep, enc_tag = batches[0]
    ld = torch.tensor(lens_dep, dtype=torch.int32, device=dev)
    lt = torch.tensor(lens_tag, dtype=torch.int32, device=dev)
    g = torch.Generator().manual_seed(99 + rank)  # synthetic gold annotations for the eval counts
    gold_pos = torch.randint(0, U, (B, L - 1), generator=g, dtype=torch.int32).to(dev)
    gold_ner = torch.randint(0, T, (B, L - 1), generator=g, dtype=torch.int32).to(dev)
    gold_head = torch.randint(0, L, (B, L), generator=g, dtype=torch.int32).to(dev)
    gold_rel = torch.randint(0, O, (B, L), generator=g, dtype=torch.int32).to(dev)
    counts = torch.zeros(8, dtype=torch.int64, device=dev)  # eval counts
    counts_ent = torch.zeros(3, dtype=torch.int64, device=dev)  # NER entities: correct, predicted, gold
    from phonlp_amd.sharding import prf, tag_tables
    ner_vocab = ["<PAD>", "<UNK>", "<EMPTY>", "<ROOT>", "O", "B-PER", "I-PER", "B-LOC", "I-LOC"][:T]
    tag_kind, tag_type = (t.to(dev) for t in tag_tables(ner_vocab))

    def all_reduce(t, op=None):
        """RCCL all-reduce of a device tensor (in place); the rehearsal mode's gloo group reduces a host copy."""
        kw = {} if op is None else {"op": op}
        if rehearsal:
            h = t.cpu()
            dist.all_reduce(h, **kw)
            t.copy_(h)
        else:
            dist.all_reduce(t, **kw)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.corpus > 0:
        # ---- BASELINE config 4: the whole corpus once, batches rank, rank + world, ... on this rank (streaming form), the
        # counts all-reduced at the end; the timed region holds the walk, the on-device batch generation and the collective
        from phonlp_amd.corpus import SyntheticCorpus, walk
        corpus = SyntheticCorpus(args.corpus, B, L, E, U, O, T, seed=777)
        walk(heads, corpus, rank, world, dev, last=2 * world)  # warm-up: two batches per rank
        fence()
        t0 = time.perf_counter()
        cnt, cnt_ent, nsent = walk(heads, corpus, rank, world, dev)
        counts_all = torch.cat([cnt, cnt_ent])
        if dist is not None:
            all_reduce(counts_all)
        fence()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if dist is not None:
            all_reduce(t, dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if rank == 0:
            c = counts_all.cpu().tolist()
            assert c[0] == args.corpus and c[6] == 0, f"eval counts inconsistent: {c}"
            print(json.dumps({
                "metric": "sentences/sec through POS+NER+dep heads, corpus walk", "value": args.corpus / elapsed,
                "unit": "sentences/s", "n_gpus": world, "steps": corpus.n_batches, "warmup": 2,
                "ms_per_step": elapsed / ((corpus.n_batches + world - 1) // world) * 1e3, "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"BASELINE config 4: {args.corpus} synthetic sentences L={L} in {corpus.n_batches} "
                                       f"distinct batches of {B} drawn on the device, batch-sharded over {world} rank(s), "
                                       "one all-reduce of the eval counts", "batch_per_gpu": B, "seq_len": L,
                           "parallelism": f"sentence-sharded x{world}"},
                "note": "the timed region includes generating every batch on the device (two randn of B x L x 768)",
                "eval_counts": {"sentences": c[0], "words": c[1], "pos_correct": c[2], "ner_correct": c[3],
                                "uas_correct": c[4], "las_correct": c[5], "ner_entities_correct": c[8],
                                "ner_entities_predicted": c[9], "ner_entities_gold": c[10]},
            }), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # The timed region runs the steps JOINED (each waits for the previous one), so that the dominant kernel's launch duration
    # inside it is the kernel's own (roofline.frac).  PHONLP_BENCH_STREAMING=1 times the streaming form instead
    # (annotate_batch(join=False): a batch's tail -- labels at the decoded heads, eval counts -- on the finishing stream, the
    # next batch's first GEMMs beside this batch's MST decode, as in a corpus walk); by default that form is measured in an
    # extra pass and reported as "streaming" beside the headline.
    STREAMING = int(os.environ.get("PHONLP_BENCH_STREAMING", "0"))
    COUNTS_INLINE = os.environ.get("PHONLP_BENCH_COUNTS_INLINE", "0") == "1"
    s_cnt = torch.cuda.Stream()

    streaming = [STREAMING]
    step_no = [0]
    rotate = [True]

    def step():
        ed_, et_ = batches[step_no[0] % N_RESIDENT] if rotate[0] else batches[0]
        step_no[0] += 1
        out = heads.annotate_batch(ed_, ld, et_, lt, join=streaming[0] == 0)
        # the count kernels ADD to persistent device vectors; the entity count only needs the NER tags, so it runs on
        # the tagger's stream behind the Viterbi instead of at the tail of the batch
        with torch.cuda.stream(out["tag_stream"]):
            ops.ner_entity_counts_into(counts_ent, out["ner"], gold_ner, lt, tag_kind, tag_type)
        # the tally of this batch runs on a stream of its own behind the batch's last kernel, so that the next batch's
        # first kernels do not queue behind it (PHONLP_BENCH_COUNTS_INLINE=1: on the batch's finishing stream, A/B)
        if COUNTS_INLINE:
            with torch.cuda.stream(out["stream"]):
                ops.eval_counts_into(counts, out["pos"], out["ner"], out["heads"], out["rels"], gold_pos, gold_ner,
                                     gold_head, gold_rel, lt, ld, out["mst_status"])
        else:
            s_cnt.wait_event(out["done"])
            with torch.cuda.stream(s_cnt):
                for k in ("pos", "ner", "heads", "rels", "mst_status"):
                    out[k].record_stream(s_cnt)
                ops.eval_counts_into(counts, out["pos"], out["ner"], out["heads"], out["rels"], gold_pos, gold_ner,
                                     gold_head, gold_rel, lt, ld, out["mst_status"])
        return out

    engine = ops.get_gemm_engine()
    other_ms = {}
    # PHONLP_BENCH_EXTRAS=0 (tools/profile_round.sh): only the timed steps and the kernels_ms pass, so that a profiler's
    # per-kernel averages over the whole command are those of the step
    EXTRAS = os.environ.get("PHONLP_BENCH_EXTRAS", "1") == "1"
    if world == 1 and EXTRAS:  # the other matrix engines on the same step, reported beside the headline number
        for other in (0, 1, 2):
            if other == engine:
                continue
            ops.set_gemm_engine(other)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(max(args.steps // 2, 1)):
                step()
            torch.cuda.synchronize()
            other_ms[other] = (time.perf_counter() - t0) / max(args.steps // 2, 1) * 1e3
        ops.set_gemm_engine(engine)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()  # the warm-up tails run on the finishing stream
    counts.zero_()
    counts_ent.zero_()
    # Inside the timed region only the dominant kernel (label stage A) is bracketed by HIP events on its launch stream;
    # bracketing every launch costs ~5 us of queue time per kernel.  PHONLP_BENCH_TIME_ALL=1 brackets them all.
    time_all = os.environ.get("PHONLP_BENCH_TIME_ALL", "0") == "1"
    kt = KernelTimer(max_samples=64 * args.steps, only=None if time_all else "linear:label_stage_a")
    kt.__enter__()  # creates the event pool before the timed region
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_last = step()
    torch.cuda.current_stream().wait_stream(out_last["stream"])
    torch.cuda.current_stream().wait_stream(out_last["tag_stream"])
    torch.cuda.current_stream().wait_stream(s_cnt)
    counts_all = torch.cat([counts, counts_ent])
    if dist is not None:
        all_reduce(counts_all)  # the only collective: eval counts over RCCL
    fence()
    elapsed = time.perf_counter() - t0
    kt.__exit__(None, None, None)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        all_reduce(t, dist.ReduceOp.MAX)
    elapsed = float(t.item())
    kernels_ms = {k: round(v[1], 4) for k, v in sorted(kt.stats.items())}
    if not time_all:  # the other kernels' durations: an extra, untimed pass with every launch bracketed (all ranks run
        # it, so that they reach the final barrier together; rank 0 reports its own)
        with KernelTimer(max_samples=64 * 4) as kt_all:
            for _ in range(4):
                step()
            torch.cuda.synchronize()
        kernels_ms = {k: round(v[1], 4) for k, v in sorted(kt_all.stats.items())}

    alone_ms = None
    fused = heads.cfg["label_fused"]
    if rank == 0 and engine == 2 and EXTRAS:  # the dominant kernel with the chip to itself (same shapes and weights; untimed extra)
        w = heads.w
        KP = ops.kp(H)
        x1 = torch.relu(torch.randn(B * L, KP, device=dev))
        x1[:, H] = 1.0
        x1[:, H + 1:] = 0.0
        x1p, x1i = ops.split_rows(x1)
        xf = torch.relu(torch.randn(B * L, 2 * KP, device=dev))
        hd = torch.randint(0, L, (B, L), dtype=torch.int32, device=dev)
        if fused:
            run_a = lambda: ops.label_at_heads_fused(x1p, x1i, w.lab_Wbtc_s[0], w.lab_Wbtc_s[1], xf, hd, ld, H, O)
        else:
            run_a = lambda: ops.label_stage_a_split(x1p, x1i, w.lab_Wbt_s[0], w.lab_Wbt_s[1], B, L, H, O, w.lab_ones_col)
        for _ in range(2):
            run_a()
        torch.cuda.synchronize()
        with KernelTimer(max_samples=32, only="linear:label_stage_a") as kt_alone:
            for _ in range(8):
                run_a()
            torch.cuda.synchronize()
        alone_ms = kt_alone.stats["linear:label_stage_a"][1]

    graph_replay = None
    if rank == 0 and world == 1 and EXTRAS and os.environ.get("PHONLP_BENCH_GRAPH", "1") == "1":
        # the same step (without the eval-count kernels) captured once as a HIP graph and replayed, against eager launches
        # of the same work; untimed extra, reported beside the headline number (which stays the eager step: its dominant
        # kernel is bracketed by HIP events on its launch stream, which a graph replay does not allow)
        cs = heads.capture(B, L)
        cs.enc_dep.copy_(enc_dep); cs.lens_dep.copy_(ld); cs.enc_tag.copy_(enc_tag); cs.lens_tag.copy_(lt)
        def timed(fn, n):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        n_g = max(args.steps, 4)
        graph_replay = {"ms_per_step": timed(cs.replay, n_g),
                        "eager_ms_per_step_same_work": timed(lambda: heads.annotate_batch(enc_dep, ld, enc_tag, lt), n_g),
                        "note": "TaskHeads.capture(B, L).replay(): the step's launches on three streams as one hipGraphLaunch; "
                                "both figures without the eval-count kernels of the headline step"}

    other_form = None
    if rank == 0 and world == 1 and EXTRAS:  # the other step form on the same batches (extra pass, same fences)
        streaming[0] = 1 - STREAMING
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            o_last = step()
        torch.cuda.current_stream().wait_stream(o_last["stream"])
        torch.cuda.current_stream().wait_stream(o_last["tag_stream"])
        torch.cuda.current_stream().wait_stream(s_cnt)
        torch.cuda.synchronize()
        o_ms = (time.perf_counter() - t0) / args.steps * 1e3
        other_form = {"form": "joined" if STREAMING else "streaming", "ms_per_step": o_ms, "value": B / (o_ms * 1e-3),
                      "unit": "sentences/s",
                      "note": "streaming = annotate_batch(join=False): back-to-back steps, a batch's tail (labels at the "
                              "decoded heads, eval counts) overlaps the next batch's first GEMMs, as in the corpus walk "
                              "(bench.py --corpus); every step completes inside the timed pass.  The headline times the "
                              "joined form so that roofline.frac is the dominant kernel's own launch duration"}
        streaming[0] = STREAMING

    single_replay = None
    if rank == 0 and world == 1 and EXTRAS and N_RESIDENT > 1:  # round 2's form: ONE resident batch replayed (cache-warm inputs)
        rotate[0] = False
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            o_last = step()
        torch.cuda.current_stream().wait_stream(o_last["stream"])
        torch.cuda.current_stream().wait_stream(o_last["tag_stream"])
        torch.cuda.current_stream().wait_stream(s_cnt)
        torch.cuda.synchronize()
        r_ms = (time.perf_counter() - t0) / args.steps * 1e3
        single_replay = {"ms_per_step": r_ms, "value": B / (r_ms * 1e-3), "unit": "sentences/s",
                         "note": "the same joined steps on ONE resident batch (its 200 MB of encoder rows stay in the "
                                 "256 MB Infinity Cache); the headline rotates over %d distinct batches" % N_RESIDENT}
        rotate[0] = True

    others = None
    if rank == 0 and world == 1 and EXTRAS and os.environ.get("PHONLP_BENCH_OTHERS", "1") == "1":
        others = other_configs(heads, dev, max(args.steps // 2, 5))

    if rank == 0:
        fl = algorithmic_flops()
        sent = B * args.steps * world
        c = counts_all.cpu().tolist()
        assert c[0] == sent and c[6] == 0, f"eval counts inconsistent: {c}"
        n, mean_ms, min_ms = kt.stats["linear:label_stage_a"]
        achieved = fl["label_stage_a"] * B / (mean_ms * 1e-3) / 1e12
        # engines 1 / 2 spend six bf16 / three f16 MFMA flops per fp32-equivalent flop: their ceilings in algorithmic
        # flops are 2500/6 and 2500/3 (the same fraction as issued 16-bit MFMA flops over the 2500 dense peak)
        peak = {0: PEAK_F32_MFMA_TFLOPS, 1: PEAK_BF16_MFMA_TFLOPS / 6.0, 2: PEAK_BF16_MFMA_TFLOPS / 3.0}[engine]
        eng_name = {0: "f32 MFMA (v_mfma_f32_32x32x2_f32)",
                    1: "bf16x6: fp32 operands split exactly into 3 bf16 pieces, 6 v_mfma_f32_32x32x16_bf16 products, "
                       "f32 accumulate (fp32-grade error, tests/test_gpu_parity.py::test_engines_against_fp64)",
                    2: "f16x3: fp32 operands scaled per row by a power of two and split into 2 f16 pieces once per "
                       "element (split images), 3 f16 MFMA products (32x32x16; the dominant kernel 16x16x32), f32 accumulate (fp32-grade error, "
                       "tests/test_gpu_parity.py::test_engines_against_fp64); small layers and stage B run bf16x6"}
        short = {0: "f32 MFMA", 1: "bf16x6", 2: "f16x3"}
        kern = {0: "linear_kernel<128,128,2,2,1,0>", 1: "linear_split256_kernel<1>",
                2: ("linear_f16x3_wide16_heads_kernel (persistent 256x256 tiles, v_mfma_f32_16x16x32_f16) with the labels-at-heads epilogue (T never stored)" if fused else
                    "linear_f16x3_dma_kernel<1, false, true>")}
        peak_note = {0: "dense fp32 MFMA peak",
                     1: "algorithmic (fp32-equivalent) TFLOP/s; engine 1 peak = 2500 bf16 dense / 6 products",
                     2: "algorithmic (fp32-equivalent) TFLOP/s; engine 2 peak = 2500 f16 dense / 3 products"}
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic = tj.get(f"linear:label_stage_a@engine{engine}" + ("+fused" if fused and engine == 2 else ""))
            traffic_src = tj.get("source")
        # the memory- / latency-bound stages the north star names, achieved HBM GB/s against the 8 TB/s peak: algorithmic
        # bytes over the in-step launch duration of the untimed all-kernels pass
        KP_ = ops.kp(H)
        M_ = B * L
        sec_bytes = {
            "arc_stage_b": M_ * 6 * KP_ * 4 + M_ * 3 * KP_ * 4 + B * L * L * 4,  # X1/X2 of 3 scorers + their stage-A rows in, log-probs out
            "viterbi": B * (L - 1) * T * 4 + T * T * 4 + B * (L - 1) * 4 + B * 4,
            "mst": B * L * L * 4 * 2 + B * L * 4 + B * 4,                         # arc log-probs (read by both decode phases), heads + status out
            "label_at_heads": M_ * O * KP_ * 4 + M_ * KP_ * 4 + M_ * 8,
            "label_reduce": O * (_lib.lib().phonlp_label_kpc(H) // 32) * M_ * 4 + M_ * 4 + M_ * 8,  # per-tile shares + head rows in, ids + scores out
            "tagger_tail": M_ * (U + T + 2) * 4 + M_ * (H + T) * 4,
            "split_rows": M_ * KP_ * 4 * 2,
        }
        secondary = []
        for k, nbytes in sec_bytes.items():
            if k in kernels_ms and kernels_ms[k] > 0:
                gbs = nbytes / (kernels_ms[k] * 1e-3) / 1e9
                secondary.append({"kernel": k, "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": round(gbs / PEAK_HBM_GBS, 4), "ms_in_step": kernels_ms[k],
                                  "algorithmic_bytes": nbytes})
        sec_note = ("viterbi and mst are latency-bound (dependent chains of L steps / ~35 contraction rounds per decode on "
                    "KBs of data): their HBM fraction is reported as the north star asks, the bound that holds is the chain")
        res = {
            "metric": "sentences/sec through POS+NER+dep heads at B=256,L=128",
            "value": sent / elapsed, "unit": "sentences/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo)" if rehearsal else ""),
            "gemm_engine": eng_name[engine],
            "other_engine_ms_per_step": {short[k]: v for k, v in other_ms.items()} if other_ms else None,
            "config": {"workload": "BASELINE config 3: synthetic B=256 L=128 full heads incl. Chu-Liu-Edmonds MST "
                                   "(E=768, H=400, n_rel=33, n_ner=9, n_upos=24), encoder outputs resident in HBM",
                       "batch_per_gpu": B, "seq_len": L, "parallelism": f"sentence-sharded x{world}"},
            "mfma_util_pct_whole_path_executed": 100.0 * fl["executed"] * B * args.steps / elapsed / (peak * 1e12),
            "mfma_util_pct_whole_path_reference_flops": 100.0 * fl["total"] * B * args.steps / elapsed / (peak * 1e12),
            "mfma_util_note": "executed = the FLOPs the timed step multiplies (folded dep_hid / POS embedding, labels at the "
                              "decoded heads); reference_flops counts the reference's own algorithm incl. the dense label "
                              "stage B (111 GFLOP/batch) and dep_hid (49 GFLOP/batch) that this path does not execute",
            "roofline": {"kernel": kern[engine] + " as bilinear stage A of the label scorer",
                         "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "peak_note": peak_note[engine],
                         "launches_timed": n, "mean_ms": mean_ms, "min_ms": min_ms,
                         "note": "timed inside the step, where the MST and tagger kernels share the chip with it "
                                 "on side streams",
                         "algorithmic_flops_per_launch": fl["label_stage_a"] * B,
                         "alone": None if alone_ms is None else {
                             "mean_ms": alone_ms, "achieved": fl["label_stage_a"] * B / (alone_ms * 1e-3) / 1e12,
                             "frac": fl["label_stage_a"] * B / (alone_ms * 1e-3) / 1e12 / peak,
                             "note": "the same kernel, shapes and weights with the chip to itself (X1 = relu(N(0,1))), "
                                     "8 launches after the timed region; `achieved` above is the in-step figure"}},
            "roofline_secondary": secondary, "roofline_secondary_note": sec_note,
            "step_form": "streaming" if STREAMING else "joined",
            "streaming" if not STREAMING else "joined": other_form,
            "graph_replay": graph_replay,
            "single_batch_replay": single_replay,
            "resident_batches_rotated": N_RESIDENT,
            "other_configs": others,
            "kernels_ms": kernels_ms,
            "kernels_ms_note": "mean per launch, every launch bracketed by HIP events" +
                               ("" if time_all else " in an extra untimed pass after the timed region (inside it only "
                                                    "label stage A is bracketed)"),
            "eval_counts": {"sentences": c[0], "words": c[1], "pos_correct": c[2], "ner_correct": c[3],
                            "uas_correct": c[4], "las_correct": c[5], "ner_entities_correct": c[8],
                            "ner_entities_predicted": c[9], "ner_entities_gold": c[10],
                            "ner_entity_prf": [round(v, 4) for v in prf(c[8], c[9], c[10])]},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(sd, enc_dep_h, enc_tag_h, lens_dep, lens_tag)
            res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
