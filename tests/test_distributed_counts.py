"""Multi-GPU logic on CPU: 2 gloo ranks shard a corpus by sentence batches (no data-path collective) and
all-reduce the eval-count vector exactly like bench.py does over RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from phonlp_amd.sharding import eval_counts, shard_batches


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_batches, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(5)
    # the same synthetic "predictions" and gold on every rank; each rank scores only its own batches
    B, L = 8, 12
    pred = torch.randint(0, 4, (n_batches, 4, B, L), generator=g, dtype=torch.int32)
    gold = torch.randint(0, 4, (n_batches, 4, B, L), generator=g, dtype=torch.int32)
    lens = torch.randint(2, L + 1, (n_batches, B), generator=g, dtype=torch.int32)
    counts = torch.zeros(8, dtype=torch.int64)
    for i in shard_batches(n_batches, rank, world):
        p, q = pred[i], gold[i]
        counts += eval_counts(p[0], p[1], p[2], p[3], q[0], q[1], q[2], q[3], lens[i], lens[i])
    dist.all_reduce(counts)
    if rank == 0:
        ref = torch.zeros(8, dtype=torch.int64)
        for i in range(n_batches):
            p, q = pred[i], gold[i]
            ref += eval_counts(p[0], p[1], p[2], p[3], q[0], q[1], q[2], q[3], lens[i], lens[i])
        out.put((counts.tolist(), ref.tolist()))
    dist.destroy_process_group()


def test_two_rank_sharding_and_count_reduction():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, out)) for r in range(2)]
    for p in procs:
        p.start()
    got, ref = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == ref and got[0] == 7 * 8


def test_shards_partition_the_corpus():
    for world in (1, 2, 4, 8):
        seen = sorted(i for r in range(world) for i in shard_batches(3907, r, world))
        assert seen == list(range(3907))  # BASELINE config 4: 1M sentences = 3907 batches of 256
        sizes = [len(list(shard_batches(3907, r, world))) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
