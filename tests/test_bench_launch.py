"""`bench.py --gpus N` starts its own N ranks (the flag used to be parsed and ignored).  CPU: the launcher,
the rendezvous, the batch sharding and the count reduction run over gloo in the script's dry-run mode (no head work,
no GPU); the same launch with real head work on one GPU is tests/test_gpu_bench_launch.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, extra_env=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PHONLP_BENCH_DRYRUN="1", **(extra_env or {}))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, timeout=timeout)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks():
    rc, out, err = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert rc == 0, err
    j = _line(out)
    assert j["n_gpus"] == 2 and j["steps"] == 3
    assert j["eval_counts"]["sentences"] == 2 * 3 * 256  # weak scaling: every rank runs its own steps


def test_gpus_1_stays_one_process():
    rc, out, err = _run(["--gpus", "1", "--steps", "2"])
    assert rc == 0, err
    j = _line(out)
    assert j["n_gpus"] == 1 and j["eval_counts"]["sentences"] == 2 * 256


def test_corpus_is_sharded_over_the_ranks():
    rc, out, err = _run(["--gpus", "2", "--corpus", "3000"])
    assert rc == 0, err
    j = _line(out)
    assert j["n_gpus"] == 2 and j["steps"] == 12 and j["eval_counts"]["sentences"] == 3000


def test_mismatch_with_an_external_launcher_is_rejected():
    rc, out, err = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and "WORLD_SIZE=2" in err and out == ""


def test_a_failing_rank_fails_the_launch():
    # rank 1 dies before the rendezvous; the launcher must end rank 0 (stuck in init_process_group) and return non-zero
    rc, out, err = _run(["--gpus", "2", "--steps", "1"], {"PHONLP_BENCH_DRYRUN_FAIL_RANK": "1"}, timeout=120)
    assert rc != 0 and "rank 1 exited" in err
