"""bench.py's own multi-rank launcher on CPU: `python bench.py --gpus 2 --engine stub` re-executes itself under
torch.distributed.run with two gloo ranks, shards the batch (strong scaling), broadcasts the speaker bundle, gathers the
waveforms on rank 0 and prints ONE JSON line.  The engine is a labelled stub -- the point is the harness around it."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--engine", "stub", "--steps", "2", "--warmup", "1",
                        "--gen-tokens", "8", "--text-tokens", "16"] + extra, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout            # stdout carries exactly one line
    return json.loads(lines[0])


def test_two_ranks_strong_scaling():
    j = _run(["--gpus", "2", "--utts", "6"])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["engine"] == "stub"
    assert j["config"]["global_batch"] == 6 and j["config"]["per_gpu_batch"] == 3
    assert j["config"]["parallelism"] == "utterance-dp2"
    assert j["stub_rows_ok"] is True
    assert j["steps"] == 2 and j["warmup"] == 1 and j["value"] > 0
    assert j["metric"].startswith("STUB")       # can never be mistaken for a measurement


def test_two_ranks_weak_scaling_and_single_rank():
    j = _run(["--gpus", "2", "--utts", "3", "--weak"])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["global_batch"] == 6 and j["stub_rows_ok"] is True
    j1 = _run(["--gpus", "1", "--utts", "5"])
    assert j1["n_gpus"] == 1 and j1["config"]["global_batch"] == 5 and j1["config"]["per_gpu_batch"] == 5 and j1["stub_rows_ok"] is True


def test_eight_ranks_uneven_shares():
    """The real rank count of the metric's last point: 8 gloo ranks, 13 utterances -> shares of 2 and 1 (uneven gather), and the BASELINE split
    64 -> 8 per rank; every utterance reaches rank 0 in utterance order."""
    j = _run(["--gpus", "8", "--utts", "13"])
    assert j["n_gpus"] == 8 and j["scaling"] == "strong" and j["config"]["global_batch"] == 13
    assert j["config"]["per_gpu_batch"] == 2 and j["stub_rows_ok"] is True and j["config"]["parallelism"] == "utterance-dp8"
    j = _run(["--gpus", "8", "--utts", "64"])
    assert j["config"]["per_gpu_batch"] == 8 and j["stub_rows_ok"] is True


def test_lpt_sharding_at_eight_ranks():
    """`shard_utterances` with ragged lengths at world size 8: a partition of all utterances, deterministic, and balanced to within the longest
    utterance (the LPT bound) -- what keeps the ranks of a ragged batch finishing together."""
    import random
    from indextts_amd import dist as D
    rnd = random.Random(5)
    lengths = [rnd.randint(20, 128) for _ in range(64)]
    shares = [D.shard_utterances(64, r, 8, lengths=lengths) for r in range(8)]
    assert sorted(i for sh in shares for i in sh) == list(range(64))
    loads = [sum(lengths[i] for i in sh) for sh in shares]
    assert max(loads) - min(loads) <= max(lengths)
    assert shares == [D.shard_utterances(64, r, 8, lengths=lengths) for r in range(8)]
    assert [len(D.shard_utterances(64, r, 8, lengths=[128] * 64)) for r in range(8)] == [8] * 8


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--engine", "stub", "--gpus", "4"], capture_output=True,
                       text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_overlapped_steps_two_ranks():
    """`--overlap`: the two-stage software pipeline (decode of step k + 1 on a worker thread, render of step k on the main thread; collectives
    only from the main thread) delivers the same rows as the sequential loop, on one rank and on two."""
    j = _run(["--gpus", "2", "--utts", "6", "--overlap"])
    assert j["n_gpus"] == 2 and j["stub_rows_ok"] is True and j["config"]["step_overlap"] is True and j["steps"] == 2
    j1 = _run(["--gpus", "1", "--utts", "5", "--overlap"])
    assert j1["stub_rows_ok"] is True and j1["config"]["step_overlap"] is True
