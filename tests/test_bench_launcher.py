"""bench.py's own N > 1 launch path on CPU: `python bench.py --gpus 2` with no external launcher must start two
ranks itself, form a process group of that size, shard / gather, and print ONE JSON line with n_gpus = 2 on rank 0.
--rehearse-cpu swaps the backend for gloo and the HIP batch for the CPU oracle (there is no GPU here); everything else
-- argument handling, self-launch, rank environment, sharding, the all-gather, the cross-rank checks, the line -- is the
code that runs on the 8-GPU node with the nccl (RCCL) backend."""
import json
import os
import subprocess
import sys

import pytest

import oracle_lib as O
from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


@pytest.mark.parametrize("workload", ["c3", "c4"])
def test_self_launch_two_ranks_gloo(workload):
    pr = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", workload, "--small", "--rehearse-cpu"])
    assert pr.returncode == 0, pr.stderr.decode()[-2000:]
    lines = [l for l in pr.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout.decode()   # ONE line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["dist"]["world_size"] == 2 and line["dist"]["backend"] == "gloo"
    assert len(line["dist"]["device_of_rank"]) == 2
    assert line["dist"]["gathered_identical_on_all_ranks"] is True
    assert line["scaling"] == ("weak" if workload == "c3" else "strong")
    assert "invalid" in line   # a rehearsal is never a measurement
    if workload == "c4":
        # strong scaling: the pair list is split over the ranks and the gathered vector is the whole upper triangle
        seqs = [O.gen(1, 2, i, 120) for i in range(12)]
        want = sum(O.score("nw", seqs[i], seqs[j], 1, -1, -1)[0] for i in range(12) for j in range(i + 1, 12))
        assert line["dist"]["gathered_score_sum"] == want
    else:
        # weak scaling: rank r aligns the same patterns against ITS OWN texts; the gathered vector holds both shards
        pats = [O.gen(1, 0, p, 150) for p in range(24)]
        want = 0
        for r in range(2):
            txts = [O.gen(1, 1, r * 4 + t, 300) for t in range(4)]
            want += sum(O.score("sw", p, t, 1, -1, -1)[0] for p in pats for t in txts)
        assert line["dist"]["gathered_score_sum"] == want


@pytest.mark.parametrize("workload,mode", [("g", "nw"), ("gb", "sw")])
def test_full_alignment_batches_shard_over_ranks(workload, mode):
    """r03: --workload g / gb honour --gpus N: the list of N x 4096 pairs is dealt in contiguous blocks (rank r: block r), every rank
    aligns its block in full, ONE all-gather per step carries the per-pair scores and op counts of all blocks to every rank."""
    pr = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", workload, "--small", "--rehearse-cpu"])
    assert pr.returncode == 0, pr.stderr.decode()[-2000:]
    lines = [l for l in pr.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout.decode()
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "invalid" in line
    assert line["dist"]["world_size"] == 2 and line["dist"]["backend"] == "gloo" and line["dist"]["gathered_identical_on_all_ranks"] is True
    txts = [O.gen(1, 1, t, 200) for t in range(4)]
    want = sum(O.score(mode, O.gen(1, 0, r * 8 + p, 40), txts[p % 4], 1, -1, -1)[0] for r in range(2) for p in range(8))
    assert line["dist"]["gathered_score_sum"] == want
    assert line["verified_vs_cpu"]["bit_exact"] is True


def test_gpus_must_match_world_size():
    pr = _run(["--gpus", "2", "--small", "--rehearse-cpu"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert pr.returncode != 0 and b"WORLD_SIZE=1" in pr.stderr
    pr = _run(["--gpus", "1", "--small", "--rehearse-cpu"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert pr.returncode != 0 and b"WORLD_SIZE=2" in pr.stderr


def test_a_failing_rank_fails_the_launch():
    """rank 1 dies before the process group forms: the parent must stop rank 0 and exit non-zero, not hang."""
    pr = _run(["--gpus", "2", "--small", "--rehearse-cpu", "--workload", "c3"], {"BENCH_TEST_FAIL_RANK": "1"}, timeout=120)
    assert pr.returncode != 0
    assert not [l for l in pr.stdout.decode().splitlines() if l.startswith("{")]
