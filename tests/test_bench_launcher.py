"""bench.py --gpus N stands alone (round-3 verdict item 2): without a launcher environment the parent spawns the N ranks itself,
relays ONE line with n_gpus == N, and every mismatch between --gpus and the ranks that exist is a non-zero exit.  CPU only:
--launcher-selftest runs the distributed plumbing (gloo) and no compute."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ORBFE_BENCH_FAIL_RANK")}
    env["ORBFE_DIST_TIMEOUT_S"] = "60"
    env.update(kw)
    return env


def _run(args, **kw):
    return subprocess.run([sys.executable, BENCH] + args, env=_env(**kw), capture_output=True, text=True, timeout=300)


def test_self_launch_two_ranks_prints_one_line_with_n_gpus_2():
    r = _run(["--gpus", "2", "--launcher-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(rows) == 1, r.stdout
    out = json.loads(rows[0])
    assert out["n_gpus"] == 2
    assert out["config"]["rccl"]["ranks"] == 2 and out["config"]["rccl"]["self_launched"] is True
    assert out["config"]["rccl"]["broadcast_bytes"] > 4096  # parameters + pattern checksum + the vocabulary blob
    # the strong-scaling pass with step chains in flight: every (step, pair) enqueued exactly once over both ranks, step k on chain k % C
    assert out["strong_cover_exactly_once"] is True and out["strong_chains"] == 4


def test_strong_pass_with_chains_in_flight_covers_every_pair_exactly_once():
    """bench.py's strong_pass: P pairs per step in total, dealt by dist.shard_pairs; steps dealt to C chains by dist.chain_schedule."""
    sys.path.insert(0, ROOT)
    from orbslam2_amd import dist as D
    for world in (1, 2, 4, 8):
        for chains in (1, 3, 4):
            seen = {}
            for rank in range(world):
                for k, c in D.chain_schedule(20, chains):
                    assert c == k % chains
                    for p in D.shard_pairs(64, rank, world):
                        seen[(k, p)] = seen.get((k, p), 0) + 1
            assert len(seen) == 20 * 64 and set(seen.values()) == {1}


def test_launcher_environment_with_fewer_ranks_than_gpus_is_an_error():
    # what the driver's N = 1 style start looked like to round 3's bench.py when a launcher HAD set WORLD_SIZE=1: it must not report n_gpus 1
    r = _run(["--gpus", "2", "--launcher-selftest"], RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""


def test_a_rank_that_dies_before_the_rendezvous_fails_the_whole_run():
    r = _run(["--gpus", "2", "--launcher-selftest"], ORBFE_BENCH_FAIL_RANK="1")
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert r.stdout.strip() == ""
    assert "rank 1 exited with code 3" in r.stderr


def test_single_rank_selftest_needs_no_launcher():
    r = _run(["--gpus", "1", "--launcher-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip())
    assert out["n_gpus"] == 1 and out["config"]["rccl"]["self_launched"] is False
