"""bench.py's OWN launcher plumbing at world size 2, on the CPU over gloo: no 8-GPU node was ever available to
the builder, so the first real multi-GPU launch must not be the first time this code runs.  bench.py is started exactly as the driver
starts it -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2
--steps K --warmup W` -- with BENCH_REHEARSE_PLUMBING=1, which swaps ONLY the engine for a stand-in that sleeps (no GPU here): rank /
world parsing, the --gpus check, shard offsets, the barrier + synchronise brackets, the MAX over ranks of the wall time, the logging
collective (reduce_episode_stats) and "one JSON line, rank 0 only" are bench.py's real code paths."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(nproc, gpus, steps=12, warmup=3, extra_env=None):
    env = dict(os.environ, BENCH_REHEARSE_PLUMBING="1", OMP_NUM_THREADS="1")
    env.update(extra_env or {})
    port = 29600 + os.getpid() % 1000 + nproc
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", str(steps), "--warmup", str(warmup)]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)


def test_two_ranks_one_json_line_max_time_and_shards():
    steps = 12
    r = _launch(2, 2, steps=steps)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line (rank 0 only) expected, got {len(lines)}"
    j = json.loads(lines[0])
    assert j["rehearsal"] is True and j["value"] is None and j["n_gpus"] == 2 and j["steps"] == steps and j["scaling"] == "weak"
    assert j["config"]["total_envs"] == 2 * j["config"]["envs_per_gpu"] == 2 * 8192
    assert j["config"]["parallelism"].startswith("env-sharded x2")
    # the stand-in sleeps 1 ms per step on rank 0 and 2 ms on rank 1: the reported time is the MAX over the ranks
    assert j["ms_per_step"] >= 2.0 * 0.95, j["ms_per_step"]
    # the logging collective summed both ranks' finished episodes: 8192 episodes of return 1 (rank 0) and 8192 of return 2 (rank 1)
    st = j["episode_stats_last_step"]
    assert st["episodes"] == 2 * 8192 and abs(st["mean_return"] - 1.5) < 1e-12 and st["mean_length"] == 50


def test_gpus_flag_must_match_the_world():
    r = _launch(2, 4)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in (r.stderr + r.stdout)
    env = dict(os.environ, BENCH_REHEARSE_PLUMBING="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode != 0 and "torch.distributed.run" in (r.stderr + r.stdout)          # told how to launch N > 1


def test_single_process_rehearsal_line():
    env = dict(os.environ, BENCH_REHEARSE_PLUMBING="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2"], env=env, capture_output=True, text=True,
                       timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["rehearsal"] is True and "secondary" not in j and "cpu_baseline" not in j
