"""N > 1 path on CPU: two gloo ranks, each stepping its shard (global env ids via env_id_offset), statistics reduced
with the same helper bench.py / training loops use.  Shards must reproduce the single-process run exactly."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.common import make_oracle

N_PER, STEPS = 48, 55


def _run(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mycobotgym_amd.sharding import shard, reduce_episode_stats
    off, total = shard(rank, world, N_PER)
    ora = make_oracle(N_PER, controller_type="joint", reward_type="dense", seed=5, env_id_offset=off, n_threads=2)
    ora.reset(seed=5)
    rng = np.random.default_rng(9)
    acc = torch.zeros(4, dtype=torch.float64)
    for t in range(STEPS):
        a_all = rng.uniform(-1, 1, (total, 7)).astype(np.float32)       # same global action table on every rank
        o = ora.step(a_all[off:off + N_PER])
        done = torch.as_tensor(o["truncated"].astype(bool))
        st = reduce_episode_stats(torch.as_tensor(o["ep_return"]), torch.as_tensor(o["ep_length"]),
                                  torch.as_tensor(o["is_success"]), done)
        acc += torch.tensor([st["episodes"], st["episodes"] * (st["mean_return"] if st["episodes"] else 0.0),
                             st["episodes"] * (st["mean_length"] if st["episodes"] else 0.0), 0.0])
    np.save(os.path.join(out, f"obs{rank}.npy"), o["obs"])
    if rank == 0:
        np.save(os.path.join(out, "acc.npy"), acc.numpy())
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_process(built, tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_run, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    # single process, all 2 * N_PER envs
    ora = make_oracle(2 * N_PER, controller_type="joint", reward_type="dense", seed=5, env_id_offset=0)
    ora.reset(seed=5)
    rng = np.random.default_rng(9)
    episodes = ret = length = 0.0
    for t in range(STEPS):
        o = ora.step(rng.uniform(-1, 1, (2 * N_PER, 7)).astype(np.float32))
        d = o["truncated"].astype(bool)
        episodes += d.sum(); ret += o["ep_return"][d].sum(); length += o["ep_length"][d].sum()
    got = np.concatenate([np.load(tmp_path / "obs0.npy"), np.load(tmp_path / "obs1.npy")])
    assert np.array_equal(got, o["obs"])                     # bit-identical: trajectories do not depend on the sharding
    acc = np.load(tmp_path / "acc.npy")
    assert acc[0] == episodes == 2 * N_PER and np.isclose(acc[1], ret) and np.isclose(acc[2], length)


def test_shard_helper():
    from mycobotgym_amd.sharding import shard
    assert shard(0, 8, 8192) == (0, 65536) and shard(7, 8, 8192) == (57344, 65536)
    with pytest.raises(ValueError):
        shard(8, 8, 8192)
