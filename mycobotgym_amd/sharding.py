"""Multi-GPU layout of the rollout: contiguous env-index blocks per rank, no exchange on the step path.

Environments are independent, so N GPUs hold N disjoint shards; reset/DR random streams are keyed by the GLOBAL
env id (``env_id_offset + local index``), which makes every env's trajectory independent of how many ranks share
the job.  The only collective is the reduction of episode statistics for logging (bytes, latency-bound), issued
through ``torch.distributed`` -- RCCL over xGMI on the GPUs (backend "nccl"), gloo in the CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard(rank: int, world: int, envs_per_rank: int):
    """-> (env_id_offset, total_envs) for weak scaling with a fixed per-rank env count."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return rank * envs_per_rank, world * envs_per_rank


def reduce_episode_stats(ep_return: torch.Tensor, ep_length: torch.Tensor, is_success: torch.Tensor,
                         done: torch.Tensor, group=None, device=None) -> dict:
    """Sum over all ranks of [finished episodes, their returns, lengths, successes] -> host dict.
    One 32-byte all_reduce; call it once per logging interval, never per step.  `device`: where the four numbers are
    reduced (default: where the inputs live -- the GPU, i.e. RCCL; "cpu" for a gloo group)."""
    d = done.to(torch.float64)
    stats = torch.stack([d.sum(), (ep_return.double() * d).sum(), (ep_length.double() * d).sum(),
                         (is_success.double() * d).sum()])
    if device is not None:
        stats = stats.to(device)
    if dist.is_available() and dist.is_initialized():        # also at world size 1: the call the multi-GPU job makes is the call that is tested
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    n, r, l, s = stats.tolist()
    return {"episodes": n, "mean_return": r / n if n else float("nan"), "mean_length": l / n if n else float("nan"),
            "success_rate": s / n if n else float("nan")}
