"""Data parallelism over the GPUs of one node: one process per GPU, the batch sharded over
ranks (every image keeps all of its agents on one GPU - the only cross-row operation,
the message mean, reduces over agents of the SAME image: networks/message.py:17), weights
replicated, and ONE all-reduce of the flat fp32 gradient buffer per iteration
(``torch.distributed`` backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed code: this is new.  Equal shard sizes
make the mean over (agents, batch) of training/trainer.py:111 equal to the average of the
per-shard means, so averaging gradients reproduces the big-batch gradient, except for
``standardize`` (functions.py:54-55) whose statistics are per shard by default (what DDP
on the reference would do) and global with ``exact_standardize`` (one extra 3-double
all-reduce, C ABI phases 1/2 of marl_a2c_loss_fwd_bwd).
"""

from __future__ import annotations

from typing import Tuple

import torch as th
import torch.distributed as dist


def shard_seed(base: int, rank: int) -> int:
    """Rank-offset seeds so that shards draw different positions / noise."""
    return base * 1_000_003 + 7919 * rank + 1


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Equal contiguous shards; the batch must divide evenly (keeps means exact)."""
    if n % world != 0:
        raise ValueError(f"global batch {n} is not divisible by world size {world}")
    per = n // world
    return rank * per, (rank + 1) * per


def _all_reduce_sum(t: th.Tensor, group=None) -> th.Tensor:
    """Sum-all-reduce in place.  GPU tensors go straight to RCCL ("nccl" backend); under a
    "gloo" group (CPU-only rendezvous, the 2-rank tests that share one GPU) they bounce through
    the host."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class GradAllReduce:
    """Sum-all-reduce of the flat gradient buffer; returns the scale (1 / world) that the
    Adam kernel applies while it reads the gradient (no separate divide pass)."""

    def __init__(self, world: int, group=None) -> None:
        self.world = world
        self.group = group

    def __call__(self, flat_grads: th.Tensor) -> float:
        _all_reduce_sum(flat_grads, self.group)
        return 1.0 / self.world


def allreduce_adv_stats(stats: th.Tensor, group=None) -> th.Tensor:
    """(n, sum, sum of squares) of the advantages summed over ranks: the exchange step of
    the exact global ``standardize``."""
    return _all_reduce_sum(stats, group)


def broadcast_parameters(flat_params: th.Tensor, src: int = 0, group=None) -> None:
    """Identical initial weights on every rank: one broadcast of the flat parameter buffer."""
    if flat_params.is_cuda and dist.get_backend(group) == "gloo":
        host = flat_params.cpu()
        dist.broadcast(host, src=src, group=group)
        flat_params.copy_(host)
    else:
        dist.broadcast(flat_params, src=src, group=group)


def allreduce_confusion(conf_mat: th.Tensor, group=None) -> th.Tensor:
    """Evaluation metrics over the WHOLE evaluation set: sum the per-rank confusion matrices."""
    return _all_reduce_sum(conf_mat, group)
