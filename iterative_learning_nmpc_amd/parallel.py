"""Sharding of independent rollouts over the GPUs of one node and the one exchange step.

The reference runs its rollouts in plain sequential loops
(DAgger/example/data_collection_pretrain_omini_vc_policy_1direction_perturbed.py:202-247) and
reduces them with the tracking-error / OOD rule
(Behavior_Cloning/utils/data_collection_force_perturbation.py:123-158,
 Behavior_Cloning/examples/test_train_policy.py:127-134).  Here rollouts are independent problems:
  - contiguous shards, B/G rollouts per rank, one process per GPU, no collective in the solve path;
  - ONE all-gather per learning iteration of the per-rollout tracking errors (fp32 [B/G, K]),
    RCCL over xGMI on GPUs (backend "nccl"), gloo on CPU for tests;
  - every rank then evaluates the same update (OOD mask, sampling weights) -- no further traffic.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of `total` rollouts; the first `total % world` ranks get one more."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_tracking_errors(err_local: torch.Tensor, total: int) -> torch.Tensor:
    """Gather the [B_local, K] tracking errors of all ranks into the full [total, K] tensor, in
    rollout order.  Equal shards go through one all_gather_into_tensor (a single direct exchange
    on the xGMI mesh); ragged shards are padded to the largest shard first."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        assert err_local.shape[0] == total
        return err_local
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(total, rank, world)
    if err_local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank}: shard has {err_local.shape[0]} rollouts, expected {hi - lo}")
    K = err_local.shape[1]
    largest = shard_bounds(total, 0, world)
    n_max = largest[1] - largest[0]
    send = err_local.contiguous()
    if send.shape[0] != n_max:
        send = torch.cat([send, send.new_zeros(n_max - send.shape[0], K)])
    if send.is_cuda and dist.get_backend() == "gloo":       # rehearsal on a box with fewer GPUs than ranks
        host = send.cpu()
        recv_h = host.new_empty(world * n_max, K)
        dist.all_gather_into_tensor(recv_h, host)
        recv = recv_h.to(send.device)
    else:
        recv = send.new_empty(world * n_max, K)
        dist.all_gather_into_tensor(recv, send)
    if total == world * n_max:
        return recv
    parts = []
    for r in range(world):
        a, b = shard_bounds(total, r, world)
        parts.append(recv[r * n_max: r * n_max + (b - a)])
    return torch.cat(parts)


OOD_THRESHOLD_REFERENCE = 4.0     # on the reference's 44-slot state row (data_collection_force_perturbation.py:149)


def ood_threshold(n_state: int) -> float:
    """The out-of-distribution threshold for a recorded state row of `n_state` slots.

    The reference's 4.0 is a Euclidean distance over the 43 non-phase slots of its row [phase, v(18), q[2:](17),
    base_wrt_feet(8)] (DAgger/utils/RolloutMPC.py:221), i.e. a per-slot RMS deviation of 4.0 / sqrt(43) = 0.61.  The
    centroidal rollouts record the 19-slot sub-vector that plant has ([phase, v(6), z, yaw, pitch, roll,
    base_wrt_feet(8)], include/nmpc.h) [decl]; the threshold is mapped by keeping that per-slot RMS deviation:
    4.0 sqrt((n_state - 1) / 43) -- 2.59 for 19 slots, 4.0 for 44."""
    return OOD_THRESHOLD_REFERENCE * ((n_state - 1) / 43.0) ** 0.5


def all_gather_validity(valid_local: torch.Tensor, total: int) -> torch.Tensor:
    """The ranks' per-rollout validity flags (bool [B_local]: the rollout ran to the end, nothing in it was discarded) as
    one bool [total] in rollout order -- rides the same exchange as the errors, so that every rank masks the same
    rollouts in the update."""
    return all_gather_tracking_errors(valid_local.to(torch.float32).unsqueeze(1), total).squeeze(1) > 0.5


def learning_update(err_all: torch.Tensor, threshold: float = 4.0, ood_weight: float = 5.0,
                    valid: Optional[torch.Tensor] = None):
    """OOD mask and sampling weights from the gathered errors (identical on every rank).

    valid: bool [total] per rollout -- a rollout that was terminated early and could not be redone is not data: the
    reference deletes its file (RolloutMPC.py:432-435), here it gets sampling weight 0 and is never out-of-distribution.
    A non-finite error (a rollout whose solver failed) is treated the same way whether or not `valid` says so -- compared
    with the threshold it would silently read as in-distribution, weight 1."""
    ok = torch.isfinite(err_all)
    if valid is not None:
        ok = ok & valid.reshape(-1, *([1] * (err_all.dim() - 1)))
    ood = (err_all > threshold) & ok
    weights = torch.where(ood, err_all.new_full((), ood_weight), err_all.new_ones(()))
    weights = torch.where(ok, weights, err_all.new_zeros(()))
    return ood, weights
