"""Whole-swarm observation packing across the GPUs of a node (optional).

Env shards are independent -- the step itself never communicates.  A consumer that wants every env's observation in
one place (a logger, a learner) gathers the ranks' ``[E_local, D, 20]`` blocks with one collective:
``torch.distributed.all_gather_into_tensor`` = ncclAllGather on RCCL over xGMI when the tensors live on the GPUs
(gloo in the CPU tests).  Never called by the step path."""
from __future__ import annotations

import torch


def all_gather_observations(obs: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """obs [E_local, D, 20] on this rank -> [world * E_local, D, 20] on every rank (rank-major env order).
    Every rank must pass the same shape and dtype.  With no process group it returns ``obs`` itself."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return obs
    world = dist.get_world_size()
    obs = obs.contiguous()
    if out is None:
        out = torch.empty((world * obs.shape[0],) + tuple(obs.shape[1:]), dtype=obs.dtype, device=obs.device)
    if dist.get_backend() == "gloo" and not hasattr(dist, "all_gather_into_tensor"):
        parts = [torch.empty_like(obs) for _ in range(world)]
        dist.all_gather(parts, obs)
        out.copy_(torch.cat(parts, dim=0))
        return out
    try:
        dist.all_gather_into_tensor(out, obs)
    except (RuntimeError, NotImplementedError):          # older gloo builds: no flat-tensor all-gather
        parts = [torch.empty_like(obs) for _ in range(world)]
        dist.all_gather(parts, obs)
        out.copy_(torch.cat(parts, dim=0))
    return out
