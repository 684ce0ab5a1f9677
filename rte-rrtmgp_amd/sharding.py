"""Column sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" = RCCL on
ROCm, "gloo" on CPU in the tests).

The hot path has no cross-column dependence (every recurrence runs over layers of one column), so
columns shard trivially: contiguous ranges per rank, k-distribution tables replicated, no data-path
collective.  The only exchange is the reduction of broadband flux diagnostics: the domain-mean
flux profile (all-reduce of 2 x (nlay+1) sums) and, when a caller wants the assembled field, an
all-gather of the per-rank (ncol_local, nlay+1) slabs.
"""
from __future__ import annotations

from typing import Tuple


def shard_columns(ncol_global: int, rank: int, world: int) -> Tuple[int, int]:
    """(first column, number of columns) owned by ``rank``: contiguous, sizes differ by at most 1."""
    base, rem = divmod(ncol_global, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def allreduce_mean_profile(flux_up, flux_dn, ncol_global: int, group=None):
    """Domain-mean broadband flux profiles from per-rank fluxes.

    ``flux_up``/``flux_dn``: torch tensors holding the Fortran array (ncol_local, nlay+1), i.e. of
    torch shape (nlay+1, ncol_local).  Returns a (2, nlay+1) tensor, identical on every rank."""
    import torch
    import torch.distributed as dist

    prof = torch.stack([flux_up.sum(dim=1), flux_dn.sum(dim=1)])
    if dist.is_available() and dist.is_initialized():
        if prof.is_cuda and dist.get_backend(group) == "gloo":  # (ranks sharing one device in the plumbing test: through the host)
            host = prof.cpu()
            dist.all_reduce(host, group=group)
            prof = host.to(prof.device)
        else:
            dist.all_reduce(prof, group=group)
    return prof / float(ncol_global)


def allgather_fluxes(flux, ncol_global: int, group=None):
    """Assemble the global (ncol_global, nlay+1) field (torch shape (nlay+1, ncol_global)) on every rank
    from contiguous per-rank slabs of possibly different widths."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return flux
    world = dist.get_world_size(group)
    nlev = flux.shape[0]
    widths = [shard_columns(ncol_global, r, world)[1] for r in range(world)]
    wmax = max(widths)
    via_host = flux.is_cuda and dist.get_backend(group) == "gloo"
    pad = torch.zeros(nlev, wmax, dtype=flux.dtype, device=("cpu" if via_host else flux.device))
    pad[:, : flux.shape[1]] = flux
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:, :w] for p, w in zip(parts, widths)], dim=1).to(flux.device)
