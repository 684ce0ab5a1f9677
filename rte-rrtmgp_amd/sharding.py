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


#: shard boundaries fall on multiples of this many columns (one wavefront = 512 bytes of a row of doubles) when every rank
#: still gets at least ALIGN_MIN_SHARD columns
ALIGN_COLUMNS = 64
ALIGN_MIN_SHARD = 1024


def shard_columns(ncol_global: int, rank: int, world: int, align: int | None = None) -> Tuple[int, int]:
    """(first column, number of columns) owned by ``rank``: contiguous ranges that cover 0 ... ncol_global - 1 in rank order.

    ``align``: shard boundaries are multiples of that many columns (the last rank takes what remains), sizes differ by less
    than ``2 * align``.  Default: 64 when every rank gets at least 1024 columns, else 1 (sizes differ by at most one).
    Why: the arrays of the kernel interface are dense (ncol_local, nlay, ...) with the column fastest, so a rank's rows start
    ``8 * ncol_local`` bytes apart.  1e6 columns on 8 ranks are 125 000 each -- rows 1 000 000 bytes apart, 64-byte but not
    128-byte aligned: every second row's 512-byte wave request straddles five cache lines instead of four, and every kernel of
    the chain is 4-8 % slower (measured, one MI355X, LW chain: 5.28 M columns/s at 125 000 columns against 5.54-5.56 M at
    124 992 / 125 056 / 128 000).  Shards of 125 056 + 7 x 124 992 columns avoid that."""
    if align is None:
        align = ALIGN_COLUMNS if ncol_global >= world * ALIGN_MIN_SHARD else 1
    if align <= 1:
        base, rem = divmod(ncol_global, world)
        start = rank * base + min(rank, rem)
        return start, base + (1 if rank < rem else 0)
    nblk = -(-ncol_global // align)  # blocks of `align` columns, the last one possibly short
    base, rem = divmod(nblk, world)
    b0 = rank * base + min(rank, rem)
    b1 = b0 + base + (1 if rank < rem else 0)
    start, end = min(b0 * align, ncol_global), min(b1 * align, ncol_global)
    return start, end - start


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
