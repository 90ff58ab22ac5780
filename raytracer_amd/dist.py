"""Multi-GPU glue: one process per GPU, samples partitioned by index, films summed once with an all-reduce
(torch.distributed: backend "nccl" is RCCL over xGMI on MI355X, "gloo" on CPU for tests).

The reference is single-process; this is the SURVEY.md §8e design: every rank holds a full scene replica
(~0.4 GB per million triangles, against 288 GB), renders a disjoint range of global sample indices into its own
fp32 film, and the un-normalised films (Film::film, Imaging.h:219) plus their SPP counts are summed.  The RNG key is
a function of (seed, global sample index, pixel), so the union of the ranks' samples does not depend on the rank count.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def sample_range(total_spp: int, rank: int, world: int):
    """Contiguous, balanced slice [first, first+count) of the global sample indices for `rank`."""
    base, extra = divmod(int(total_spp), int(world))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def allreduce_film(film: torch.Tensor, local_spp: int) -> int:
    """In-place sum of the accumulation film over ranks; returns the summed SPP (Film::SPP)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(local_spp)
    dist.all_reduce(film, op=dist.ReduceOp.SUM)
    n = torch.tensor([int(local_spp)], dtype=torch.int64, device=film.device)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return int(n.item())


def render_distributed(rt, mode: int, total_spp: int, seed: int, film: torch.Tensor) -> int:
    """Renders this rank's share of `total_spp` into `film` (a device tensor bound with rt.bind_film) and reduces."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    first, count = sample_range(total_spp, rank, world)
    rt.bind_film(film.data_ptr())
    if count:
        rt.render(mode, count, seed, first)
    rt.sync()
    spp = allreduce_film(film, count)
    rt.set_spp(spp)
    return spp
