"""Multi-GPU glue: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" in the CPU tests). The render itself has no data-path collective (samples are independent);
the only exchange is the film reduce, the analogue of the per-worker film sum in
renderer::export_image (src/renderer/renderer.hpp:125-135).

Two partitionings:
  tiles   -- every pixel is owned by one rank for all samples (renderer.shard_tiles): the reduce
             adds exact zeros, so the N-GPU film is bit-identical to the 1-GPU film;
  samples -- every rank renders the whole frame for its own sample indices: a true sum, equal to
             the 1-GPU film up to the order of fp64 additions.
"""
import torch
import torch.distributed as dist


def sample_range(rank, world, total_spp):
    """Contiguous, disjoint sample-index ranges covering [0, total_spp)."""
    per, extra = divmod(total_spp, world)
    begin = rank * per + min(rank, extra)
    return begin, per + (1 if rank < extra else 0)


def reduce_film(film, sample_count, dst=0, all_ranks=False):
    """Sums per-rank films (tensor [H, W, 3] float64, on the device for nccl) and sample counts.

    Returns (film, total_count); the film tensor is reduced IN PLACE (valid on `dst`, or everywhere
    with all_ranks=True)."""
    count = torch.tensor([int(sample_count)], dtype=torch.int64, device=film.device)
    if all_ranks:
        dist.all_reduce(film, op=dist.ReduceOp.SUM)
        dist.all_reduce(count, op=dist.ReduceOp.SUM)
    else:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
        dist.reduce(count, dst=dst, op=dist.ReduceOp.SUM)
    return film, int(count.item())
