"""Multi-GPU plumbing: one process per GPU, samples split by range, one sum-reduce.

Every rank holds the whole (small) scene and renders ALL pixels for its own slice of the
per-pixel sample grid into its own float accumulation buffer; the buffers are summed onto
rank 0 with a single torch.distributed reduce (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests), after which rank 0 resolves the image.  The reference has no
counterpart: it is a single process (framework/renderer.cpp:39-53 only fans out threads).
"""


def sample_range(rank, world_size, spp):
    """Contiguous, balanced slice [begin, end) of the sample indices s = aax*aa_steps + aay."""
    base, extra = divmod(spp, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def reduce_accum(accum, dst=0):
    """Sum the per-rank accumulation buffers onto `dst` (in place there)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum
