"""Data-parallel plumbing shared by the trainers: one process per GPU, started by
`python -m torch.distributed.run --nproc-per-node N -m gan_amd.main ...` (RCCL over xGMI).

The reference trains on one device.  Here every rank holds a full replica and takes every N-th batch of
the loop the reference runs (global batch = N x batch_size); gradients are summed over the ranks per
optimiser update (optim.AdamOptimizer.all_reduce) so the replicas stay bit-identical, and rank 0 alone
prints, samples and writes checkpoints."""
import os

import torch


def init_from_env(backend="nccl"):
    """Join the job torch.distributed.run started (WORLD_SIZE / RANK / LOCAL_RANK / MASTER_* in the
    environment) and bind this rank's GPU.  Returns (world_size, rank); (1, 0) outside such a job."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 1, 0
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC for RCCL's peer mappings
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0    # gloo: rehearsal on one GPU
    torch.cuda.set_device(local)
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return world, dist.get_rank()


def rank_of(world_size, process_group=None):
    if world_size <= 1:
        return 0
    import torch.distributed as dist
    return dist.get_rank(process_group)


def shard_start(step, rank, world_size, batch_size):
    """First item of this rank's batch in global step `step`: the ranks take consecutive batches of the
    sequence the single-device loop would walk through."""
    return (step * world_size + rank) * batch_size


def rank_seed(seed, rank):
    """Seed of a rank's own random draws (z, crops, flips, wrong images); rank 0 keeps `seed`."""
    return seed if rank == 0 or seed is None else seed + 7919 * rank
