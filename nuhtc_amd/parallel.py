"""Multi-GPU layout of the WSI path: tiles are independent, so the tile list is sharded in contiguous
blocks across one process per GPU with no data-path collective; the only exchange is one variable-length gather of
per-detection records before the cross-tile merge (pattern of mmdet/apis/test.py:161-191 `collect_results_gpu`:
all_gather of the counts, then of buffers padded to the maximum) over RCCL/xGMI (backend "nccl") or gloo (CPU tests)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """One process per GPU, launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def shard_range(n, rank, world):
    """Contiguous block [lo, hi) of `n` tiles for `rank` (keeps spatial locality for the merge); sizes differ by <= 1."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_records(rec, group=None):
    """rec: (n_i, F) tensor of this rank's records (any n_i >= 0) -> list over ranks of (n_r, F) tensors (on every rank).
    Two collectives: counts (1 int64 per rank), then the payload padded to the maximum count."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [rec]
    world = dist.get_world_size(group)
    n = torch.tensor([rec.shape[0]], dtype=torch.int64, device=rec.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)
    pad = torch.zeros((m,) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
    pad[:rec.shape[0]] = rec
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return [b[:c] for b, c in zip(bufs, counts)]
