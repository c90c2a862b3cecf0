"""Frame sharding across the GPUs of one node: frames are independent through the whole path, so
each rank owns a contiguous block of frames, weights are replicated, and the ONLY exchange is one all_gather
of the per-frame result records (keypoints + camera records, < 1 KB per frame) at the end of a pass.
`backend='nccl'` is RCCL over xGMI on ROCm; the same code runs on gloo for CPU tests."""
import torch
import torch.distributed as dist


def shard_range(n_frames: int, rank: int, world: int):
    """Contiguous, balanced block [start, stop) of `n_frames` for `rank` (first n % world ranks get one more)."""
    base, rem = divmod(n_frames, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def pack_records(kpts: torch.Tensor, *records: torch.Tensor) -> torch.Tensor:
    """(B,57,3) fp32 keypoints + any number of (B,n) uint8 record tensors -> (B, bytes) uint8 rows."""
    B = kpts.shape[0]
    parts = [kpts.contiguous().view(torch.uint8).reshape(B, -1)] + [r.reshape(B, -1) for r in records]
    return torch.cat(parts, dim=1).contiguous()


def gather_records(local: torch.Tensor, counts=None) -> torch.Tensor:
    """One collective: every rank receives the records of all frames, in frame order.
    `local` (b_r, bytes) uint8; ranks may hold different frame counts (`counts` = list per rank)."""
    if not dist.is_initialized():
        return local
    world = dist.get_world_size()
    if counts is None:
        counts = [local.shape[0]] * world
    mx = max(counts)
    if local.shape[0] < mx:                       # pad to a common size so that one all_gather suffices
        pad = torch.zeros((mx - local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    out = torch.empty((world * mx, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    if all(c == mx for c in counts):
        return out
    return torch.cat([out[r * mx: r * mx + counts[r]] for r in range(world)], dim=0)
