"""Frame sharding across the GPUs of one node: frames are independent through the whole path, so
each rank owns a contiguous block of frames, weights are replicated, and the ONLY exchange is ONE all_gather
of the per-frame result records (keypoints + camera records, < 1 KB per frame) after the rank's LAST batch
(`RecordLog`: the records of every batch stay on the rank until then; north_star: "a single RCCL gather over xGMI at the end").
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


class RecordLog:
    """The per-rank side of the path's single collective: packed records of every batch a rank has processed stay on the rank
    (`add`), and `gather` issues ONE all_gather_into_tensor for all of them after the last batch.  No collective, hence no RCCL
    stream, is active while batches are in flight: a per-batch gather sat on the solve stream behind the solve (every step's
    collective waited for the slowest rank's slowest Levenberg-Marquardt fit) and took one of the four hardware queues the
    network and the solve streams need (pipeline.py).  Ranks may hold different frame counts (ragged shards: `counts`)."""

    def __init__(self):
        self.parts = []

    def add(self, packed: torch.Tensor):
        self.parts.append(packed)

    def __len__(self):
        return sum(p.shape[0] for p in self.parts)

    def local(self) -> torch.Tensor:
        """This rank's records in submission order, (frames of the rank, bytes) uint8."""
        if not self.parts:
            raise ValueError('RecordLog.local(): no batch was logged')
        if len(self.parts) > 1:
            self.parts = [torch.cat(self.parts, dim=0)]
        return self.parts[0]

    def gather(self, counts=None) -> torch.Tensor:
        """The one collective: (all frames of all ranks in frame order, bytes).  `counts` = frames per rank when they differ.
        The caller orders the current stream behind the producers of the logged tensors first (CalibrationPipeline.gather_all)."""
        out = gather_records(self.local(), counts)
        self.parts = []
        return out
