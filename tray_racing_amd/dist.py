"""Image-tile sharding across the GPUs of one node and the end-of-frame gather.

The reference is single-device; this is the multi-GPU design SURVEY.md section 8(e)
derives from its per-pixel independence (src/rt_cpu/rt_cpu.rs:35-37): the
read-only scene is replicated, 8x8 tiles are dealt round-robin to ranks
(tile % world == rank, include/trx.h trx_shard), every rank traces its tiles
into a compact buffer (TRX_LAYOUT_SHARD) and ONE all-gather of 8 B/ray over
RCCL/xGMI brings the frame together.  There is no other collective on the path.

Everything here is tensor plumbing (torch.distributed); it works on CPU
tensors with the gloo backend, which is how tests/ covers world_size > 1.
"""
import torch
import torch.distributed as dist


def shard_tiles(width, height, rank, world):
    """Number of 8x8 tiles owned by `rank` (same rule as trx_shard_tiles)."""
    tiles = ((width + 7) // 8) * ((height + 7) // 8)
    if rank >= world or tiles <= rank:
        return 0
    return (tiles - rank + world - 1) // world


def max_shard_tiles(width, height, world):
    return shard_tiles(width, height, 0, world)


def pixel_index_of_records(width, height, world, device="cpu"):
    """For the gathered [world, L*64] record array: the full-image pixel id of
    every record, or -1 for padding / pixels outside the image.  L = tiles of rank 0."""
    tx = (width + 7) // 8
    ty = (height + 7) // 8
    L = max_shard_tiles(width, height, world)
    r = torch.arange(world, device=device).view(world, 1, 1)
    lt = torch.arange(L, device=device).view(1, L, 1)
    k = torch.arange(64, device=device).view(1, 1, 64)
    tile = lt * world + r
    px = (tile % tx) * 8 + (k & 7)
    py = (tile // tx) * 8 + (k >> 3)
    valid = (tile < tx * ty) & (px < width) & (py < height)
    pix = py * width + px
    return torch.where(valid, pix, torch.full_like(pix, -1)).view(world, L * 64)


class FrameGather:
    """Gathers per-rank compact hit shards (int64 view of {t f32, prim u32}) into a frame.

    `local` buffers hold max_shard_tiles*64 records (8 B each, viewed as int64) so
    every rank contributes an equal-sized block to all_gather_into_tensor.
    """

    def __init__(self, width, height, rank, world, device):
        self.width, self.height, self.rank, self.world = width, height, rank, world
        self.device = torch.device(device)
        self.records = max_shard_tiles(width, height, world) * 64
        self.gathered = torch.empty((world, self.records), dtype=torch.int64, device=self.device)
        idx = pixel_index_of_records(width, height, world, self.device)
        self._valid = (idx >= 0).view(-1)
        self._pix = idx.view(-1)[self._valid]

    def new_local(self):
        # +inf / 0xFFFFFFFF (miss) everywhere so padding is well defined
        import numpy as np
        miss = int(np.array([0x7F800000 | (0xFFFFFFFF << 32)], dtype=np.uint64).view(np.int64)[0])
        return torch.full((self.records,), miss, dtype=torch.int64, device=self.device)

    def gather(self, local, async_op=False):
        """One all-gather of the frame's hit records (the only collective on the path)."""
        if self.world == 1:
            self.gathered[0].copy_(local)
            return None
        return dist.all_gather_into_tensor(self.gathered.view(-1), local, async_op=async_op)

    def assemble(self, out=None):
        """De-interleave gathered tile records into the row-major image (int64 per pixel)."""
        if out is None:
            out = torch.empty(self.width * self.height, dtype=torch.int64, device=self.device)
        out[self._pix] = self.gathered.view(-1)[self._valid]
        return out


def hits_to_int64(hits_np):
    """numpy structured {t,prim} -> int64 tensor sharing the 8-byte records."""
    import numpy as np
    return torch.from_numpy(np.ascontiguousarray(hits_np).view(np.int64))


def int64_to_hits(t):
    import numpy as np
    from .host import HIT_DTYPE
    return t.detach().cpu().numpy().view(HIT_DTYPE)
