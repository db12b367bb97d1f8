"""Image-tile sharding across the GPUs of one node and the end-of-frame gather.

The reference is single-device; this is the multi-GPU design SURVEY.md section 8(e)
derives from its per-pixel independence (src/rt_cpu/rt_cpu.rs:35-37): the
read-only scene is replicated, 8x8 tiles are dealt round-robin to ranks
(tile % world == rank, include/trx.h trx_shard), every rank traces its tiles
into a compact buffer (TRX_LAYOUT_SHARD) and ONE all-gather of 8 B/ray over
RCCL/xGMI brings a frame - or a batch of frames - together.  There is no other
collective on the path.

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


_INDEX_CACHE = {}   # (width, height, world, m, device) -> gather index shared by every FrameGather of that geometry


class FrameGather:
    """Gathers per-rank compact hit shards (int64 view of {t f32, prim u32}) into frames.

    One buffer holds up to `batch` frames of every rank, laid out [world][m][records] for a
    batch of m frames, so that ONE all-gather completes m frames: rank r traces frame f of
    the batch straight into `slot(f, m)` (its own block of the buffer) and the all-gather
    runs in place.  `records` = max_shard_tiles*64, equal on every rank.  Fewer, larger
    collectives are what RCCL over xGMI rewards (a 1080p frame is only 2 MB per rank).
    """

    def __init__(self, width, height, rank, world, device, batch=1):
        self.width, self.height, self.rank, self.world = width, height, rank, world
        self.device = torch.device(device)
        self.batch = max(1, int(batch))
        self.records = max_shard_tiles(width, height, world) * 64
        self.flat = torch.empty(world * self.batch * self.records, dtype=torch.int64, device=self.device)
        self._key = (width, height, world, str(self.device))

    @property
    def gathered(self):
        """[world, records] view of a one-frame gather."""
        return self.flat[: self.world * self.records].view(self.world, self.records)

    def slot(self, f=0, m=1):
        """Where this rank's frame f of an m-frame batch lives (trace into it, then gather(m=m))."""
        assert 0 <= f < m <= self.batch
        o = (self.rank * m + f) * self.records
        return self.flat[o:o + self.records]

    def new_local(self):
        # +inf / 0xFFFFFFFF (miss) everywhere so padding is well defined
        import numpy as np
        miss = int(np.array([0x7F800000 | (0xFFFFFFFF << 32)], dtype=np.uint64).view(np.int64)[0])
        return torch.full((self.records,), miss, dtype=torch.int64, device=self.device)

    def gather(self, local=None, async_op=False, m=1):
        """One all-gather of m frames of hit records (the only collective on the path), in place;
        a separate `local` buffer (m == 1) is copied into this rank's block first."""
        if local is not None:
            assert m == 1
            self.slot(0, 1).copy_(local)
        if self.world == 1:
            return None
        n = m * self.records
        return dist.all_gather_into_tensor(self.flat[: self.world * n], self.flat[self.rank * n:(self.rank + 1) * n],
                                           async_op=async_op)

    def prepare(self, *batch_sizes):
        """Build the gather indices for these batch sizes now (the first build synchronises with the device)."""
        for m in batch_sizes:
            if m:
                self._batch_index(m)

    def _batch_index(self, m):
        """For an m-frame gather [world][m][records]: the record to read for every pixel of every frame."""
        key = self._key + (m,)
        if key not in _INDEX_CACHE:
            idx = pixel_index_of_records(self.width, self.height, self.world, self.device).view(-1)
            src = torch.nonzero(idx >= 0).view(-1)                 # record -> pixel, valid records only
            inv = torch.empty(self.width * self.height, dtype=torch.int64, device=self.device)
            inv[idx[src]] = src                                    # pixel -> record of a 1-frame gather
            inv_rank, inv_off = inv // self.records, inv % self.records
            f = torch.arange(m, device=self.device).view(m, 1)
            _INDEX_CACHE[key] = ((inv_rank.view(1, -1) * m + f) * self.records + inv_off.view(1, -1)).reshape(-1)
        return _INDEX_CACHE[key]

    def assemble(self, out=None, m=1):
        """De-interleave gathered tile records into row-major images: out is [m * width*height] int64
        (one index_select, no host synchronisation)."""
        if out is None:
            out = torch.empty(m * self.width * self.height, dtype=torch.int64, device=self.device)
        torch.index_select(self.flat, 0, self._batch_index(m), out=out.view(-1))
        return out


def hits_to_int64(hits_np):
    """numpy structured {t,prim} -> int64 tensor sharing the 8-byte records."""
    import numpy as np
    return torch.from_numpy(np.ascontiguousarray(hits_np).view(np.int64))


def int64_to_hits(t):
    import numpy as np
    from .host import HIT_DTYPE
    return t.detach().cpu().numpy().view(HIT_DTYPE)


class AbiFrameGather(FrameGather):
    """The same batch gather through the C ABI instead of torch.distributed: trx_comm_* (RCCL loaded by libtrx.so
    itself), trx_gather_shards (the in-place ncclAllGather, on the caller's stream) and trx_assemble_frames (one
    de-interleave kernel, no index tensor).  This is what a Rust / C host binds.

    The 128-byte communicator id comes from rank 0 (`unique_id()`) and reaches the other ranks by the host's own
    means; `from_process_group` uses a torch.distributed broadcast for that and nothing else."""

    def __init__(self, width, height, rank, world, device, id_bytes, batch=1):
        super().__init__(width, height, rank, world, device, batch=batch)
        import ctypes as C
        from . import _lib as L
        self._L, self._C = L, C
        self._comm = C.c_void_p()
        self._owns = True
        self._chain = None   # streams that share one communicator: the event the previous collective recorded
        if id_bytes is None:
            return  # no communicator: the caller moves the shards itself (tests: staged through gloo), the ABI assembles
        dev = self.device.index if self.device.index is not None else 0
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(id_bytes))
        L.check(L.load().trx_comm_create(buf, rank, world, dev, C.byref(self._comm)))

    @classmethod
    def without_communicator(cls, width, height, rank, world, device, batch=1):
        """Only trx_assemble_frames (and the buffer layout) from the ABI; the shards travel by the caller's means."""
        return cls(width, height, rank, world, device, None, batch=batch)

    @classmethod
    def sharing(cls, other):
        """Another buffer on `other`'s communicator (one communicator per rank, several streams): the collectives of the
        buffers that share it are chained by an event, so they reach RCCL in one order on every rank whatever their streams."""
        fg = cls(other.width, other.height, other.rank, other.world, other.device, None, batch=other.batch)
        fg._comm, fg._owns = other._comm, False
        if other._chain is None:
            other._chain = {"event": None}
        fg._chain = other._chain
        return fg

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _lib as L
        buf = (C.c_ubyte * 128)()
        L.check(L.load().trx_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_process_group(cls, width, height, device, batch=1):
        rank, world = dist.get_rank(), dist.get_world_size()
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            ident = torch.frombuffer(bytearray(cls.unique_id()), dtype=torch.uint8).clone()
        on_gpu = dist.get_backend() == "nccl"
        t = ident.to(device) if on_gpu else ident
        dist.broadcast(t, src=0)
        return cls(width, height, rank, world, device, bytes(t.cpu().numpy().tobytes()), batch=batch)

    def world_size(self):
        if not self._comm:
            return self.world   # no communicator of its own (shards staged by the caller)
        return self._L.load().trx_comm_world_size(self._comm)

    def gather(self, local=None, async_op=False, m=1, root=None):
        """root=None: the in-place all-gather (every rank ends up with every shard); root=r: ncclSend / ncclRecv to
        rank r only (trx_gather_shards_root), after which only rank r may assemble."""
        if local is not None:
            assert m == 1
            self.slot(0, 1).copy_(local)
        cur = torch.cuda.current_stream(self.device)
        stream = cur.cuda_stream
        lib = self._L.load()
        if self._chain is not None and self._chain["event"] is not None:
            cur.wait_event(self._chain["event"])   # the previous collective on this communicator (another stream's)
        if root is None:
            self._L.check(lib.trx_gather_shards(self._comm, self._C.c_void_p(self.flat.data_ptr()), m * self.records,
                                                self._C.c_void_p(stream)))
        else:
            self._L.check(lib.trx_gather_shards_root(self._comm, self._C.c_void_p(self.flat.data_ptr()), m * self.records,
                                                     int(root), self._C.c_void_p(stream)))
        if self._chain is not None:
            self._chain["event"] = torch.cuda.Event()
            self._chain["event"].record(cur)
        return None  # enqueued on the current stream: nothing to wait for on the host

    def assemble(self, out=None, m=1):
        if out is None:
            out = torch.empty(m * self.width * self.height, dtype=torch.int64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._L.check(self._L.load().trx_assemble_frames(self._C.c_void_p(self.flat.data_ptr()), self.records, self.width,
                                                        self.height, self.world, m, self._C.c_void_p(out.data_ptr()),
                                                        self._C.c_void_p(stream)))
        return out

    def prepare(self, *batch_sizes):
        pass  # no index tensors to build

    def close(self):
        if self._comm and self._owns:
            self._L.load().trx_comm_destroy(self._comm)
        self._comm = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
