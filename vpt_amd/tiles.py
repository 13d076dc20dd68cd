"""Image-plane sharding across the GPUs of one node + the per-frame gather (no reference counterpart).

Rank g of G owns the row blocks b (rows_per_block rows each) with b % G == g.  Blocks are interleaved so that
cube-hitting and cube-missing rows balance; per-pixel seeds depend on GLOBAL pixel coordinates only, so the
gathered image is bit-identical to a single-GPU render.  Every rank pads its local buffer to the same number of
rows, so the frame gather is ONE equal-sized all_gather over RCCL (xGMI is fully connected: each peer's slice
arrives on its own link); rows are put back in order with one index_select.
"""
import numpy as np

ROWS_PER_BLOCK = 8


def local_rows(height, world, rows_per_block=ROWS_PER_BLOCK):
    """rows in every rank's (padded) local buffer — must match vpt_renderer_local_rows"""
    if world == 1:
        return height
    nblocks = (height + rows_per_block - 1) // rows_per_block
    return ((nblocks + world - 1) // world) * rows_per_block


def row_owner(height, world, rows_per_block=ROWS_PER_BLOCK):
    """for every global row j: (rank, local_row) holding it"""
    j = np.arange(height)
    if world == 1:
        return np.zeros(height, dtype=np.int64), j.astype(np.int64)
    b = j // rows_per_block
    return (b % world).astype(np.int64), ((b // world) * rows_per_block + j % rows_per_block).astype(np.int64)


def gather_index(height, world, rows_per_block=ROWS_PER_BLOCK):
    """flat row index into the gathered [world * local_rows] buffer for every global row"""
    rank, lrow = row_owner(height, world, rows_per_block)
    return rank * local_rows(height, world, rows_per_block) + lrow


class FrameGather:
    """all_gather of the per-rank RGBA16F render buffers and reassembly into the [H][W][4] frame.

    ``dist`` is torch.distributed (backend nccl == RCCL on GPUs, gloo in the CPU tests)."""

    def __init__(self, dist, torch, width, height, device, rows_per_block=ROWS_PER_BLOCK, nbuf=2, always_collective=False):
        self.dist, self.torch = dist, torch
        self.always_collective = always_collective and dist.is_initialized()   # exercise RCCL even with one rank
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.W, self.H = width, height
        self.rows = local_rows(height, self.world, rows_per_block)
        self.rows_per_block = rows_per_block
        self.send = [torch.zeros((self.rows, width, 4), dtype=torch.float16, device=device) for _ in range(nbuf)]
        self.recv = [torch.zeros((self.world * self.rows, width, 4), dtype=torch.float16, device=device) for _ in range(nbuf)]
        self.index = torch.as_tensor(gather_index(height, self.world, rows_per_block), device=device)
        self.work = [None] * nbuf

    def shard(self):
        return (self.rank, self.world, self.rows_per_block)

    def wait(self, k):
        """make the current stream wait for the gather that last used buffer k (before it is overwritten)"""
        if self.work[k] is not None:
            self.work[k].wait()
            self.work[k] = None

    def gather(self, k):
        """enqueue the all_gather of send[k] into recv[k]; overlaps with whatever is launched next"""
        if self.world == 1 and not self.always_collective:
            self.recv[k] = self.send[k]
            return
        self.work[k] = self.dist.all_gather_into_tensor(self.recv[k].view(-1), self.send[k].view(-1), async_op=True)

    def frame(self, k):
        """the assembled [H][W][4] frame of buffer k (waits for its gather)"""
        self.wait(k)
        if self.world == 1:
            return self.recv[k][:self.H]
        return self.recv[k].index_select(0, self.index)


class RcclFrameGather:
    """The same pipeline below the C ABI (vpt_gather_*): the library owns the send / receive buffers, an RCCL
    communicator and a communication stream; one call per frame enqueues kernel + all_gather.  ``id_bytes`` is the
    128-byte RCCL id obtained by ONE rank (``RcclFrameGather.unique_id()``) and shared with the others, e.g. through
    ``torch.distributed.broadcast_object_list``."""

    def __init__(self, renderer, id_bytes, rank, world, root=-1):
        """root = -1: every rank receives each frame (all_gather); root = k: only rank k does (grouped send/recv)"""
        import ctypes as C
        from . import _native as N
        self._N, self._C = N, C
        self.renderer = renderer
        self.rank, self.world = rank, world
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(id_bytes), 128)
        N.check(N.lib().vpt_gather_create(renderer._h, buf, rank, world, C.byref(h)))
        self._h = h
        self.root = -1
        if root != -1:
            self.set_root(root)

    def set_root(self, root):
        self._N.check(self._N.lib().vpt_gather_set_root(self._h, int(root)))
        self.root = int(root)

    def receives(self):
        return self.root < 0 or self.root == self.rank

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _native as N
        buf = C.create_string_buffer(128)
        N.check(N.lib().vpt_gather_unique_id(buf))
        return bytes(buf.raw)

    def render(self):
        """one AbstractRenderer.render() (fused pass) into the send buffer + asynchronous all_gather"""
        r = self.renderer
        r._bind_volume()
        u = r._prepare_frame_uniforms()
        self._N.check(self._N.lib().vpt_gather_render(self._h, self._C.byref(u)))

    def play(self, count, fused=False):
        """`count` frames (kernel + gather each) by one native call; fused (MCM): count passes in one launch, one gather"""
        r = self.renderer
        r._bind_volume()
        u, vars_ = r._collect_frames(count)
        self._N.check(self._N.lib().vpt_gather_play(self._h, self._C.byref(u), vars_.ctypes.data_as(self._C.c_void_p),
                                                    count, self._N.PLAY_FUSED if fused else self._N.PLAY_EAGER))

    def synchronize(self):
        self._N.check(self._N.lib().vpt_gather_synchronize(self._h))

    def frame(self):
        w, h = self.renderer._size()
        out = np.empty((h, w, 4), dtype=np.float16)
        self._N.check(self._N.lib().vpt_gather_read_frame(self._h, out.ctypes.data_as(self._C.c_void_p), out.nbytes))
        return out

    def destroy(self):
        if self._h:
            self._N.lib().vpt_gather_destroy(self._h)
            self._h = None
