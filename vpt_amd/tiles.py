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

    ``dist`` is torch.distributed (backend nccl == RCCL on GPUs, gloo in the CPU tests).

    ``frames_per_gather`` = F > 1 buckets the exchange: a rank renders F consecutive frames into the F slots of a send buffer
    and ONE all_gather moves them all (every frame is still delivered to every rank, F - 1 frames later at most).  An
    asynchronous torch.distributed collective costs the host ~25 us whatever its size (tools/torch_pipeline_probe.py), more than
    a 1/8 shard's kernel takes: per-frame collectives leave the GPUs idle half of the time from N = 4 on.

    Streaming interface (what bench.py drives):  t = acquire() -> render into t -> commit();  flush() sends a partial bucket;
    last_frame() / last_sent() give the most recent frame assembled / as this rank rendered it.  The per-buffer calls
    wait(b) / gather(b) / frame(b) with send[b] / recv[b] remain for F = 1."""

    def __init__(self, dist, torch, width, height, device, rows_per_block=ROWS_PER_BLOCK, nbuf=2, always_collective=False, frames_per_gather=1, texel='rgba16f'):
        self.dist, self.torch = dist, torch
        self.always_collective = always_collective and dist.is_initialized()   # exercise RCCL even with one rank
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.W, self.H = width, height
        self.rows = local_rows(height, self.world, rows_per_block)
        self.rows_per_block = rows_per_block
        self.F = max(1, int(frames_per_gather))
        self.nbuf = nbuf
        # texel: 'rgba16f' = the render buffers; 'rgba8' = the frames as a tone mapper shows them (vpt_renderer_play_into_display): half the bytes
        dtype = {'rgba16f': torch.float16, 'rgba8': torch.uint8}[texel]
        self.texel_bytes = 8 if texel == 'rgba16f' else 4
        self._send = [torch.zeros((self.F, self.rows, width, 4), dtype=dtype, device=device) for _ in range(nbuf)]
        self._recv = [torch.zeros((self.world, self.F, self.rows, width, 4), dtype=dtype, device=device) for _ in range(nbuf)]
        # F = 1 views with the shapes of the per-buffer interface: [rows][W][4] and [world * rows][W][4]
        self.send = [t[0] for t in self._send]
        self.recv = [t.view(self.world * self.F * self.rows, width, 4) for t in self._recv]
        self.index = torch.as_tensor(gather_index(height, self.world, rows_per_block), device=device)
        self.work = [None] * nbuf
        self._next = 0                 # index of the next frame to acquire
        self._pending = 0              # committed frames of the current bucket that no collective has taken yet
        self._aliased = [False] * nbuf
        self._last_slot = None

    def shard(self):
        return (self.rank, self.world, self.rows_per_block)

    # ---- per-buffer interface --------------------------------------------------------------------
    def wait(self, k):
        """make the current stream wait for the gather that last used buffer k (before it is overwritten)"""
        if self.work[k] is not None:
            self.work[k].wait()
            self.work[k] = None

    def gather(self, k):
        """enqueue the all_gather of buffer k (all its F slots); overlaps with whatever is launched next"""
        if self.world == 1 and not self.always_collective:
            self._aliased[k] = True                         # one rank: the frame is the send buffer itself
            return
        self.work[k] = self.dist.all_gather_into_tensor(self._recv[k].view(-1), self._send[k].view(-1), async_op=True)

    def _assembled(self, k, j):
        self.wait(k)
        if self.world == 1:
            return (self._send[k][j] if self._aliased[k] else self._recv[k][0, j])[:self.H]
        return self._recv[k][:, j].reshape(self.world * self.rows, self.W, 4).index_select(0, self.index)

    def frame(self, k):
        """the assembled [H][W][4] frame of buffer k, slot 0 (waits for its gather)"""
        return self._assembled(k, 0)

    # ---- streaming interface ---------------------------------------------------------------------
    def acquire(self):
        """the [rows][W][4] tensor the next frame is rendered into (waits for the gather that last read its buffer when a new
        bucket starts)"""
        b, j = (self._next // self.F) % self.nbuf, self._next % self.F
        if j == 0:
            self.wait(b)
        return self._send[b][j]

    def acquire_bucket(self):
        """at a bucket boundary: the [F][rows][W][4] tensor the next F frames are rendered into, slot by slot (waits for the gather
        that last read it); None inside an open bucket"""
        if self._next % self.F:
            return None
        b = (self._next // self.F) % self.nbuf
        self.wait(b)
        return self._send[b]

    def commit_bucket(self):
        """the F frames of the bucket handed out by acquire_bucket() have been enqueued: send it"""
        b = (self._next // self.F) % self.nbuf
        self._next += self.F
        self._last_slot = (b, self.F - 1)
        self._pending = 0
        self.gather(b)

    def bucket_closes(self):
        """True when commit() of the frame acquired last will send the bucket"""
        return self._next % self.F == self.F - 1

    def commit(self):
        """the frame acquired last has been enqueued: a full bucket is sent"""
        b, j = (self._next // self.F) % self.nbuf, self._next % self.F
        self._next += 1
        self._pending += 1
        self._last_slot = (b, j)
        if j == self.F - 1:
            self.gather(b)
            self._pending = 0

    def flush(self):
        """send a partial bucket (its unused slots travel as they are); the next frame starts a new bucket"""
        if self._pending:
            b = ((self._next - 1) // self.F) % self.nbuf
            self.gather(b)
            self._pending = 0
            self._next = ((self._next + self.F - 1) // self.F) * self.F

    def wait_all(self):
        for k in range(self.nbuf):
            self.wait(k)

    def _last(self):
        if self._last_slot is None:
            raise RuntimeError("no frame has been committed")
        if self._pending:                                   # still in an open bucket
            self.flush()
        return self._last_slot

    def last_frame(self):
        """the most recently committed frame, assembled (flushes its bucket if it is still open, waits for the gather)"""
        b, j = self._last()
        return self._assembled(b, j)

    def last_sent(self):
        """the most recently committed frame as this rank rendered it ([rows][W][4])"""
        b, j = self._last()
        return self._send[b][j]


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
