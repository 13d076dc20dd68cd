"""Device context: the object passed where the reference passes its WebGL2RenderingContext (`gl`)."""
import ctypes as C

from . import _native as N


class Context:
    """One HIP device + stream (vpt_context).  RenderingContext.js:66-106 creates the GL context the
    reference's Volume / renderers receive as ``gl``; this is its replacement."""

    def __init__(self, device=0, stream=None):
        """stream: optional raw hipStream_t handle (int) owned by the caller, e.g.
        torch.cuda.current_stream().cuda_stream; default = a private non-blocking stream."""
        L = N.lib()
        h = C.c_void_p()
        if stream is None:
            N.check(L.vpt_context_create(int(device), C.byref(h)))
        else:
            N.check(L.vpt_context_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(h)))
        self._h = h
        self.device = int(device)

    @staticmethod
    def device_count():
        n = C.c_int(0)
        N.check(N.lib().vpt_device_count(C.byref(n)))
        return n.value

    def synchronize(self):
        N.check(N.lib().vpt_context_synchronize(self._h))

    def destroy(self):
        if self._h:
            N.lib().vpt_context_destroy(self._h)
            self._h = None

    def probe_math(self, which, values):
        import numpy as np
        values = np.ascontiguousarray(values, dtype=np.float32)
        n = values.size // 2 if which in (N.PROBE_ATAN2, N.PROBE_MIN, N.PROBE_MAX, N.PROBE_POW) else values.size
        out = np.empty(n, dtype=np.float32)
        N.check(N.lib().vpt_probe_math(self._h, which, values.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), n))
        return out

    def stream_read_rate(self, nbytes=4 << 30, iterations=10):
        """measured HBM streaming-read rate in GB/s (vpt_probe_stream_read)"""
        g = C.c_double(0)
        N.check(N.lib().vpt_probe_stream_read(self._h, int(nbytes), int(iterations), C.byref(g)))
        return g.value
