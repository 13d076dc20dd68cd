"""Loaders — src/js/loaders/*.js: byte-range sources the readers pull from.

The reference's interface is two methods (AbstractLoader.js:3-11): ``readLength()`` and ``readData(start, end)``
(end exclusive, as ``Blob.slice`` / the HTTP Range header AjaxLoader.js:20-26 builds).  ``BlobLoader`` wraps bytes
already in memory (BlobLoader.js:3-21; here any bytes-like object or numpy uint8 array, returned as zero-copy views);
``FileLoader`` is the server-side stand-in for AjaxLoader's ranged fetches: a file on disk read with ``os.pread``, so a
multi-gigabyte volume is never resident on the host as a whole.  Networking itself (AjaxLoader) is out of scope.
"""
import os

import numpy as np


class AbstractLoader:
    def readLength(self):
        raise NotImplementedError

    def readData(self, start, end):
        raise NotImplementedError


class BlobLoader(AbstractLoader):
    """src/js/loaders/BlobLoader.js:3-21"""

    def __init__(self, blob):
        if isinstance(blob, np.ndarray):
            if blob.dtype != np.uint8:
                raise TypeError('BlobLoader expects uint8 data')
            self.blob = np.ascontiguousarray(blob).reshape(-1)
        else:
            self.blob = np.frombuffer(blob, dtype=np.uint8)

    def readLength(self):
        return int(self.blob.size)

    def readData(self, start, end):
        return self.blob[max(int(start), 0):max(int(end), 0)]          # Blob.slice clamps to the blob's extent


class FileLoader(AbstractLoader):
    """ranged reads of a local file (extension; the role AjaxLoader.js:3-30 plays against an HTTP server)"""

    def __init__(self, path):
        self.url = path
        self._fd = os.open(path, os.O_RDONLY)
        self._length = os.fstat(self._fd).st_size

    def close(self):
        if self._fd is not None:
            os.close(self._fd)
            self._fd = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def readLength(self):
        return int(self._length)

    def readData(self, start, end):
        start = min(max(int(start), 0), self._length); end = min(max(int(end), start), self._length)
        out = np.empty(end - start, dtype=np.uint8)
        view, done = memoryview(out), 0
        while done < end - start:
            n = os.preadv(self._fd, [view[done:]], start + done)
            if n <= 0:
                raise IOError('short read from %s' % self.url)
            done += n
        return out


def LoaderFactory(which):
    """src/js/loaders/LoaderFactory.js:4-12 ('ajax' is networking: not built; 'file' is the local stand-in)"""
    table = {'blob': BlobLoader, 'file': FileLoader}
    if which not in table:
        raise RuntimeError('No suitable class')
    return table[which]
