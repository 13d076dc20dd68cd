"""Readers — src/js/readers/*.js: RAW, ZIP (stored entries) and BVP volume containers over a loader.

Checked against the reference's own readers: tests/golden/readers_r01.json holds what RAWReader.js, ZIPReader.js and
BVPReader.js returned for a synthetic archive when run under node in the build container
(tests/golden/make_reader_fixture.py).  The reference's methods are ``async``; here they are plain calls.
"""
import json
import struct

import numpy as np

from .loaders import AbstractLoader, BlobLoader

# WebGL2 enums carried by reader metadata (RAWReader.js:36-38)
GL_RED, GL_R8, GL_UNSIGNED_BYTE = 6403, 33321, 5121
GL_RG, GL_RG8 = 33319, 33323              # two-channel volumes (value + e.g. gradient magnitude) of BVP manifests
GL_RGB, GL_RGB8, GL_RGBA, GL_RGBA8 = 6407, 32849, 6408, 32856      # byte manifests with more channels: texture(uVolume, p).rg reads the first two
GL_FLOAT, GL_HALF_FLOAT, GL_R32F, GL_R16F = 5126, 5131, 33326, 33325   # float volumes (Volume.js:84-105 maps FLOAT -> Float32Array, HALF_FLOAT -> Uint16Array)


class AbstractReader:
    """src/js/readers/AbstractReader.js:1-15"""

    def __init__(self, loader):
        self._loader = loader

    def readMetadata(self):
        raise NotImplementedError

    def readBlock(self, block):
        raise NotImplementedError


class RAWReader(AbstractReader):
    """src/js/readers/RAWReader.js:3-70: a raw u8 volume exposed as one placement per z slice.
    ``loader``: an AbstractLoader, or (extension) bytes / a uint8 array, wrapped in a BlobLoader."""

    def __init__(self, loader, options=None):
        super().__init__(loader if isinstance(loader, AbstractLoader) else BlobLoader(loader))
        options = options or {}
        self.width = options.get('width', 0)                      # :8-12
        self.height = options.get('height', 0)
        self.depth = options.get('depth', 0)

    def readMetadata(self):                                        # :15-63
        modality = {
            'name': 'default',
            'dimensions': {'width': self.width, 'height': self.height, 'depth': self.depth},
            'transform': {'matrix': [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]},
            'format': GL_RED, 'internalFormat': GL_R8, 'type': GL_UNSIGNED_BYTE,
            'placements': [],
        }
        blocks = []
        for i in range(self.depth):
            modality['placements'].append({'index': i, 'position': {'x': 0, 'y': 0, 'z': i}})
            blocks.append({'url': 'default', 'format': 'raw',
                           'dimensions': {'width': self.width, 'height': self.height, 'depth': 1}})
        return {'meta': {'version': 1}, 'modalities': [modality], 'blocks': blocks}

    def readBlock(self, block):                                    # :65-70
        slice_bytes = self.width * self.height
        return self._loader.readData(block * slice_bytes, (block + 1) * slice_bytes)


class ZIPReader(AbstractReader):
    """src/js/readers/ZIPReader.js:3-100: end-of-central-directory record in the last 22 bytes (no archive comment),
    central directory walk, entries returned as their raw stored bytes (the reference never inflates: method 0 only)."""

    def __init__(self, loader):
        super().__init__(loader)
        self._eocd = None
        self._cd = None

    def getFiles(self):                                            # :13-19
        if self._cd is None:
            self._readCD()
        return [e['name'] for e in self._cd]

    def readFile(self, fileName):                                  # :21-40
        if self._cd is None:
            self._readCD()
        entry = next((e for e in self._cd if e['name'] == fileName), None)
        if entry is None:
            raise RuntimeError('ZIPReader: file %s not in CD' % fileName)
        header_start = entry['headerOffset']
        header_end = header_start + 30
        header = bytes(self._loader.readData(header_start, header_end))
        name_len, extra_len = struct.unpack_from('<HH', header, 26)
        data_start = header_end + name_len + extra_len
        return self._loader.readData(data_start, data_start + entry['compressedSize'])

    def _readEOCD(self):                                           # :42-58
        MIN_EOCD_SIZE = 22
        length = self._loader.readLength()
        offset = max(length - MIN_EOCD_SIZE, 0)
        size = min(length, MIN_EOCD_SIZE)
        data = bytes(self._loader.readData(offset, offset + size))
        entries, = struct.unpack_from('<H', data, 10)
        cd_size, cd_offset = struct.unpack_from('<II', data, 12)
        self._eocd = {'entries': entries, 'size': cd_size, 'offset': cd_offset}

    def _readCD(self):                                             # :60-93
        if self._eocd is None:
            self._readEOCD()
        start = self._eocd['offset']
        data = bytes(self._loader.readData(start, start + self._eocd['size']))
        offset, entries = 0, []
        for _ in range(self._eocd['entries']):
            gpflag, method = struct.unpack_from('<HH', data, offset + 8)
            compressed, uncompressed = struct.unpack_from('<II', data, offset + 20)
            name_len, extra_len, comment_len = struct.unpack_from('<HHH', data, offset + 28)
            header_offset, = struct.unpack_from('<I', data, offset + 42)
            name = data[offset + 46:offset + 46 + name_len].decode('utf-8')
            entries.append({'gpflag': gpflag, 'method': method, 'compressedSize': compressed, 'uncompressedSize': uncompressed,
                            'name': name, 'headerOffset': header_offset})
            offset += 46 + name_len + extra_len + comment_len
        self._cd = entries


class BVPReader(AbstractReader):
    """src/js/readers/BVPReader.js:4-34: a ZIP whose manifest.json is the metadata object and whose blocks are entries"""

    def __init__(self, loader):
        super().__init__(loader)
        self._metadata = None
        self._zipReader = ZIPReader(self._loader)

    def readMetadata(self):                                        # :13-20
        data = self._zipReader.readFile('manifest.json')
        self._metadata = json.loads(bytes(data).decode('utf-8'))
        return self._metadata

    def readBlock(self, block):                                    # :22-29
        if not self._metadata:
            self.readMetadata()
        return self._zipReader.readFile(self._metadata['blocks'][block]['url'])


def ReaderFactory(which):
    """src/js/readers/ReaderFactory.js:5-14"""
    table = {'bvp': BVPReader, 'raw': RAWReader, 'zip': ZIPReader}
    if which not in table:
        raise RuntimeError('No suitable class')
    return table[which]
