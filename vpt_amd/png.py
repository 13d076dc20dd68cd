"""Minimal PNG encoder (zlib only) for the animation recorder and the examples: 8-bit RGBA, no filtering."""
import struct
import zlib

import numpy as np


def encode_png(rgba, bottom_up=True):
    """rgba: [h][w][4] uint8.  bottom_up: row 0 is the bottom row (the GL convention of the render buffers)."""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w, c = rgba.shape
    assert c == 4
    rows = rgba[::-1] if bottom_up else rgba
    raw = b"".join(b"\x00" + rows[j].tobytes() for j in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def decode_png(data):
    """inverse of encode_png for its own output (filter type 0 only): -> [h][w][4] uint8, top row first"""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, w, h, idat = 8, 0, 0, b""
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h = struct.unpack(">II", body[:8])
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 4 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 4).copy()


def write_png(path, rgba, bottom_up=True):
    with open(path, "wb") as f:
        f.write(encode_png(rgba, bottom_up))
