"""Reader / writer of the reference's `.lines` / `.scene` / `.tmpl` files.

Format (modules/core/include/openfdcm/core/serialization.h:42-57,59-97,137-139 + the packio v0.2.1
container, restated from the shipped assets):

    0   16  signature "OPENFDCM" zero padded
    16   6  3 x u16 container version (0, 2, 0)
    22   1  u8 compression flag (1 = zlib)
    23   8  u64 uncompressed body length
    31   8  u64 compressed body length
    39   .  zlib stream of: 45-byte packed LinesSerialHeader, then N records of 4 float32 x1,y1,x2,y2

File I/O is host work and stays in Python (not part of the accelerated path).
"""
import os
import struct
import time
import zlib

import numpy as np

_SIGNATURE = b"OPENFDCM" + b"\x00" * 8
_CONTAINER_VERSION = (0, 2, 0)
_HEADER = struct.Struct("<HIHH8sHHHHHHIBHQ")  # LinesSerialHeader, packed (45 bytes)
assert _HEADER.size == 45
_VERSION = (0, 10, 0)


def read(filepath):
    """openfdcm.read (core.cpp:41): returns a (4, N) float32 array."""
    if not os.path.exists(filepath):
        raise RuntimeError(f"File '{filepath}' does not exist")
    with open(filepath, "rb") as f:
        blob = f.read()
    if len(blob) < 39 or blob[:16] != _SIGNATURE:
        raise RuntimeError(f"File '{filepath}' is not an OPENFDCM line file")
    compressed = blob[22]
    ulen, clen = struct.unpack("<QQ", blob[23:39])
    body = blob[39:39 + clen]
    if compressed:
        body = zlib.decompress(body)
    if len(body) != ulen:
        raise RuntimeError(f"File '{filepath}' is truncated")
    hdr = _HEADER.unpack(body[:45])
    line_format, record_len, n = hdr[12], hdr[13], hdr[14]
    offset = hdr[11]
    if line_format != 0:
        raise RuntimeError(f"Line data format not recognized, found <{record_len}>")
    data = np.frombuffer(body, dtype="<f4", count=4 * n, offset=offset)
    return np.ascontiguousarray(data.reshape(n, 4).T)


def write(filepath, linearray):
    """openfdcm.write (core.cpp:42)."""
    a = np.asarray(linearray, dtype=np.float32)
    if a.ndim != 2 or a.shape[0] != 4:
        raise ValueError(f"expected a (4, N) line array, got shape {a.shape}")
    if os.path.exists(filepath):
        try:
            os.remove(filepath)
        except OSError:
            raise RuntimeError(f"File '{filepath}' can't be overwritten")
    tm = time.gmtime()
    hdr = _HEADER.pack(0, 0, 0, 0, b"\x00" * 8, _VERSION[0], _VERSION[1], _VERSION[2], tm.tm_yday - 1,
                       tm.tm_year - 1900, 45, 45, 0, 16, a.shape[1])
    body = hdr + np.ascontiguousarray(a.T, dtype="<f4").tobytes()
    comp = zlib.compress(body)
    try:
        with open(filepath, "wb") as f:
            f.write(_SIGNATURE)
            f.write(struct.pack("<3H", *_CONTAINER_VERSION))
            f.write(bytes([1]))
            f.write(struct.pack("<QQ", len(body), len(comp)))
            f.write(comp)
    except OSError:
        raise RuntimeError(f"Cannot write file '{filepath}'")
