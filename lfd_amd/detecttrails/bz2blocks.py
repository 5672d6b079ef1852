"""A ``.fits.bz2`` frame decoded on several cores at once.

The reference decompresses a missing frame's ``.bz2`` twin with the ``bunzip2`` program into ``$FITS_DUMP`` and reads the
result back (detecttrails.py:81-109); one frame is ~14 independent bzip2 blocks of 900 kB, decoded one after the other: ~0.4 s
on one core.  The blocks of a bzip2 stream can be decoded independently once they are found: every block starts with the
48-bit magic 0x314159265359 at an arbitrary BIT offset and the stream ends with 0x177245385090 + the combined CRC.  Here the
block starts are located (``bytes.find`` for the five bytes a magic covers completely at each of the eight bit shifts), every block is re-wrapped as a
one-block stream ("BZh<level>" + block + end-of-stream magic + the block's own CRC, which is the combined CRC of a one-block
stream) and handed to libbz2 on a thread pool (``bz2.decompress`` releases the interpreter lock).  libbz2 checks every block's
CRC, so a false magic inside compressed data (probability ~1e-7 per file) shows up as an error and the file is decoded the
ordinary way; so is anything that is not a single plain stream.  Nothing here touches the GPU.
"""
import bz2

import numpy as np

_BLOCK = bytes.fromhex("314159265359")
_EOS = bytes.fromhex("177245385090")


def _shifted(a, k, lo=0, hi=None):
    """Bytes of the bit stream of ``a`` (uint8 array) starting k bits into byte ``lo`` (k in 0..7), up to byte ``hi``."""
    hi = len(a) if hi is None else min(hi, len(a))
    if k == 0:
        return a[lo:hi].tobytes()
    cur = a[lo:hi]
    nxt = np.empty_like(cur)
    nxt[:-1] = cur[1:]
    nxt[-1] = a[hi] if hi < len(a) else 0
    return ((cur << k) | (nxt >> (8 - k))).tobytes()        # (uint8 arithmetic: the shifted-out bits fall off)


def _find_all(data, pattern):
    """Bit offsets of every occurrence of the 48-bit pattern in the bit stream of ``data`` (bytes), ascending.  A pattern that
    starts k bits into a byte covers the next five bytes completely: those are searched for with ``bytes.find`` on the file as
    it is (no shifted copies), the two partial bytes around them are compared afterwards."""
    P = int.from_bytes(pattern, "big")
    out = []
    n = len(data)
    for k in range(8):
        if k == 0:
            p = data.find(pattern)
            while p >= 0:
                out.append(8 * p)
                p = data.find(pattern, p + 1)
            continue
        mid = ((P >> k) & 0xFFFFFFFFFF).to_bytes(5, "big")
        head, tail = P >> (40 + k), P & ((1 << k) - 1)       # the pattern's first 8 - k bits, its last k bits
        q = data.find(mid)
        while q >= 0:
            p = q - 1
            if p >= 0 and q + 5 < n and (data[p] & ((1 << (8 - k)) - 1)) == head and (data[q + 5] >> (8 - k)) == tail:
                out.append(8 * p + k)
            q = data.find(mid, q + 1)
    return sorted(out)


def _find_magics(data, a):
    """(block starts, end-of-stream marks) as bit offsets: the library's native scan (no interpreter lock held), or the
    pure-Python search where the library cannot be loaded (it is host code: no GPU needed)."""
    try:
        from .. import _native
        lib = _native.lib()
    except (ImportError, OSError):
        return _find_all(data, _BLOCK), _find_all(data, _EOS)
    cap = 4096
    out = np.zeros(cap, np.uint64)
    n = lib.lfdmi_bz2_find_blocks(a.ctypes.data, len(data), out.ctypes.data, cap)
    if n < 0 or n > cap:
        return _find_all(data, _BLOCK), _find_all(data, _EOS)
    v = out[:n]
    return [int(x >> 1) for x in v if not (int(x) & 1)], [int(x >> 1) for x in v if int(x) & 1]


def split_blocks(data):
    """[(one-block bzip2 stream), ...] of a single-stream bzip2 file, or None when the file is not one plain stream."""
    if len(data) < 14 or data[:3] != b"BZh" or not (0x31 <= data[3] <= 0x39):
        return None
    a = np.frombuffer(data, np.uint8)
    starts, ends = _find_magics(data, a)
    if not starts or starts[0] != 32 or len(ends) != 1 or ends[0] < starts[-1]:   # (several end marks: concatenated streams)
        return None
    eos = ends[0]
    if (eos + 80 + 7) // 8 != len(data):                  # something follows the stream (a second stream, padding): not handled here
        return None
    bounds = starts + [eos]
    head = data[:4]
    streams = []
    for b0, b1 in zip(bounds[:-1], bounds[1:]):
        nbits = b1 - b0
        if nbits < 80:
            return None
        k, lo = b0 & 7, b0 >> 3
        body = _shifted(a, k, lo, lo + (nbits + 7) // 8)  # the block, byte aligned; its last byte may hold bits of the next block
        full, r = divmod(nbits, 8)
        crc = body[6:10]                                  # the block's CRC follows its magic
        tail = int.from_bytes(_EOS + crc, "big")          # 80 bits
        if r:
            keep = body[full] >> (8 - r)                  # the block's last r bits
            val = (keep << 80) | tail
            nb = r + 80
            val <<= (-nb) % 8
            tail_bytes = val.to_bytes((nb + 7) // 8, "big")
        else:
            tail_bytes = tail.to_bytes(10, "big")
        streams.append(head + body[:full] + tail_bytes)
    return streams


def decompress(data, pool=None, min_blocks=2):
    """``bz2.decompress(data)``; with a thread pool the blocks of the stream are decoded side by side."""
    if pool is None:
        return bz2.decompress(data)
    streams = split_blocks(data)
    if not streams or len(streams) < min_blocks:
        return bz2.decompress(data)
    try:
        parts = list(pool.map(bz2.decompress, streams))
    except (OSError, ValueError, EOFError):               # a false block magic inside compressed data: libbz2's CRC check caught it
        return bz2.decompress(data)
    return b"".join(parts)


_POOL = None


def shared_pool():
    """A process-wide pool for callers without one of their own (the frame-at-a-time path: one .fits.bz2 at a time, its blocks
    on up to 16 of the cores this process may run on)."""
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        from .. import usable_cores
        cores = usable_cores()
        _POOL = ThreadPoolExecutor(max(2, min(16, cores)), thread_name_prefix="lfd-bz2")
    return _POOL
