"""lfd_amd.detecttrails.bz2blocks: a bzip2 stream split at its block boundaries and decoded block by block equals
bz2.decompress on the whole stream (the drop-in's .fits.bz2 ingest, detecttrails.py:81-109)."""
import bz2
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from lfd_amd.detecttrails import bz2blocks


def test_blocks_decode_to_the_same_bytes():
    rng = np.random.default_rng(5)
    with ThreadPoolExecutor(4) as pool:
        for level, nbytes in ((1, 350_000), (1, 100_000), (3, 1_000_000), (9, 2_000_000), (1, 99_999), (1, 10)):
            # float-like data: compresses a little, blocks end at arbitrary bit offsets
            raw = (rng.normal(0, 0.025, nbytes // 4 + 1).astype(">f4").tobytes() + bytes(rng.integers(0, 4, 1000, dtype=np.uint8)))[:nbytes]
            comp = bz2.compress(raw, level)
            streams = bz2blocks.split_blocks(comp)
            assert streams is not None
            want_blocks = -(-len(raw) // (level * 100_000 - 19))   # libbz2 fills blocks to 100k * level - 19 bytes
            assert abs(len(streams) - want_blocks) <= 1, (level, nbytes, len(streams), want_blocks)
            assert b"".join(bz2.decompress(s) for s in streams) == raw
            assert bz2blocks.decompress(comp, pool) == raw
            assert bz2blocks.decompress(comp) == raw


def test_what_is_not_one_plain_stream_takes_the_ordinary_route():
    raw = bytes(range(256)) * 2000
    with ThreadPoolExecutor(2) as pool:
        two = bz2.compress(raw[:300_000], 1) + bz2.compress(raw[300_000:], 1)     # two concatenated streams
        assert bz2blocks.split_blocks(two) is None and bz2blocks.decompress(two, pool) == raw
        assert bz2blocks.split_blocks(b"not bzip2 at all") is None
        empty = bz2.compress(b"")
        assert bz2blocks.decompress(empty, pool) == b""
        # a block magic planted in the data being compressed does not survive compression as a magic; one planted in the
        # compressed stream breaks a block's CRC: the whole stream is decoded the ordinary way (and raises what bz2 raises)
        comp = bytearray(bz2.compress(raw, 1))
        comp[len(comp) // 2:len(comp) // 2 + 6] = bytes.fromhex("314159265359")
        try:
            bz2blocks.decompress(bytes(comp), pool)
            raised = False
        except (OSError, ValueError, EOFError):
            raised = True
        assert raised
