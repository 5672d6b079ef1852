"""bzip2 on the device (lfdmi_bz2_*, lfd_amd/csrc/k_bz2.h): whole files against Python's bz2 module -- the step the reference
performs with `bunzip2` in front of every compressed frame (detecttrails.py:81-109) -- and the loader that uses it."""
import bz2
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pack(blobs):
    off, cur = [], 0
    for c in blobs:
        off.append(cur)
        cur += (len(c) + 7) & ~7
    src = np.zeros(max(cur, 8), np.uint8)
    for o, c in zip(off, blobs):
        src[o:o + len(c)] = np.frombuffer(c, np.uint8)
    return src, off, [len(c) for c in blobs]


def _plains():
    rng = np.random.default_rng(7)
    hdr = b"".join(c.ljust(80) for c in (b"SIMPLE  =                    T", b"BITPIX  =                  -32", b"END")).ljust(2880)
    img = rng.normal(0.0, 0.025, (700, 1024)).astype(">f4")
    img[100:180, 50:900] = 0.0                                       # long zero runs: the run-length layer and RUNA / RUNB
    sky = (rng.normal(1000.0, 3.0, (400, 1024)).astype(np.float32)).astype(">f4")    # compressible: few distinct high bytes
    return {
        "text": b"the quick brown fox jumps over the lazy dog. " * 40 + b"!",
        "runs": b"a" * 1000 + b"b" * 5 + bytes(300) + b"xyz" * 7 + b"\xfb" * 2000 + b"q" * 4 + b"r" * 259 + b"ssss",
        "noise": rng.integers(0, 256, 300000, dtype=np.uint8).tobytes(),
        "one": b"z",
        "four": b"zzzz",
        "five": b"zzzzz",
        "zeros": bytes(2_500_000),
        "fits": hdr + img.tobytes(),
        "sky": hdr + sky.tobytes(),
        "few_symbols": bytes(rng.integers(0, 3, 50000, dtype=np.uint8)),
        "count_equals_byte": bytes([5]) * 9 + b"x" + bytes([251]) * 600 + b"y",
        "periodic": b"abcd" * 5000,                                  # its BWT permutation has cycles shorter than the block
        "periodic_text": b"hello hello hello world" * 3,
        "allbytes": bytes(range(256)) * 50,
    }


@pytest.mark.parametrize("level", [1, 9])
def test_whole_files_against_python_bz2(level):
    from lfd_amd import _native as Nv
    plains = _plains()
    blobs = [bz2.compress(p, level) for p in plains.values()]
    src, off, ln = _pack(blobs)
    with Nv.Bz2Decoder(0) as z:
        out_len, status, heads = z.decode(src, off, ln, 4 << 20, 2880)
        for i, (name, p) in enumerate(plains.items()):
            assert status[i] == 0, (name, int(status[i]), Nv.BZ2_STATUS.get(int(status[i])))
            assert int(out_len[i]) == len(p), name
            assert z.fetch(i, 0, len(p)).tobytes() == p, name
            assert heads[i].tobytes() == p[:2880].ljust(2880, b"\0"), name
            if len(p) > 5000:
                assert z.fetch(i, 1234, 3000).tobytes() == p[1234:4234], name
        # a second batch on the same handle (buffers reused), other order
        src2, off2, ln2 = _pack(blobs[::-1])
        out_len2, status2, _ = z.decode(src2, off2, ln2, 4 << 20)
        assert list(out_len2) == list(out_len[::-1]) and not status2.any()


def test_streams_joined_end_to_end_are_one_file():
    """`bzip2 -c a b > ab.bz2`, pbzip2: several streams, each with its own level, blocks, end mark and CRC."""
    from lfd_amd import _native as Nv
    rng = np.random.default_rng(11)
    parts = [rng.integers(0, 256, 260000, dtype=np.uint8).tobytes(), b"", b"middle " * 1000, bytes(400000)]
    joined = b"".join(bz2.compress(p, lvl) for p, lvl in zip(parts, (1, 9, 5, 2)))
    assert bz2.decompress(joined) == b"".join(parts)
    damaged = bytearray(joined)
    damaged[-3] ^= 1                                                 # the last stream's CRC
    src, off, ln = _pack([joined, bytes(damaged)])
    with Nv.Bz2Decoder(0) as z:
        out_len, status, _ = z.decode(src, off, ln, 2 << 20)
        assert status[0] == 0 and z.fetch(0, 0, int(out_len[0])).tobytes() == b"".join(parts)
        assert status[1] != 0


def test_more_blocks_than_the_budget_are_decoded_in_passes(monkeypatch):
    """$LFDMI_BZ2_MAX_BLOCKS bounds the tables: whole files are grouped, every group is a pass of its own."""
    from lfd_amd import _native as Nv
    monkeypatch.setenv("LFDMI_BZ2_MAX_BLOCKS", "5")
    rng = np.random.default_rng(3)
    plains = [rng.integers(0, 256, 250000 + 50000 * k, dtype=np.uint8).tobytes() for k in range(5)] + [b"", b"tiny"]
    blobs = [bz2.compress(p, 1) for p in plains]                     # level 1: 100 kB blocks, 3 - 5 per file
    src, off, ln = _pack(blobs)
    with Nv.Bz2Decoder(0) as z:
        out_len, status, heads = z.decode(src, off, ln, 1 << 20, 16)
        assert not status.any(), list(status)
        for i, p in enumerate(plains):
            assert int(out_len[i]) == len(p)
            if p:
                assert z.fetch(i, 0, len(p)).tobytes() == p
            assert heads[i].tobytes() == p[:16].ljust(16, b"\0")


def test_what_the_decoder_declines_is_reported_not_guessed():
    from lfd_amd import _native as Nv
    good = bz2.compress(_plains()["noise"], 9)
    flipped = bytearray(good)
    flipped[len(flipped) // 2] ^= 0x10
    blobs = [bz2.compress(b"abc")[:-1] + b"BZ",                      # ends inside the stream's CRC
             bytes(flipped), b"not bzip2 at all", good + b"\0", good[:len(good) // 2], good]
    src, off, ln = _pack(blobs)
    with Nv.Bz2Decoder(0) as z:
        out_len, status, _ = z.decode(src, off, ln, 1 << 20)
        assert all(int(s) != 0 for s in status[:5]), list(status)
        assert status[5] == 0 and z.fetch(5, 0, int(out_len[5])).tobytes() == _plains()["noise"]
        with pytest.raises(Nv.NativeError):
            z.fetch(1, 0, 10)
        # too small an output buffer: declined as such
        out_len, status, _ = z.decode(*_pack([good]), 1000)
        assert int(status[0]) == 9


def test_loader_decodes_a_chunk_of_bz2_frames_on_the_device(tmp_path, monkeypatch):
    from lfd_amd import _native as Nv, synth
    from lfd_amd.detecttrails import loader, sdssfiles
    shape, n = (256, 384), 6
    frames, cats = [], []
    for k in range(n):
        img, cat, _ = synth.make_portable_frame(k, shape, n_star=7)
        frames.append(img)
        cats.append(cat)
    hdr = synth.write_boss_tree(tmp_path, frames, cats, field0=100, bz2_fields=set(range(100, 100 + n)))
    bad = sdssfiles.filename("frame", 94, 1, 103, "r") + ".bz2"     # one damaged file: the host decoder's error, as before
    data = bytearray(open(bad, "rb").read())
    data[len(data) // 2] ^= 0x40
    open(bad, "wb").write(bytes(data))
    keys = [(94, 1, "r", f) for f in range(100, 100 + n)]
    monkeypatch.setenv("LFD_BZ2_DEVICE_MIN", "2")
    with Nv.Context(0, shape[0], shape[1], 8) as ctx:
        with loader.FrameLoader(ctx, shape, 8, threads=3) as ld:
            out = ld.load(keys, 0)
            for i in range(n):
                if i == 3:
                    assert isinstance(out.error[i], (OSError, ValueError, EOFError))
                    continue
                assert out.error[i] is None and out.slot[i] >= 0
                assert np.array_equal(out.buffer[out.slot[i]].astype(np.float32), frames[i])
                assert loader.header_values(out.hdr[i], ["TAI"]) == [hdr["TAI"]]
            assert ld.bz2_stats["device_frames"] == n - 1 and ld.bz2_stats["host_frames"] == 1
            assert out.device is None                                # (one frame went the host way: the chunk meets in the pinned slots)
        monkeypatch.setenv("LFD_BZ2_DEVICE", "0")
        with loader.FrameLoader(ctx, shape, 8, threads=3) as ld:
            out2 = ld.load(keys, 0)
            assert ld.bz2_stats["device_frames"] == 0
            for i in range(n):
                if i != 3:
                    assert np.array_equal(out2.buffer[out2.slot[i]], out.buffer[out.slot[i]])
                else:
                    assert type(out2.error[i]) is type(out.error[i]) and str(out2.error[i]) == str(out.error[i])


def test_damaged_files_never_decode_to_something_else():
    """Bit flips, truncations, spliced and random bytes: whatever Python's bz2 makes of a damaged file, the device decoder either
    declines it or produces the same bytes (no wrong data, no fault)."""
    from lfd_amd import _native as Nv
    rng = np.random.default_rng(99)
    base = [bz2.compress(rng.integers(0, 256, 150000, dtype=np.uint8).tobytes(), 1),
            bz2.compress(bytes(rng.integers(0, 4, 200000, dtype=np.uint8)), 9),
            bz2.compress(b"".join(bytes([b]) * int(n) for b, n in zip(rng.integers(0, 256, 3000), rng.integers(1, 300, 3000))), 9)]
    blobs = []
    for k in range(240):
        b = bytearray(base[k % 3])
        kind = k % 4
        if kind == 0:                                                # one bit
            i = int(rng.integers(0, len(b)))
            b[i] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:                                              # a burst of random bytes
            i = int(rng.integers(0, len(b) - 16))
            b[i:i + 16] = rng.integers(0, 256, 16, dtype=np.uint8).tobytes()
        elif kind == 2:                                              # cut short / something appended
            b = b[:int(rng.integers(4, len(b)))] if k % 8 == 2 else b + bytes(rng.integers(0, 256, 5, dtype=np.uint8))
        else:                                                        # the header region
            i = int(rng.integers(0, min(200, len(b))))
            b[i] = int(rng.integers(0, 256))
        blobs.append(bytes(b))
    want = []
    for b in blobs:
        try:
            want.append(bz2.decompress(b))
        except (OSError, ValueError, EOFError):
            want.append(None)
    src, off, ln = _pack(blobs)
    with Nv.Bz2Decoder(0) as z:
        out_len, status, _ = z.decode(src, off, ln, 1 << 20)
        agreed = 0
        for i, w in enumerate(want):
            if status[i] == 0:
                assert w is not None and z.fetch(i, 0, int(out_len[i])).tobytes() == w, i
                agreed += 1
        assert agreed <= sum(w is not None for w in want)
        # and the handle is fine afterwards
        good = base[0]
        ol, st, _ = z.decode(*_pack([good]), 1 << 20)
        assert st[0] == 0 and z.fetch(0, 0, int(ol[0])).tobytes() == bz2.decompress(good)


def test_a_chunk_of_compressed_frames_stays_on_the_device(tmp_path, monkeypatch, oracle):
    """All frames of a chunk exist only as .fits.bz2: they are decompressed on the GPU, their data units gathered in device memory
    and handed to the detection kernels there (big-endian, swapped in place); rows equal the oracle's, and equal the run that sends
    the decoded frames through the pinned slots."""
    from lfd_amd import results, synth
    from lfd_amd.detecttrails import DetectTrails, default_params, loader, sdssfiles
    shape, n = (512, 768), 10
    frames, cats = [], []
    for k in range(n):
        img, cat, _ = synth.make_portable_frame(k, shape)
        frames.append(img)
        cats.append(cat)
    hdr = synth.write_boss_tree(tmp_path, frames, cats, field0=100, bz2_all=True)
    pb, pd, prs = default_params()
    rs = oracle.rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})
    want = []
    for k in range(n):
        rec = oracle.detect_frame(frames[k].copy(), pb, pd, cats[k], rs)
        if rec["found"]:
            want.append(results.format_result_row(94, 1, "r", 100 + k, hdr, rec))
    assert len(want) >= 3
    seen = []
    real_load = loader.FrameLoader.load

    def spy(self, *a, **k):
        out = real_load(self, *a, **k)
        seen.append(out.device is not None)
        return out
    monkeypatch.setattr(loader.FrameLoader, "load", spy)
    rows = {}
    for keep in ("1", "0"):
        monkeypatch.setenv("LFD_BZ2_KEEP_ON_DEVICE", keep)
        save = tmp_path / ("out" + keep)
        save.mkdir()
        dt = DetectTrails(run=94, camcol=1, filter="r", savepath=str(save))
        dt.process(batch=16)
        rows[keep] = [ln.strip() for ln in open(dt.results) if ln.strip()]
        assert open(dt.errors).read() == ""
        assert dt.last_stats["bz2"]["device_frames"] == n
    assert seen == [True, False]
    assert rows["1"] == want and rows["0"] == want


def test_two_chunks_loading_at_once_over_compressed_mixed_and_plain_chunks(tmp_path, monkeypatch, oracle):
    """LFD_LOADER_DEPTH=2 (two decoders, three sets of chunk buffers): a selection whose chunks are all-compressed (stay on the device),
    mixed (meet in the pinned slots) and all-plain, in that order and back; rows equal the oracle's, in the selection's order."""
    from lfd_amd import results, synth
    from lfd_amd.detecttrails import DetectTrails, default_params, fitslite, sdssfiles
    shape, n = (512, 768), 40
    frames, cats = [], []
    for k in range(n):
        img, cat, _ = synth.make_portable_frame(k % 12, shape)
        frames.append(img)
        cats.append(cat)
    compressed = set(range(100, 110)) | {112, 113, 114, 115, 116, 118} | set(range(130, 140))   # chunks: all, mixed, none, all
    hdr = synth.write_boss_tree(tmp_path, frames, cats, field0=100, bz2_fields=compressed)
    pb, pd, prs = default_params()
    rs = oracle.rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})
    want = []
    for k in range(n):
        rec = oracle.detect_frame(frames[k].copy(), pb, pd, cats[k], rs)
        if rec["found"]:
            want.append(results.format_result_row(94, 1, "r", 100 + k, hdr, rec))
    assert len(want) >= 10
    monkeypatch.setenv("LFD_BZ2_DEVICE_MIN", "4")
    monkeypatch.setenv("LFD_LOADER_DEPTH", "2")
    dt = DetectTrails(run=94, camcol=1, filter="r", savepath=str(tmp_path))
    dt.process(batch=10)
    assert [ln.strip() for ln in open(dt.results) if ln.strip()] == want
    assert open(dt.errors).read() == ""
    assert dt.last_stats["bz2"]["device_frames"] == 26 and dt.last_stats["bz2"]["host_frames"] == 0


def test_files_written_by_the_bzip2_program(tmp_path):
    """The reference's own tool chain: files compressed by the `bzip2` program (as SDSS serves them) and what `bunzip2` makes of them
    (detecttrails.py:88-109 runs exactly that), single files and `bzip2 -c a b`."""
    import shutil
    import subprocess
    from lfd_amd import _native as Nv
    if shutil.which("bzip2") is None or shutil.which("bunzip2") is None:
        pytest.skip("no bzip2 program")
    plains = _plains()
    a, b = tmp_path / "a.fits", tmp_path / "b.fits"
    a.write_bytes(plains["fits"])
    b.write_bytes(plains["sky"])
    blobs = [subprocess.run(["bzip2", "-9", "-c", str(a)], check=True, capture_output=True).stdout,
             subprocess.run(["bzip2", "-1", "-c", str(b)], check=True, capture_output=True).stdout,
             subprocess.run(["bzip2", "-c", str(a), str(b)], check=True, capture_output=True).stdout]
    want = [subprocess.run(["bunzip2", "-c"], input=x, check=True, capture_output=True).stdout for x in blobs]
    assert want[0] == plains["fits"] and want[2] == plains["fits"] + plains["sky"]
    src, off, ln = _pack(blobs)
    with Nv.Bz2Decoder(0) as z:
        out_len, status, _ = z.decode(src, off, ln, 8 << 20)
        for i in range(3):
            assert status[i] == 0 and z.fetch(i, 0, int(out_len[i])).tobytes() == want[i], i
