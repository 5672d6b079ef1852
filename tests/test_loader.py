"""The frame ingest of DetectTrails.process (lfd_amd/detecttrails/loader.py + the native readers of lfd_amd/csrc/fits_reader.h)
without a GPU: the reader threads fill the (here: ordinary numpy) staging buffers with the raw big-endian data units of the
frame files and the padded catalogue arrays with the photoObj columns, frames of a filter get neighbouring slots, files the
fast path declines go through the general reader, and missing files become per-frame errors (reference flow:
detecttrails.py:73-117, removestars.py:96-130).  The library is only used for its host-side entry points here."""
import bz2
import os

import numpy as np

from lfd_amd import synth
from lfd_amd.detecttrails import fitslite, loader, sdssfiles


class _Pin:
    def __init__(self, nbytes):
        self.array = np.zeros(nbytes, np.uint8)

    def close(self):
        self.array = None


class _Ctx:                                                          # stands in for _native.Context.pinned_buffer
    def pinned_buffer(self, nbytes):
        return _Pin(nbytes)


def _tree(tmp_path, n=6, shape=(64, 96)):
    frames, cats = [], []
    for k in range(n):
        img, cat, _ = synth.make_portable_frame(k, shape, n_star=7)
        frames.append(img)
        cats.append(cat)
    cats[3] = None                                                   # no photoObj file for field 103
    hdr = synth.write_boss_tree(tmp_path, frames, cats, field0=100, bz2_fields={101})
    return frames, cats, hdr


def test_chunk_into_staging_memory(tmp_path, monkeypatch):
    frames, cats, hdr = _tree(tmp_path)
    # field 104: an int16 image with BZERO (not a plain float32 frame): the general reader takes it
    p4 = sdssfiles.filename("frame", 94, 1, 104, "r")
    fitslite.write_image(p4, np.arange(64 * 96, dtype=np.int16).reshape(64, 96), dict(hdr, BZERO=3.0, BSCALE=2.0))
    os.remove(sdssfiles.filename("frame", 94, 1, 105, "r"))          # field 105: frame file missing altogether
    keys = [(94, 1, "r", f) for f in range(100, 106)]
    with loader.FrameLoader(_Ctx(), (64, 96), 8, threads=3) as ld:
        out = ld.load(keys, 1)
        assert out.buffer.dtype == np.dtype(">f4") and out.buffer.shape == (8, 64, 96)
        for i in (0, 1, 2):                                          # plain, .bz2, plain: raw big-endian data in their slots
            assert out.slot[i] >= 0 and out.error[i] is None
            assert np.array_equal(out.buffer[out.slot[i]].astype(np.float32), frames[i])
            assert loader.header_values(out.hdr[i], ["TAI", "CD2_1"]) == [hdr["TAI"], hdr["CD2_1"]]
            got = out.cat_of(i)
            for k in ("ROWC", "COLC", "PSFMAG", "PETROTH90", "NOBSERVE", "NDETECT"):
                assert np.array_equal(got[k], cats[i][k]) and got[k].dtype == cats[i][k].dtype
            assert out.cats["count"][out.slot[i]] == len(cats[i]["NOBSERVE"])
        assert isinstance(out.error[3], FileNotFoundError) and "photoObj" in str(out.error[3])
        assert out.slot[4] < 0 and out.error[4] is None
        assert out.array[4].dtype == np.float32 and out.array[4][1, 1] == (96 + 1) * 2.0 + 3.0
        assert isinstance(out.hdr[4], dict)
        assert isinstance(out.error[5], FileNotFoundError) and "bz2 compressed version not found" in str(out.error[5])
        again = ld.load(keys[:3], 0)                                  # the other buffer
        assert np.array_equal(again.buffer[again.slot[2]].astype(np.float32), frames[2])


def test_frames_of_one_filter_get_neighbouring_slots(tmp_path):
    frames, cats, hdr = _tree(tmp_path, n=4)
    for f in range(100, 104):                                        # the same pixels as filter 'g' files
        src = sdssfiles.filename("frame", 94, 1, f, "r")
        if os.path.exists(src):
            os.link(src, sdssfiles.filename("frame", 94, 1, f, "g"))
        else:
            os.link(src + ".bz2", sdssfiles.filename("frame", 94, 1, f, "g") + ".bz2")
    keys = [(94, 1, flt, f) for f in (100, 101, 102) for flt in ("r", "g")]   # interleaved filters
    with loader.FrameLoader(_Ctx(), (64, 96), 6, threads=2) as ld:
        out = ld.load(keys, 0)
    slots_g = sorted(out.slot[i] for i, k in enumerate(keys) if k[2] == "g")
    slots_r = sorted(out.slot[i] for i, k in enumerate(keys) if k[2] == "r")
    assert slots_g == [0, 1, 2] and slots_r == [3, 4, 5]
    for i, k in enumerate(keys):                                      # ... in the caller's order inside a filter
        assert np.array_equal(out.buffer[out.slot[i]].astype(np.float32), frames[k[3] - 100])
    assert [out.slot[i] for i, k in enumerate(keys) if k[2] == "r"] == [3, 4, 5]


def test_catalog_reader_matches_the_general_one(tmp_path):
    """The native photoObj reader against fitslite.read_table on the astropy-written photoObj fixture (a variable-length
    column and unrelated columns before the wanted ones)."""
    import glob
    fix = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "fits", "photoobj*.fits")))
    assert fix, "photoObj fixture missing"
    for path in fix:
        want = fitslite.read_table(path, ("ROWC", "COLC", "PETROTH90", "PSFMAG", "NOBSERVE", "NDETECT"))
        got = loader.read_catalog(path)
        if got is None:                                               # scaled columns: handed to the general reader by design
            continue
        for k, v in want.items():
            assert np.array_equal(got[k], np.asarray(v, got[k].dtype)), k


def test_header_scan_helpers():
    cards = [fitslite._card("SIMPLE", True), fitslite._card("BITPIX", -32), fitslite._card("NAXIS", 2),
             fitslite._card("NAXIS1", 96), fitslite._card("NAXIS2", 64), fitslite._card("COMMENT", "END is near"),
             fitslite._card("TAI", 4649973000.5)]
    hdr = fitslite._finish_header(cards)
    assert loader.header_end(hdr) == 2880 and loader.header_end(hdr[:500]) == -1
    assert loader.card_value(hdr, b"BITPIX") == -32 and loader.card_value(hdr, b"TAI") == 4649973000.5
    assert loader.card_value(hdr, b"BZERO") is None
    long = fitslite._finish_header(cards + [fitslite._card("K%d" % i, i) for i in range(40)])
    assert loader.header_end(long + b"\0" * 100) == 2 * 2880
    assert bz2.decompress(bz2.compress(long)) == long


def test_native_reader_statuses(tmp_path):
    """NaN / infinity in a catalogue (math.ceil raises in removestars.py:113-130), more rows than the padded arrays hold, a
    truncated frame file, a header longer than the kept copy: each handled as the reference's flow would, frame by frame."""
    frames, cats, hdr = _tree(tmp_path, n=6)
    from lfd_amd.detecttrails import fitslite as F
    def rewrite(field, cat):
        cols = dict(cat)
        m = len(cols["NOBSERVE"])
        cols["OBJC_TYPE"] = np.zeros(m, np.int32)
        cols["TYPE"] = np.zeros((m, 5), np.int32)
        F.write_table(sdssfiles.filename("photoObj", 94, 1, field), cols)
    bad = {k: v.copy() for k, v in cats[0].items()}
    bad["PSFMAG"][2, 3] = np.nan
    rewrite(100, bad)
    bad = {k: v.copy() for k, v in cats[1].items()}
    bad["ROWC"][0, 0] = np.inf
    rewrite(101, bad)
    big = {k: np.concatenate([v] * 3) for k, v in cats[2].items()}       # 21 rows > max_obj = 16 below
    rewrite(102, big)
    rewrite(103, cats[0])
    p4 = sdssfiles.filename("frame", 94, 1, 104, "r")                    # truncated inside the image
    data = open(p4, "rb").read()
    open(p4, "wb").write(data[:len(data) // 2])
    long_hdr = dict(hdr, **{"K%03d" % i: float(i) for i in range(300)})   # 9 header blocks > HDR_CAP
    F.write_image(sdssfiles.filename("frame", 94, 1, 105, "r"), frames[5], long_hdr)
    keys = [(94, 1, "r", f) for f in range(100, 106)]
    with loader.FrameLoader(_Ctx(), (64, 96), 8, threads=2, max_obj=16) as ld:
        out = ld.load(keys, 0)
    assert isinstance(out.error[0], ValueError) and "NaN" in str(out.error[0])
    assert isinstance(out.error[1], OverflowError)
    assert out.error[2] is None and out.slot[2] < 0 and np.array_equal(out.array[2], frames[2])
    assert len(out.cat_of(2)["NOBSERVE"]) == 21 and np.array_equal(out.cat_of(2)["ROWC"], big["ROWC"])
    assert out.error[3] is None and out.slot[3] >= 0 and np.array_equal(out.cat_of(3)["COLC"], cats[0]["COLC"])
    assert out.error[4] is not None and out.slot[4] < 0
    assert out.slot[5] >= 0 and len(out.hdr[5]) > loader.HDR_CAP
    assert loader.header_values(out.hdr[5], ["K299", "TAI"]) == [299.0, hdr["TAI"]]
    assert np.array_equal(out.buffer[out.slot[5]].astype(np.float32), frames[5])


def test_table_without_the_type_columns_is_a_keyerror_for_every_batch_size(tmp_path):
    """read_photoObj (removestars.py:97-104) also asks for OBJC_TYPE and TYPE: a table without them is a KeyError (an
    errors.txt entry) in the reference and frame by frame; the native batch reader must not quietly accept it."""
    frames, cats, hdr = _tree(tmp_path)
    from lfd_amd.detecttrails import fitslite as F
    F.write_table(sdssfiles.filename("photoObj", 94, 1, 100), dict(cats[0]))                 # the six columns only
    keys = [(94, 1, "r", f) for f in (100, 101)]
    with loader.FrameLoader(_Ctx(), (64, 96), 4, threads=2) as ld:
        out = ld.load(keys, 0)
    assert isinstance(out.error[0], KeyError) and out.slot[0] < 0
    assert out.error[1] is None and out.slot[1] >= 0
    assert loader.read_catalog(sdssfiles.filename("photoObj", 94, 1, 100)) is None


def test_truncated_and_hostile_table_headers_are_declined_not_read_past(tmp_path):
    """A photoObj file cut inside its table header's last block, and headers whose sizes do not fit the file:
    status 'malformed', nothing read outside the buffer (run under ASan by tools/oracle_sanitize.sh)."""
    frames, cats, hdr = _tree(tmp_path)
    p = sdssfiles.filename("photoObj", 94, 1, 100)
    data = open(p, "rb").read()
    e0 = loader.header_end(data)
    e1 = e0 + loader.header_end(data[e0:])
    cases = {"cut_in_table_header": data[:e1 - 1000], "cut_in_primary_header": data[:e0 - 100], "cut_in_rows": data[:e1 + 10]}
    tbl = bytearray(data)
    k = data.index(b"NAXIS1  =", e0)
    tbl[k:k + 30] = b"NAXIS1  =  9223372036854775807"                                        # row_bytes * nrows wraps 64 bits
    cases["row_bytes_wraps"] = bytes(tbl)
    tbl = bytearray(data)
    k = data.index(b"TFORM1  =", e0)
    tbl[k:k + 40] = (b"TFORM1  = '99999999999999999999E'" + b" " * 40)[:40]
    cases["repeat_count_overflows"] = bytes(tbl)
    for name, blob in cases.items():
        q = str(tmp_path / (name + ".fits"))
        open(q, "wb").write(blob)
        assert loader.read_catalog(q) is None, name
