"""Frame ingest for ``DetectTrails.process``: a pool of reader threads that puts the frames of a chunk straight into the
library's page-locked staging memory.

The reference reads one frame at a time with ``fitsio`` (detecttrails.py:73-117: decompress a ``.bz2`` to ``$FITS_DUMP`` if
need be, ``fitsio.read`` -- a read, a byte swap and a copy --, then the photoObj table, removestars.py:96-104) between two
GPU-sized pieces of work.  Here the file bytes of a ``BITPIX = -32`` image are the big-endian float32 frame, so a reader
thread parses the header, ``readinto``s the data unit into its slot of a pinned buffer (``bz2.decompress`` + one copy for
``.fits.bz2``) and the library uploads that memory in place and swaps the bytes on the device
(``lfdmi_detect_batch_raw(..., LFDMI_F32_BE, ..., LFDMI_HOST_PINNED)``).  ``open`` / ``readinto`` / ``bz2`` / large numpy
copies release the GIL, so the pool scales with the cores; what stays under the GIL per frame is a header scan and the
column views of the photoObj table (~0.2 ms).  Two pinned buffers: chunk k + 1 is read while chunk k is on the GPU.

Frames this fast path cannot take (other BITPIX, BSCALE / BZERO, a different shape) are read by ``fitslite`` and returned as
native float32 arrays for the ordinary per-frame path; a missing file is that frame's error, as in the reference.
"""
import bz2
import os
import re
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import fitslite, sdssfiles

BLOCK = fitslite.BLOCK
_END = b"END" + b" " * 77
_TCARD = re.compile(rb"(TFORM|TTYPE)(\d+) *= *'([^']*)'")
_CAT5 = ("ROWC", "COLC", "PSFMAG", "PETROTH90")
_CAT1 = ("NOBSERVE", "NDETECT")


def header_end(buf, start=0):
    """Offset just past the header that starts at ``start`` (a multiple of 2880), or -1 if its END card is not in ``buf``."""
    pos = start
    while True:
        i = buf.find(_END, pos)
        if i < 0:
            return -1
        if (i - start) % 80 == 0:
            return start + ((i - start) // BLOCK + 1) * BLOCK
        pos = i + 1


def card_value(hdr, key):
    """Value of header card ``key`` (bytes, <= 8 chars) as fitslite parses it, or None when the card is absent."""
    k = key.ljust(8) + b"= "
    pos = 0
    while True:
        i = hdr.find(k, pos)
        if i < 0:
            return None
        if i % 80 == 0:
            return fitslite._parse_card(hdr[i:i + 80].decode("ascii", "replace"))[1]
        pos = i + 1


def read_catalog(path):
    """The six photoObj columns remove_stars uses (removestars.py:97-104 reads eight; OBJC_TYPE and TYPE are never looked at)
    as native arrays: ROWC / COLC / PSFMAG / PETROTH90 [n, 5] float32, NOBSERVE / NDETECT [n] int32.  One regex pass over the
    table header instead of a card-by-card parse."""
    with open(path, "rb", buffering=0) as f:
        buf = f.read()
    e0 = header_end(buf, 0)
    if e0 < 0:
        raise ValueError(f"{path}: truncated FITS header")
    primary = buf[:e0]
    naxis = card_value(primary, b"NAXIS") or 0
    off = e0
    if naxis:
        n = abs(card_value(primary, b"BITPIX")) // 8
        for i in range(1, naxis + 1):
            n *= card_value(primary, b"NAXIS%d" % i)
        off += (n + BLOCK - 1) // BLOCK * BLOCK
    e1 = header_end(buf, off)
    if e1 < 0:
        raise ValueError(f"{path}: truncated FITS header")
    hdr = buf[off:e1]
    if str(card_value(hdr, b"XTENSION") or "").strip() != "BINTABLE":
        raise ValueError(f"{path}: extension 1 is not a binary table")
    row_bytes, nrows, nf = card_value(hdr, b"NAXIS1"), card_value(hdr, b"NAXIS2"), card_value(hdr, b"TFIELDS")
    forms, names = [None] * (nf + 1), {}
    for kind, idx, val in _TCARD.findall(hdr):
        i = int(idx)
        if i > nf:
            continue
        if kind == b"TFORM":
            forms[i] = val.strip()
        else:
            names[val.strip().upper().decode("ascii", "replace")] = i
    want = {names.get(c) for c in _CAT5 + _CAT1}
    if None in want:
        raise KeyError(f"{path}: missing columns {sorted(c for c in _CAT5 + _CAT1 if c not in names)}")
    out, pos = {}, 0
    inv = {i: c for c, i in names.items() if i in want}
    for i in range(1, nf + 1):
        form = forms[i]
        if form is None:
            raise ValueError(f"{path}: TFORM{i} missing")
        j = 0
        while j < len(form) and 48 <= form[j] <= 57:
            j += 1
        rep = int(form[:j]) if j else 1
        code = chr(form[j])
        if code in "PQ":
            width = (8 if code == "P" else 16) * rep
        elif code == "X":
            width = (rep + 7) // 8
        elif code in "CM":
            width = (8 if code == "C" else 16) * rep
        else:
            dt, size = fitslite._TFORM[code]
            width = size * rep
            if i in inv:
                if hdr.find(b"TSCAL%d" % i) >= 0 or hdr.find(b"TZERO%d" % i) >= 0:
                    return None                              # scaled columns: the general reader handles them
                name = inv[i]
                col = np.ndarray((nrows, rep), dt, buf, e1 + pos, (row_bytes, np.dtype(dt).itemsize))
                out[name] = col.astype(np.float32 if name in _CAT5 else np.int32)
                if name in _CAT1:
                    out[name] = out[name][:, 0]
        pos += width
    return out


def _catalog(run, camcol, field):
    path = sdssfiles.filename("photoObj", run=run, camcol=camcol, field=field)
    cat = read_catalog(path)
    if cat is None:
        from .removestars import read_photoObj_arrays
        cat = read_photoObj_arrays(path)
    return cat


class Loaded:
    """What the pool made of one chunk: ``keys`` in the caller's order; for key i either ``slot[i]`` >= 0 (its raw
    big-endian frame is slot ``slot[i]`` of the pinned buffer), or ``array[i]`` (a native float32 frame for the ordinary
    path), or ``error[i]`` (the exception the reference would have logged); ``hdr[i]`` = raw header bytes (fast path) or
    the parsed dict; ``cat[i]`` = photoObj columns."""

    def __init__(self, keys):
        n = len(keys)
        self.keys = list(keys)
        self.slot = [-1] * n
        self.array = [None] * n
        self.error = [None] * n
        self.hdr = [None] * n
        self.cat = [None] * n
        self.buffer = None                                   # [slots, h, w] '>f4' view of the pinned memory


class FrameLoader:
    """``threads`` readers, two pinned buffers of ``slots`` frames of ``shape`` each (allocated through ``ctx``)."""

    def __init__(self, ctx, shape, slots, threads=None):
        self.shape = tuple(shape)
        self.slots = int(slots)
        h, w = self.shape
        self.frame_bytes = h * w * 4
        try:
            cores = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            cores = os.cpu_count() or 1
        self.threads = int(threads or os.environ.get("LFD_LOADER_THREADS", 0) or max(2, min(32, cores)))
        self.pins = [ctx.pinned_buffer(self.slots * self.frame_bytes) for _ in range(2)]
        self.views = [p.array.view(">f4").reshape(self.slots, h, w) for p in self.pins]
        self.pool = ThreadPoolExecutor(self.threads, thread_name_prefix="lfd-loader")

    def close(self):
        self.pool.shutdown(wait=True)
        self.views = None
        for p in self.pins:
            p.close()
        self.pins = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- one frame ------------------------------------------------------------------------------------------------
    def _read_frame(self, key, dst_u8):
        """Frame ``key`` into ``dst_u8`` (this slot's bytes) when the file is a plain big-endian float32 image of the expected
        shape: returns (True, raw header bytes).  Otherwise (False, (native float32 array, header dict))."""
        run, camcol, flt, field = key
        path = sdssfiles.filename("frame", run=run, camcol=camcol, field=field, filter=flt)
        if os.path.exists(path):
            with open(path, "rb", buffering=0) as f:
                head = f.read(4 * BLOCK)
                end = header_end(head)
                while end < 0:
                    more = f.read(4 * BLOCK)
                    if not more:
                        raise ValueError("truncated FITS header")
                    head += more
                    end = header_end(head)
                if len(head) < end:
                    head += f.read(end - len(head))
                    if len(head) < end:
                        raise ValueError("truncated FITS header")
                hdr = head[:end]
                if self._fast(hdr):
                    got = len(head) - end                    # data bytes that came with the header read
                    mv = memoryview(dst_u8)
                    if got:
                        k = min(got, self.frame_bytes)
                        mv[:k] = head[end:end + k]
                    while got < self.frame_bytes:
                        r = f.readinto(mv[got:])
                        if not r:
                            raise ValueError(f"{path}: file ends inside the image")
                        got += r
                    return True, hdr
            img, h = fitslite.read_image(path)
            return False, (np.ascontiguousarray(img, dtype=np.float32), h)
        if not os.path.exists(path + ".bz2"):
            raise FileNotFoundError(("File {0} or its bz2 compressed version not found. "
                                     "Are you sure they exist?").format(path))
        with open(path + ".bz2", "rb", buffering=0) as f:
            raw = bz2.decompress(f.read())                   # in memory; no $FITS_DUMP round trip (detecttrails.py:88-109)
        end = header_end(raw)
        if end < 0 or len(raw) < end:
            raise ValueError("truncated FITS header")
        hdr = raw[:end]
        if self._fast(hdr):
            if len(raw) < end + self.frame_bytes:
                raise ValueError(f"{path}.bz2: file ends inside the image")
            np.copyto(dst_u8, np.frombuffer(raw, np.uint8, self.frame_bytes, end))
            return True, hdr
        img, h = fitslite.read_image_bytes(raw, path + ".bz2")
        return False, (np.ascontiguousarray(img, dtype=np.float32), h)

    def _fast(self, hdr):
        h, w = self.shape
        return (card_value(hdr, b"BITPIX") == -32 and card_value(hdr, b"NAXIS") == 2 and card_value(hdr, b"NAXIS1") == w
                and card_value(hdr, b"NAXIS2") == h and card_value(hdr, b"BSCALE") in (None, 1, 1.0)
                and card_value(hdr, b"BZERO") in (None, 0, 0.0))

    def _job(self, out, i, slot, dst_u8):
        try:
            fast, what = self._read_frame(out.keys[i], dst_u8)
            if fast:
                out.slot[i], out.hdr[i] = slot, what
            else:
                out.array[i], out.hdr[i] = what
            run, camcol, _, field = out.keys[i]
            out.cat[i] = _catalog(run, camcol, field)
        except Exception as e:  # noqa: BLE001 - this frame's errors.txt entry (detecttrails.py:133-139)
            out.slot[i] = -1
            out.array[i] = None
            out.error[i] = e

    # -- a chunk --------------------------------------------------------------------------------------------------
    def load(self, keys, which):
        """Read ``keys`` (at most ``slots``) into pinned buffer ``which`` (0 / 1).  Frames of one filter get neighbouring slots
        (remove_stars' magnitude cap depends on the filter, so a GPU call takes one filter's frames: a contiguous slice)."""
        if len(keys) > self.slots:
            raise ValueError("chunk larger than the loader's buffers")
        out = Loaded(keys)
        out.buffer = self.views[which]
        raw = self.pins[which].array
        order = sorted(range(len(keys)), key=lambda i: keys[i][2])      # stable: the caller's order inside a filter
        futs = []
        for slot, i in enumerate(order):
            dst = raw[slot * self.frame_bytes:(slot + 1) * self.frame_bytes]
            futs.append(self.pool.submit(self._job, out, i, slot, dst))
        for f in futs:
            f.result()
        return out


def header_values(hdr, keys):
    """The header values of ``keys`` from raw header bytes (fast path) or a parsed dict, as fitslite would give them."""
    if isinstance(hdr, dict):
        return [hdr[k] for k in keys]
    vals = []
    for k in keys:
        v = card_value(hdr, k.encode("ascii"))
        if v is None:
            raise KeyError(k)
        vals.append(v)
    return vals
