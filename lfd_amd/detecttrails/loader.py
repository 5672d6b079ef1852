"""Frame ingest for ``DetectTrails.process``: the files of a chunk go straight into the library's page-locked staging memory,
read by native threads.

The reference reads one frame at a time with ``fitsio`` (detecttrails.py:73-117: decompress a ``.bz2`` to ``$FITS_DUMP`` if
need be, ``fitsio.read`` -- a read, a byte swap and a copy --, then the photoObj table and a Python loop over its rows,
removestars.py:96-130) between two GPU-sized pieces of work.  Here the file bytes of a ``BITPIX = -32`` image ARE the
big-endian float32 frame, so ``lfdmi_fits_read_frames`` (lfd_amd/csrc/fits_reader.h) lets a pool of C++ threads scan the
headers and ``pread`` the data units into the slots of a pinned buffer, ``lfdmi_fits_read_photoobj`` puts the six catalogue
columns remove_stars uses into the padded arrays the GPU call takes, and the library uploads the pinned memory in place and
swaps the bytes on the device (``lfdmi_detect_batch_raw(..., LFDMI_F32_BE, ..., LFDMI_HOST_PINNED)``).  Two pinned buffers:
chunk k + 1 is read while chunk k is on the GPU.  No per-frame Python runs on the fast path: an earlier version with Python
reader threads (``readinto`` + header scans) read as fast but held the interpreter lock 0.3 ms per frame, which doubled the
time of the thread driving the GPU.

What the native readers decline comes back here: ``.fits.bz2`` frames (``bz2.decompress`` in a small thread pool, then the
same slot), other BITPIX / BSCALE / BZERO / shapes and unusual photoObj layouts (``fitslite``, then the ordinary per-frame
path); a missing file is that frame's error, as in the reference.
"""
import bz2
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .. import _native
from . import bz2blocks, fitslite, sdssfiles

BLOCK = fitslite.BLOCK
_END = b"END" + b" " * 77
_CAT5 = ("ROWC", "COLC", "PSFMAG", "PETROTH90")
_CAT1 = ("NOBSERVE", "NDETECT")
HDR_CAP = 8 * BLOCK          # raw header bytes kept per frame (SDSS frame headers are three to four blocks)
MAX_OBJ = 4096               # catalogue rows per field the padded arrays hold (photoObj fields have a few hundred to ~2 000)


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


def _paths(strings):
    arr = (C.c_char_p * len(strings))()
    arr[:] = [os.fsencode(s) for s in strings]
    return arr


class Loaded:
    """What the loader made of one chunk: ``keys`` in the caller's order; for key i either ``slot[i]`` >= 0 (its raw
    big-endian frame is slot ``slot[i]`` of the pinned buffer ``buffer`` and its catalogue row ``slot[i]`` of ``cats``), or
    ``array[i]`` (a native float32 frame for the ordinary per-frame path, catalogue in ``cat[i]``), or ``error[i]`` (the
    exception the reference would have logged); ``hdr[i]`` = raw header bytes or a parsed dict."""

    def __init__(self, keys):
        n = len(keys)
        self.keys = list(keys)
        self.slot = [-1] * n
        self.array = [None] * n
        self.error = [None] * n
        self.hdr = [None] * n
        self.cat = [None] * n
        self.buffer = None                                   # [slots, h, w] '>f4' view of the pinned memory
        self.shape = None                                    # (h, w) of the chunk's frames
        self.device = None                                   # or: _native.DeviceFrames (the same slots in device memory: frames that
                                                             # were decompressed on the GPU and stayed there; ``buffer`` is then unused)
        self.fetch = None                                    # slot -> '>f4' (h, w) host copy of a device frame (valid while the chunk is loaded)

    def frame_host(self, slot):
        """Slot ``slot`` as a host array, wherever the chunk's frames are."""
        return self.buffer[slot] if self.device is None else self.fetch(slot)
        self.cats = None                                     # padded catalogue arrays of the buffer's slots + "count"

    def cat_of(self, i):
        """photoObj columns of key i as a dict of arrays (what the per-frame path takes)."""
        if self.cat[i] is not None or self.slot[i] < 0:
            return self.cat[i]
        s = self.slot[i]
        m = int(self.cats["count"][s])
        return {k: self.cats[k][s, :m] for k in _CAT5 + _CAT1}


class FrameLoader:
    """``threads`` native reader threads, two pinned buffers of ``slots`` frames of ``shape`` (allocated through ``ctx``)."""

    def __init__(self, ctx, shape, slots, threads=None, max_obj=MAX_OBJ, depth=1, expect_bz2=False):
        self.shape = tuple(shape)
        self.slots = int(slots)
        self.max_obj = int(max_obj)
        h, w = self.shape
        self.frame_bytes = h * w * 4
        from .. import usable_cores
        cores = usable_cores()
        # this rank's share of the host: every core it may use (affinity mask and cgroup quota), divided among the ranks torchrun started on this node
        share = max(1, cores // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", 1))))
        self.threads = int(threads or os.environ.get("LFD_LOADER_THREADS", 0) or max(2, min(64, share)))
        self.lib = _native.lib()
        # depth chunks can be loading at once while one more is being processed: depth + 1 sets of buffers.  depth = 2 is for
        # selections that are decompressed on the GPU (two decoders: the Huffman stage of one chunk -- bound by the CUs' scalar
        # units -- overlaps the inverse BWT of the previous one -- bound by HBM's random-access rate)
        self.ctx = ctx
        self.depth = max(1, int(depth))
        self.nbuf = self.depth + 1
        self.pins = [None] * self.nbuf                       # page-locked chunk buffers, made when a chunk first needs one (a selection that
        self.views = [None] * self.nbuf                      # is decompressed on the GPU and stays there never does)
        if not expect_bz2:
            for k in range(self.nbuf):
                self._ensure(k)
        self.cats = []
        for _ in range(self.nbuf):
            c = {k: np.zeros((self.slots, self.max_obj, 5), np.float32) for k in _CAT5}
            c.update({k: np.zeros((self.slots, self.max_obj), np.int32) for k in _CAT1})
            c["count"] = np.zeros(self.slots, np.int32)
            self.cats.append(c)
        self.hdrs = [np.zeros((self.slots, HDR_CAP), np.uint8) for _ in range(self.nbuf)]
        # .bz2 frames and other exceptions to the fast path, one thread per usable core (libbz2 releases the interpreter lock).  A
        # chunk with at least as many .bz2 frames as threads decodes whole files, one per thread (nothing is cheaper per frame);
        # a smaller selection has the ~14 bzip2 blocks of each file decoded side by side (bz2blocks), so that one frame does not
        # wait ~0.4 s on one core while the others idle.  (Round 3 capped the pool at 16 threads whatever the rank's share.)
        self.pool = ThreadPoolExecutor(max(2, self.threads), thread_name_prefix="lfd-slow")
        self.block_pool = ThreadPoolExecutor(max(2, self.threads), thread_name_prefix="lfd-bz2")
        self.split_blocks = True
        # .bz2 frames are decompressed ON THE GPU, all of a chunk's files at once (lfdmi_bz2_decode_batch: hundreds of frames/s
        # against ~2 per host core); what the decoder declines (a broken block, trailing bytes: status != 0) goes the host way
        # above, which also raises what the reference would.  $LFD_BZ2_DEVICE=0: host only.
        self.ctx = ctx
        self.bz2_device = os.environ.get("LFD_BZ2_DEVICE", "1") != "0" and hasattr(ctx, "device")   # (a real _native.Context)
        self.bz2_out_cap = self.frame_bytes + int(os.environ.get("LFD_BZ2_EXTRA_MB", 4)) * (1 << 20)   # (a frame file = image + three small HDUs)
        self.bz2_keep_on_device = os.environ.get("LFD_BZ2_KEEP_ON_DEVICE", "1") != "0"   # 0: decoded frames travel to the pinned slots and back
        self.bz2_device_min = int(os.environ.get("LFD_BZ2_DEVICE_MIN", 8))   # fewer compressed frames in a chunk than this: the host decodes them
        import threading
        self._lock = threading.Lock()
        self._bz2s = [None] * self.depth                     # a decoder per chunk that can be loading at once
        self._bz2_pins = [[None, None] for _ in range(self.depth)]   # per decoder: compressed bytes of its chunk, and of the one read ahead
        self._bz2_aheads = [None] * self.depth               # (paths, sizes, offsets, futures, buffer index) of a decoder's read-ahead
        self.bz2_stats = {"device_frames": 0, "host_frames": 0, "decode_s": 0.0, "read_s": 0.0, "fetch_s": 0.0}
        self._warm = [None] * self.depth
        if expect_bz2 and self.bz2_device and self.slots >= self.bz2_device_min:
            # the selection is known to be compressed: the decoders' tables (tens of GB each) are allocated while the first files are read
            def warm(lane):
                try:
                    bz = _native.Bz2Decoder(self.ctx.device)
                    blocks = self.slots * (self.frame_bytes // 900000 + 2)        # (900 kB blocks: the image's, the header's, the small HDUs')
                    bz.reserve(self.slots, blocks, self.bz2_out_cap, self.slots * self.frame_bytes)
                    self._bz2s[lane] = bz
                except _native.NativeError:
                    pass                                     # (the first decode tries again and reports)
            for lane in range(self.depth):
                self._warm[lane] = threading.Thread(target=warm, args=(lane,), daemon=True)
                self._warm[lane].start()

    def _ensure(self, which):
        """Chunk buffer ``which`` as (uint8 array, '>f4' [slots, h, w] view)."""
        if self.pins[which] is None:
            h, w = self.shape
            self.pins[which] = self.ctx.pinned_buffer(self.slots * self.frame_bytes)
            self.views[which] = self.pins[which].array.view(">f4").reshape(self.slots, h, w)
        return self.pins[which].array, self.views[which]

    def close(self):
        self.pool.shutdown(wait=True)
        self.block_pool.shutdown(wait=True)
        for lane in range(self.depth):
            if self._warm[lane] is not None:
                self._warm[lane].join()
                self._warm[lane] = None
            if self._bz2_aheads[lane] is not None:
                for f in self._bz2_aheads[lane][3]:
                    f.cancel() or f.exception()
                self._bz2_aheads[lane] = None
            if self._bz2s[lane] is not None:
                self._bz2s[lane].close()
                self._bz2s[lane] = None
            for k, p in enumerate(self._bz2_pins[lane]):
                if p is not None:
                    p.close()
                    self._bz2_pins[lane][k] = None
        self.views = None
        for p in self.pins:
            if p is not None:
                p.close()
        self.pins = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- exceptions to the fast path ---------------------------------------------------------------------------------
    def _fast(self, hdr):
        h, w = self.shape
        return (card_value(hdr, b"BITPIX") == -32 and card_value(hdr, b"NAXIS") == 2 and card_value(hdr, b"NAXIS1") == w
                and card_value(hdr, b"NAXIS2") == h and card_value(hdr, b"BSCALE") in (None, 1, 1.0)
                and card_value(hdr, b"BZERO") in (None, 0, 0.0))

    def _from_decompressed(self, out, i, slot, dst_u8, raw, path):
        """The decompressed bytes of a frame file: a plain float32 image goes into its pinned slot (True), anything else through
        the general reader into out.array[i]; raises what the file's defect raises."""
        end = header_end(raw)
        if end < 0 or len(raw) < end:
            raise ValueError("truncated FITS header")
        if self._fast(raw[:end]):
            if len(raw) < end + self.frame_bytes:
                raise ValueError(f"{path}: file ends inside the image")
            np.copyto(dst_u8, np.frombuffer(raw, np.uint8, self.frame_bytes, end))
            out.slot[i], out.hdr[i] = slot, raw[:end]
            return True
        img, h = fitslite.read_image_bytes(raw, path)
        out.array[i], out.hdr[i] = np.ascontiguousarray(img, dtype=np.float32), h
        return False

    def _slow_frame(self, out, i, slot, dst_u8, status):
        """Frame i was not taken by the native reader (status -1: no plain file; 1: not a plain float32 image; -2: broken)."""
        run, camcol, flt, field = out.keys[i]
        path = sdssfiles.filename("frame", run=run, camcol=camcol, field=field, filter=flt)
        try:
            if status == -1 and not os.path.exists(path):
                if not os.path.exists(path + ".bz2"):
                    raise FileNotFoundError(("File {0} or its bz2 compressed version not found. "
                                             "Are you sure they exist?").format(path))
                with open(path + ".bz2", "rb", buffering=0) as f:
                    # in memory; no $FITS_DUMP round trip (detecttrails.py:88-109)
                    raw = bz2blocks.decompress(f.read(), self.block_pool if self.split_blocks else None)
                if self._from_decompressed(out, i, slot, dst_u8, raw, path + ".bz2"):
                    return
                img, h = out.array[i], out.hdr[i]
            else:
                img, h = fitslite.read_image(path)           # (raises what the file's defect raises)
            out.array[i], out.hdr[i] = np.ascontiguousarray(img, dtype=np.float32), h
        except Exception as e:  # noqa: BLE001 - this frame's errors.txt entry (detecttrails.py:133-139)
            out.error[i] = e

    def _read_compressed(self, paths, lane, pin_k):
        """Starts reading ``paths`` into compressed-bytes buffer ``pin_k`` on the loader's pool: (sizes, offsets, futures)."""
        sizes = [os.path.getsize(p) if os.path.exists(p) else 0 for p in paths]
        offs, cur = [], 0
        for z in sizes:
            offs.append(cur)
            cur += (z + 255) & ~255
        pin = self._bz2_pins[lane][pin_k]
        if pin is None or pin.nbytes < cur:
            if pin is not None:
                pin.close()
            pin = self._bz2_pins[lane][pin_k] = self.ctx.pinned_buffer(max(cur + cur // 4, 1 << 20))
        src = pin.array

        def read(k):
            try:
                with open(paths[k], "rb", buffering=0) as f:
                    got = f.readinto(memoryview(src)[offs[k]:offs[k] + sizes[k]])
                return got == sizes[k]
            except OSError:                                  # (the host path meets the same error and reports it)
                return False
        return sizes, offs, [self.pool.submit(read, k) for k in range(len(paths))]

    def _device_bz2(self, out, todo, which, lane, devbuf, whole_chunk, next_paths=None):
        """``todo``: [(i, slot, path + '.bz2')] frames of this chunk that exist only compressed.  Decompresses them on the GPU and
        puts each image's data unit into its pinned slot; returns the entries that still have to go the host way."""
        import time
        t0 = time.perf_counter()
        paths = [p for _, _, p in todo]
        if self._warm[lane] is not None:
            self._warm[lane].join()
            self._warm[lane] = None
        if self._bz2s[lane] is None:
            self._bz2s[lane] = _native.Bz2Decoder(self.ctx.device)
        bz = self._bz2s[lane]
        ahead, self._bz2_aheads[lane] = self._bz2_aheads[lane], None
        if ahead is not None and ahead[0] == paths:           # read while the previous chunk was being decoded
            _, sizes, offs, futs, pin_k = ahead
            ok_read = [f.result() for f in futs]
        else:
            if ahead is not None:
                for f in ahead[3]:
                    f.result()
            pin_k = 0 if ahead is None else ahead[4] ^ 1
            sizes, offs, futs = self._read_compressed(paths, lane, pin_k)
            ok_read = [f.result() for f in futs]
        src = self._bz2_pins[lane][pin_k].array
        if next_paths and self.depth == 1:                    # the next chunk's files: into the other buffer, while this chunk is on the GPU
            s2, o2, f2 = self._read_compressed(next_paths, lane, pin_k ^ 1)           # (depth 2: the next chunk is being loaded beside this one anyway)
            self._bz2_aheads[lane] = (list(next_paths), s2, o2, f2, pin_k ^ 1)
        t1 = time.perf_counter()
        try:
            out_len, status, heads = bz.decode(src, offs, sizes, self.bz2_out_cap, HDR_CAP)
        except _native.NativeError as e:                     # e.g. no room for the decoder's tables on this device: the host's cores from now on
            import warnings
            warnings.warn(f"device bzip2 decoder switched off ({e}); .bz2 frames are decompressed on the host")
            self.bz2_device = False
            return list(todo)
        t2 = time.perf_counter()
        rest, files, foff, fbytes, slots_of = [], [], [], [], []
        whole = []
        for k, (i, slot, path) in enumerate(todo):
            if not ok_read[k] or status[k] != 0:
                rest.append(todo[k])
                continue
            hdr = heads[k].tobytes()
            end = header_end(hdr)
            if end < 0 or not self._fast(hdr[:end]) or int(out_len[k]) < end + self.frame_bytes:
                whole.append((k, i, slot, path))                # a long header, another pixel type, a short file: the general reader
                continue
            files.append(k); foff.append(end); fbytes.append(self.frame_bytes); slots_of.append(slot)
            out.slot[i], out.hdr[i] = slot, hdr[:end]
        # Every frame of the chunk decoded and plain: the data units are gathered in device memory and stay there (the GPU call
        # takes them as they are: compressed frames cross PCIe once, compressed).  Otherwise -- plain files in the chunk, a frame
        # for the host decoder or the general reader -- everything meets in the pinned slots as before.
        if whole_chunk and not rest and not whole and self.bz2_keep_on_device:
            h, w = self.shape
            dev = bz.frames(devbuf, self.slots, h, w)
            bz.fetch_many(files, foff, fbytes, [dev.address_of(sl) for sl in slots_of])
            out.device = dev
            src_of = {sl: (k, e) for sl, k, e in zip(slots_of, files, foff)}
            dec = bz

            def fetch(slot, _src=src_of, _dec=dec, _fb=self.frame_bytes, _shape=self.shape):   # (only inside load(): the next decode reuses the files)
                k, e = _src[slot]
                return _dec.fetch(k, e, _fb).view(">f4").reshape(_shape)
            out.fetch = fetch
        else:
            raw, out.buffer = self._ensure(which)
            bz.fetch_many(files, foff, fbytes, [raw[sl * self.frame_bytes:(sl + 1) * self.frame_bytes] for sl in slots_of])
        for k, i, slot, path in whole:
            try:
                raw, out.buffer = self._ensure(which)
                self._from_decompressed(out, i, slot, raw[slot * self.frame_bytes:(slot + 1) * self.frame_bytes],
                                        bz.fetch(k, 0, int(out_len[k])).tobytes(), path)
            except Exception as e:  # noqa: BLE001 - this frame's errors.txt entry (detecttrails.py:133-139)
                out.error[i] = e
        t3 = time.perf_counter()
        with self._lock:
            st = self.bz2_stats
            st["device_frames"] += len(todo) - len(rest)
            st["read_s"] += t1 - t0
            st["decode_s"] += t2 - t1
            st["fetch_s"] += t3 - t2
        return rest

    def _slow_catalog(self, out, i, slot, cats, status, path):
        try:
            if status == 2:
                raise ValueError("cannot convert float NaN to integer")        # math.ceil in removestars.py:113-130
            if status == 3:
                raise OverflowError("cannot convert float infinity to integer")
            from .removestars import read_photoObj_arrays
            cat = read_photoObj_arrays(path)                 # (missing file, missing columns, ...: raises as the reference's read would)
            m = len(cat["NOBSERVE"])
            if m <= self.max_obj and out.slot[i] >= 0:
                for k in _CAT5 + _CAT1:
                    cats[k][slot, :m] = cat[k]
                cats["count"][slot] = m
            else:                                            # more rows than the padded arrays hold: the per-frame path
                out.cat[i] = {k: np.asarray(cat[k]) for k in _CAT5 + _CAT1}
                if out.slot[i] >= 0:
                    out.array[i] = out.frame_host(out.slot[i]).astype(np.float32)
                    out.slot[i] = -1
        except Exception as e:  # noqa: BLE001
            out.error[i] = e
            out.slot[i] = -1
            out.array[i] = None

    def _compressed_only(self, keys):
        """The .bz2 paths of ``keys`` (the chunk after this one) if every frame of it exists only compressed, else None."""
        if not keys or not self.bz2_device or len(keys) < self.bz2_device_min:
            return None
        paths = []
        for i in sorted(range(len(keys)), key=lambda i: keys[i][2]):     # (the order load() gives the slots in)
            run, camcol, flt, field = keys[i]
            p = sdssfiles.filename("frame", run=run, camcol=camcol, field=field, filter=flt)
            if os.path.exists(p) or not os.path.exists(p + ".bz2"):
                return None
            paths.append(p + ".bz2")
        return paths

    # -- a chunk --------------------------------------------------------------------------------------------------
    def load(self, keys, which, next_keys=None, seq=None):
        """Read ``keys`` (at most ``slots``) into pinned buffer ``which`` (0 / 1).  Frames of one filter get neighbouring slots
        (remove_stars' magnitude cap depends on the filter, so a GPU call takes one filter's frames: a contiguous slice).
        ``next_keys``: the chunk that will be asked for next; if it exists only as .fits.bz2 its files are read ahead while
        this chunk is being decompressed."""
        n = len(keys)
        if n > self.slots:
            raise ValueError("chunk larger than the loader's buffers")
        seq = which if seq is None else int(seq)              # the chunk's number: which decoder, which of its device buffers
        lane, devbuf = seq % self.depth, (seq // self.depth) % 2
        out = Loaded(keys)
        out.shape = self.shape
        out.cats = cats = self.cats[which]
        hdrs = self.hdrs[which]
        order = sorted(range(n), key=lambda i: keys[i][2])          # stable: the caller's order inside a filter
        fpaths, ppaths = [], []
        for i in order:
            run, camcol, flt, field = keys[i]
            fpaths.append(sdssfiles.filename("frame", run=run, camcol=camcol, field=field, filter=flt))
            ppaths.append(sdssfiles.filename("photoObj", run=run, camcol=camcol, field=field))
        h, w = self.shape
        fstat = np.zeros(n, np.int32)
        hlen = np.zeros(n, np.int32)
        pstat = np.zeros(n, np.int32)
        P = _native._ptr
        if self._compressed_only(keys) is not None:
            fstat[:] = -1                                     # no plain file among them: nothing for the native reader, and no chunk buffer yet
        else:
            raw, out.buffer = self._ensure(which)
            rc = self.lib.lfdmi_fits_read_frames(_paths(fpaths), n, h, w, P(raw), self.threads, P(fstat), P(hdrs), HDR_CAP, P(hlen))
            if rc:
                raise _native.NativeError(rc, "lfdmi_fits_read_frames")
        rc = self.lib.lfdmi_fits_read_photoobj(_paths(ppaths), n, self.max_obj, P(cats["ROWC"]), P(cats["COLC"]), P(cats["PSFMAG"]),
                                               P(cats["PETROTH90"]), P(cats["NOBSERVE"]), P(cats["NDETECT"]), P(cats["count"]),
                                               self.threads, P(pstat))
        if rc:
            raise _native.NativeError(rc, "lfdmi_fits_read_photoobj")
        futs = []
        on_device = set()
        if self.bz2_device:
            todo = [(i, slot, fpaths[slot] + ".bz2") for slot, i in enumerate(order)
                    if int(fstat[slot]) == -1 and not os.path.exists(fpaths[slot]) and os.path.exists(fpaths[slot] + ".bz2")]
            if len(todo) >= self.bz2_device_min:             # (a handful of files: their blocks side by side on the host's cores are quicker)
                rest = self._device_bz2(out, todo, which, lane, devbuf, len(todo) == n, self._compressed_only(next_keys))
                on_device = {i for i, _, _ in todo} - {i for i, _, _ in rest}
                self.bz2_stats["host_frames"] += len(rest)
        self.split_blocks = int((fstat != 0).sum()) - len(on_device) < self.threads   # (few files for many cores: their blocks side by side)
        for slot, i in enumerate(order):
            st = int(fstat[slot])
            if i in on_device:
                continue
            if st == 0:
                out.slot[i] = slot
                if hlen[slot] <= HDR_CAP:
                    out.hdr[i] = hdrs[slot, :hlen[slot]].tobytes()
                else:                                        # an unusually long header: read it again, whole
                    with open(fpaths[slot], "rb") as f:
                        out.hdr[i] = f.read(int(hlen[slot]))
            else:
                raw, out.buffer = self._ensure(which)
                dst = raw[slot * self.frame_bytes:(slot + 1) * self.frame_bytes]
                futs.append(self.pool.submit(self._slow_frame, out, i, slot, dst, st))
        for f in futs:
            f.result()
        for slot, i in enumerate(order):                     # catalogues (after the frames: a frame error comes first, as in the reference)
            if out.error[i] is None and pstat[slot] != 0:
                self._slow_catalog(out, i, slot, cats, int(pstat[slot]), ppaths[slot])
            if out.error[i] is not None:
                out.slot[i] = -1
                out.array[i] = None
        return out


def read_catalog(path, max_obj=MAX_OBJ):
    """One photoObj file through the native reader: dict of the six columns, or None where it declines (tests, tools)."""
    lib = _native.lib()
    c = {k: np.zeros((1, max_obj, 5), np.float32) for k in _CAT5}
    c.update({k: np.zeros((1, max_obj), np.int32) for k in _CAT1})
    cnt, st = np.zeros(1, np.int32), np.zeros(1, np.int32)
    P = _native._ptr
    rc = lib.lfdmi_fits_read_photoobj(_paths([str(path)]), 1, max_obj, P(c["ROWC"]), P(c["COLC"]), P(c["PSFMAG"]), P(c["PETROTH90"]),
                                      P(c["NOBSERVE"]), P(c["NDETECT"]), P(cnt), 1, P(st))
    if rc or st[0] not in (0, 2, 3):
        return None
    return {k: c[k][0, :cnt[0]].copy() for k in _CAT5 + _CAT1}
