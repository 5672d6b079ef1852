"""ctypes binding of liblfdmi.so (include/lfdmi.h): the only way Python reaches the GPU path.

There is deliberately no CPU fallback: if the shared library is missing or no HIP device is
usable, every entry point raises.  numpy arrays are passed as host pointers (the library
stages them); objects exposing ``data_ptr()`` (torch CUDA tensors) are passed as device
pointers and used in place.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LFDMI_LIB") or os.path.join(_HERE, "csrc", "liblfdmi.so")   # ($LFDMI_LIB: a developer's variant build)

HOST, DEVICE, HOST_PINNED = 0, 1, 2
U8, F32, F64, F32_BE = 0, 1, 2, 3
PREP_NONE, PREP_BRIGHT, PREP_DIM, PREP_BRIGHT_THEN_DIM = 0, 1, 2, 3
STAGE_GRAY, STAGE_EQU, STAGE_CANNY, STAGE_BOX, STAGE_ERODED, STAGE_EQUALIZED = 0, 1, 2, 3, 4, 5
MAX_SCALES = 4
MAX_SET_LINES = 64
MAX_MORPH_K = 31

ERR_ARG, ERR_DTYPE, ERR_HIP, ERR_UNSUPPORTED, ERR_CAPACITY, ERR_NOLINES = -1, -2, -3, -4, -5, -6


# every symbol include/lfdmi.h declares (checked by the CPU test-suite)
SYMBOLS = (
    "lfdmi_version", "lfdmi_default_caps", "lfdmi_ctx_create", "lfdmi_ctx_create_sized", "lfdmi_ctx_bytes", "lfdmi_spill_count", "lfdmi_get_stats",
    "lfdmi_ctx_destroy", "lfdmi_last_error", "lfdmi_set_stream", "lfdmi_process_multiscale", "lfdmi_debug_frame_profile", "lfdmi_debug_tail", "lfdmi_debug_trig", "lfdmi_debug_fail_chunk",
    "lfdmi_max_inflight", "lfdmi_prep_u8", "lfdmi_equalize_hist", "lfdmi_dilate", "lfdmi_erode",
    "lfdmi_canny", "lfdmi_gaussian_blur", "lfdmi_fit_min_area_rect", "lfdmi_hough_lines", "lfdmi_hough_accum",
    "lfdmi_hough_dims", "lfdmi_remove_stars", "lfdmi_process_bright", "lfdmi_process_dim",
    "lfdmi_detect_batch", "lfdmi_detect_batch_raw", "lfdmi_host_alloc", "lfdmi_host_free", "lfdmi_fits_read_frames", "lfdmi_fits_read_photoobj", "lfdmi_bz2_find_blocks", "lfdmi_bz2_create", "lfdmi_bz2_destroy", "lfdmi_bz2_last_error", "lfdmi_bz2_decode_batch", "lfdmi_bz2_fetch", "lfdmi_bz2_fetch_many", "lfdmi_bz2_frames", "lfdmi_bz2_reserve", "lfdmi_bz2_timings", "lfdmi_set_stage_images", "lfdmi_get_stage", "lfdmi_get_counters", "lfdmi_enable_timing", "lfdmi_timing_select", "lfdmi_get_timing",
    "lfdmi_timing_slots", "lfdmi_timing_name",
)


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"liblfdmi error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [("lwTresh", C.c_double), ("thetaTresh", C.c_double), ("lineSetTresh", C.c_double),
                ("dro", C.c_double), ("minAreaRectMinLen", C.c_double), ("houghMethod", C.c_double),
                ("nlinesInSet", C.c_int32), ("contoursMode", C.c_int32), ("contoursMethod", C.c_int32),
                ("dilate_kh", C.c_int32), ("dilate_kw", C.c_int32), ("dilateKernel", C.c_void_p),
                ("erode_kh", C.c_int32), ("erode_kw", C.c_int32), ("erodeKernel", C.c_void_p),
                ("minFlux", C.c_double), ("addFlux", C.c_double),
                ("gaussKernel", C.c_int32), ("gaussSigma", C.c_double)]


class Caps(C.Structure):
    """lfdmi_caps: per-frame table capacities of a workspace (<= 0: the theoretical maximum)."""
    _fields_ = [("run_cap", C.c_int32), ("key_cap", C.c_int32), ("slot_cap", C.c_int32), ("list_cap", C.c_int32),
                ("peak_cap", C.c_int32), ("min_rho", C.c_double)]


class RsParams(C.Structure):
    _fields_ = [("defaultxy", C.c_int32), ("maxxy", C.c_int32), ("magcount", C.c_int32),
                ("pixscale", C.c_double), ("maxmagdiff", C.c_double), ("filter_cap", C.c_double),
                ("filter_index", C.c_int32)]


class Catalog(C.Structure):
    _fields_ = [("max_obj", C.c_int32), ("count", C.c_void_p), ("rowc", C.c_void_p),
                ("colc", C.c_void_p), ("psfmag", C.c_void_p), ("petro90", C.c_void_p),
                ("nobserve", C.c_void_p), ("ndetect", C.c_void_p), ("loc", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("found", C.c_int32), ("rho", C.c_float), ("theta", C.c_float),
                ("x1", C.c_int32), ("y1", C.c_int32), ("x2", C.c_int32), ("y2", C.c_int32),
                ("n_lines_equ", C.c_int32), ("n_lines_box", C.c_int32), ("detection", C.c_int32),
                ("rejected_by_theta", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


RESULT_DTYPE = np.dtype([("status", "<i4"), ("found", "<i4"), ("rho", "<f4"), ("theta", "<f4"),
                         ("x1", "<i4"), ("y1", "<i4"), ("x2", "<i4"), ("y2", "<i4"),
                         ("n_lines_equ", "<i4"), ("n_lines_box", "<i4"), ("detection", "<i4"),
                         ("rejected_by_theta", "<i4")])

_lib = None


def lib():
    """Load liblfdmi.so or fail loudly (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C lfd_amd/csrc` "
                              "(or `python -c 'import __graft_entry__ as g; g.build()'`)")
        _lib = C.CDLL(LIB_PATH)
        _lib.lfdmi_last_error.restype = C.c_char_p
        _lib.lfdmi_last_error.argtypes = [C.c_void_p]
        _lib.lfdmi_ctx_destroy.restype = None
        _lib.lfdmi_ctx_destroy.argtypes = [C.c_void_p]
        _lib.lfdmi_hough_dims.restype = None
        _lib.lfdmi_default_caps.restype = None
        _lib.lfdmi_ctx_bytes.restype = C.c_int64
        _lib.lfdmi_ctx_bytes.argtypes = [C.c_void_p]
        _lib.lfdmi_spill_count.restype = C.c_int64
        _lib.lfdmi_spill_count.argtypes = [C.c_void_p]
        _lib.lfdmi_timing_name.restype = C.c_char_p
        _lib.lfdmi_bz2_find_blocks.restype = C.c_int64
        _lib.lfdmi_bz2_find_blocks.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64]
        _lib.lfdmi_bz2_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        _lib.lfdmi_bz2_destroy.restype = None
        _lib.lfdmi_bz2_destroy.argtypes = [C.c_void_p]
        _lib.lfdmi_bz2_last_error.restype = C.c_char_p
        _lib.lfdmi_bz2_last_error.argtypes = [C.c_void_p]
        _lib.lfdmi_bz2_decode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_uint64,
                                                C.c_void_p, C.c_void_p]
        _lib.lfdmi_bz2_fetch.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]
        _lib.lfdmi_bz2_fetch_many.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.lfdmi_bz2_timings.argtypes = [C.c_void_p, C.c_void_p]
        _lib.lfdmi_bz2_frames.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]
        _lib.lfdmi_bz2_reserve.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_uint64]
    return _lib


def _is_dev(a):
    return hasattr(a, "data_ptr")


def _ptr(a):
    if a is None:
        return None
    if _is_dev(a):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)


def _dtype_code(a):
    name = str(a.dtype).replace("torch.", "")
    try:
        return {"uint8": U8, "float32": F32, "float64": F64}[name]
    except KeyError:
        raise TypeError(f"unsupported image dtype {a.dtype}") from None


def make_params(d, dim=False):
    """dict with the reference's key names -> (Params struct, objects to keep alive)."""
    dk = np.ascontiguousarray(d["dilateKernel"], np.uint8)
    if dk.ndim != 2:
        raise ValueError("dilateKernel must be a 2-d array")
    keep = [dk]
    p = Params()
    p.lwTresh = float(d["lwTresh"])
    p.thetaTresh = float(d["thetaTresh"])
    p.lineSetTresh = float(d["lineSetTresh"])
    p.dro = float(d["dro"])
    p.minAreaRectMinLen = float(d["minAreaRectMinLen"])
    p.houghMethod = float(d["houghMethod"])
    p.nlinesInSet = int(d["nlinesInSet"])
    p.contoursMode = int(d["contoursMode"])
    p.contoursMethod = int(d["contoursMethod"])
    p.dilate_kh, p.dilate_kw = dk.shape
    p.dilateKernel = dk.ctypes.data
    if dim:
        ek = np.ascontiguousarray(d["erodeKernel"], np.uint8)
        if ek.ndim != 2:
            raise ValueError("erodeKernel must be a 2-d array")
        keep.append(ek)
        p.erode_kh, p.erode_kw = ek.shape
        p.erodeKernel = ek.ctypes.data
        p.minFlux = float(d["minFlux"])
        p.addFlux = float(d["addFlux"])
    p.gaussKernel = int(d.get("gaussKernel", 0) or 0)       # optional smoothing of Canny's input; off in the reference
    p.gaussSigma = float(d.get("gaussSigma", 0.0) or 0.0)
    return p, keep


def make_rs_params(filter, defaultxy, filter_caps, maxxy, pixscale, magcount, maxmagdiff, **_):
    return RsParams(int(defaultxy), int(maxxy), int(magcount), float(pixscale), float(maxmagdiff),
                    float(filter_caps[filter]), "ugriz".index(filter))


class PinnedBuffer:
    """``nbytes`` of page-locked host memory from lfdmi_host_alloc; ``.array`` is a uint8 numpy view (take ``.view('>f4')``
    slices of it for frames).  Freed by ``close()`` / garbage collection; the views must not be used afterwards."""

    def __init__(self, ctx, nbytes):
        self._lib = ctx._lib
        self._p = C.c_void_p()
        self.nbytes = int(nbytes)
        self._lib.lfdmi_host_alloc.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        ctx._chk(self._lib.lfdmi_host_alloc(ctx._h, C.c_uint64(self.nbytes), C.byref(self._p)))
        self.array = np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(self._p.value))

    def close(self):
        p, self._p = getattr(self, "_p", None), None
        if p:
            self.array = None
            self._lib.lfdmi_host_free.argtypes = [C.c_void_p, C.c_void_p]
            self._lib.lfdmi_host_free(None, p)

    __del__ = close


BZ2_STATUS = {0: "ok", 1: "no block magic", 2: "randomised block", 3: "bad block header", 4: "bad compressed data",
              5: "block length mismatch", 6: "bad origin pointer", 7: "CRC mismatch", 8: "BWT cycle shorter than the block",
              9: "larger than out_cap", 10: "not a sequence of bzip2 streams"}


class DeviceFrames:
    """n big-endian float32 frames (the data units of FITS images) in device memory that belongs to a ``Bz2Decoder``: what
    ``Context.detect_batch`` takes instead of an array when the frames were decompressed on the GPU and never left it."""

    def __init__(self, ptr, shape):
        self._ptr = int(ptr)
        self.shape = tuple(int(x) for x in shape)            # (n, h, w)

    def data_ptr(self):
        return self._ptr

    @property
    def frame_bytes(self):
        return self.shape[1] * self.shape[2] * 4

    def address_of(self, k):
        return self._ptr + int(k) * self.frame_bytes

    def slice(self, a, b):
        return DeviceFrames(self.address_of(a), (int(b) - int(a), self.shape[1], self.shape[2]))


class Bz2Decoder:
    """lfdmi_bz2_*: whole ``.bz2`` files decompressed on the GPU, many at once (include/lfdmi.h).  ``decode`` takes the
    compressed files as one uint8 array + offsets / lengths and returns (out_len, status, heads); the decompressed bytes stay on
    the device until the next ``decode`` and are copied out in ranges by ``fetch`` / ``fetch_many``.  A file whose status is not
    0 was NOT decoded (a broken block, trailing bytes, ...): decompress it on the host, as the reference does."""

    def __init__(self, device=0):
        self._lib = lib()
        self._h = C.c_void_p()
        rc = self._lib.lfdmi_bz2_create(int(device), C.byref(self._h))
        if rc:
            raise NativeError(rc, "lfdmi_bz2_create")

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.lfdmi_bz2_destroy(h)

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc:
            raise NativeError(rc, (self._lib.lfdmi_bz2_last_error(self._h) or b"").decode())

    def decode(self, src, offsets, lengths, out_cap, head_bytes=0):
        src = np.ascontiguousarray(src).view(np.uint8).reshape(-1)
        off = np.ascontiguousarray(offsets, np.uint64)
        ln = np.ascontiguousarray(lengths, np.uint64)
        n = len(off)
        if n == 0 or len(ln) != n or (n and int((off + ln).max()) > src.size):
            raise ValueError("Bz2Decoder.decode: offsets / lengths do not fit the source array")
        out_len = np.zeros(n, np.uint64)
        status = np.zeros(n, np.int32)
        heads = np.zeros((n, int(head_bytes)), np.uint8) if head_bytes else None
        self._chk(self._lib.lfdmi_bz2_decode_batch(self._h, _ptr(src), _ptr(off), _ptr(ln), n, C.c_uint64(int(out_cap)),
                                                   _ptr(heads), C.c_uint64(int(head_bytes)), _ptr(out_len), _ptr(status)))
        return out_len, status, heads

    def fetch(self, i, offset, nbytes, dst=None):
        """Bytes [offset, offset + nbytes) of decompressed file i into ``dst`` (a uint8 numpy array / torch CUDA tensor; a new
        numpy array if None)."""
        if dst is None:
            dst = np.empty(int(nbytes), np.uint8)
        self._chk(self._lib.lfdmi_bz2_fetch(self._h, int(i), C.c_uint64(int(offset)), C.c_uint64(int(nbytes)), _ptr(dst),
                                            DEVICE if _is_dev(dst) else HOST))
        return dst

    def fetch_many(self, files, offsets, nbytes, dsts):
        """One range per entry, all copies queued before one wait.  dsts: numpy arrays, torch CUDA tensors, or plain integers =
        device addresses (``DeviceFrames.address_of``), all of one kind."""
        n = len(files)
        if n == 0:
            return
        f = np.ascontiguousarray(files, np.int32)
        o = np.ascontiguousarray(offsets, np.uint64)
        b = np.ascontiguousarray(nbytes, np.uint64)
        on_dev = isinstance(dsts[0], int) or _is_dev(dsts[0])
        ptrs = (C.c_void_p * n)(*[C.c_void_p(d) if isinstance(d, int) else _ptr(d) for d in dsts])
        self._chk(self._lib.lfdmi_bz2_fetch_many(self._h, n, _ptr(f), _ptr(o), _ptr(b), ptrs, DEVICE if on_dev else HOST))

    def frames(self, which, n, h, w):
        """The handle's device buffer ``which`` (0 / 1), at least n frames of h x w float32 large, as ``DeviceFrames``."""
        p = C.c_void_p()
        self._chk(self._lib.lfdmi_bz2_frames(self._h, int(which), C.c_uint64(int(n) * int(h) * int(w) * 4), C.byref(p)))
        return DeviceFrames(p.value, (n, h, w))

    def reserve(self, n_files, n_blocks, out_cap, compressed_bytes=0):
        """Allocate the decoder's tables now (optional; otherwise inside the first ``decode``)."""
        self._chk(self._lib.lfdmi_bz2_reserve(self._h, int(n_files), int(n_blocks), C.c_uint64(int(out_cap)), C.c_uint64(int(compressed_bytes))))

    def timings(self):
        ms = np.zeros(5, np.float32)
        self._chk(self._lib.lfdmi_bz2_timings(self._h, _ptr(ms)))
        return dict(zip(("upload+magics", "huffman+mtf", "sort", "walk", "rle+crc+out"), (float(x) for x in ms)))


class Context:
    """One GPU, one HIP stream, workspace for ``max_inflight`` frames of up to max_h x max_w."""

    def __init__(self, device=0, max_h=1489, max_w=2048, max_inflight=8, caps=None):
        """caps: None = the library's default table capacities (frames that need more are re-run through a
        worst-case workspace inside the library); "worst" = every table at its theoretical maximum; or a dict
        with any of run_cap / key_cap / slot_cap / list_cap / peak_cap / min_rho (others keep their defaults)."""
        self._h = C.c_void_p()
        self._lib = lib()
        if caps is None:
            rc = self._lib.lfdmi_ctx_create(int(device), int(max_h), int(max_w), int(max_inflight),
                                            C.byref(self._h))
        else:
            c = Caps()
            if caps != "worst":
                self._lib.lfdmi_default_caps(int(max_h), int(max_w), C.byref(c))
                for k, v in dict(caps).items():
                    setattr(c, k, v)
            rc = self._lib.lfdmi_ctx_create_sized(int(device), int(max_h), int(max_w), int(max_inflight),
                                                  C.byref(c), C.byref(self._h))
        if rc:
            msg = self._lib.lfdmi_last_error(self._h).decode() if self._h else "context creation failed"
            h, self._h = self._h, C.c_void_p()
            if h:
                self._lib.lfdmi_ctx_destroy(h)
            raise NativeError(rc, msg)
        self.device, self.max_h, self.max_w, self.max_inflight = device, max_h, max_w, max_inflight

    def close(self):
        h = getattr(self, "_h", None)
        if h:
            self._h = None
            self._lib.lfdmi_ctx_destroy(h)

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc:
            raise NativeError(rc, self._lib.lfdmi_last_error(self._h).decode())

    def workspace_bytes(self):
        """Device bytes of the workspace (worst-case spill workspace included once it exists)."""
        return int(self._lib.lfdmi_ctx_bytes(self._h))

    def spill_count(self):
        """Frames this context has re-run through its worst-case workspace."""
        return int(self._lib.lfdmi_spill_count(self._h))

    def set_stream(self, stream_handle):
        self._chk(self._lib.lfdmi_set_stream(self._h, C.c_void_p(stream_handle or 0)))

    STAT_NAMES = ("spilled_frames", "scan_giveups", "general_reruns", "general_chunks", "chunks", "cap_growths", "scan_fused_on")

    def stats(self):
        """lfdmi_get_stats as a dict: what the context did besides the fast path (never changes a result, always costs time)."""
        out = np.zeros(len(self.STAT_NAMES), np.int64)
        self._chk(self._lib.lfdmi_get_stats(self._h, _ptr(out), len(out)))
        return dict(zip(self.STAT_NAMES, (int(v) for v in out)))

    def debug_trig(self, y, x):
        """(angle_deg, cos/2, sin/2) float32 as the rectangle kernels compute them from atan2(y, x) (developer check)."""
        y = np.ascontiguousarray(y, np.float64)
        x = np.ascontiguousarray(x, np.float64)
        n = y.size
        out = [np.zeros(n, np.float32) for _ in range(3)]
        self._chk(self._lib.lfdmi_debug_trig(self._h, n, _ptr(y), _ptr(x), *[_ptr(o) for o in out]))
        return out

    def debug_tail(self, h1, n1, h2, n2, navg, dro, thetaTresh, lineSetTresh, which, shape):
        """The tail of a pass on line sets given from outside: the device's check_theta (k_finalize) and the library's host-side
        dictify_hough, as detect_batch runs them.  h1 / h2: (n, kmax, 2) float32, n1 / n2: lines per set -> record array."""
        h1 = np.ascontiguousarray(h1, np.float32)
        h2 = np.ascontiguousarray(h2, np.float32)
        n1 = np.ascontiguousarray(n1, np.int32)
        n2 = np.ascontiguousarray(n2, np.int32)
        n, kmax = h1.shape[0], h1.shape[1]
        assert h2.shape == h1.shape and n1.shape == (n,) and n2.shape == (n,)
        out = np.zeros(n, RESULT_DTYPE)
        self._chk(self._lib.lfdmi_debug_tail(self._h, n, kmax, _ptr(h1), _ptr(n1), _ptr(h2), _ptr(n2), int(navg), C.c_double(dro),
                                             C.c_double(thetaTresh), C.c_double(lineSetTresh), int(which), int(shape[0]), int(shape[1]),
                                             _ptr(out)))
        return out

    def debug_fail_chunk(self, chunk):
        """Developer hook: the next detect_batch fails (ERR_ARG) at the top of chunk ``chunk`` (tests of the error path)."""
        self._chk(self._lib.lfdmi_debug_fail_chunk(self._h, int(chunk)))

    def set_stage_images(self, mode):
        """Which calls keep the 8-bit stage images for get_stage: -1 the per-pass calls do and detect_batch does not
        (default), 0 none (batches), 1 all."""
        self._chk(self._lib.lfdmi_set_stage_images(self._h, int(mode)))

    def enable_timing(self, on=True):
        self._chk(self._lib.lfdmi_enable_timing(self._h, int(on)))

    def timing_select(self, names=None):
        """Bracket only the launches of the given timing slots (kernel names as in get_timing), or all (None)."""
        mask = 0
        if names:
            known = [self._lib.lfdmi_timing_name(i).decode() for i in range(self._lib.lfdmi_timing_slots())]
            for nm in ([names] if isinstance(names, str) else names):
                mask |= 1 << known.index(nm)
        self._lib.lfdmi_timing_select.argtypes = [C.c_void_p, C.c_uint64]
        self._chk(self._lib.lfdmi_timing_select(self._h, mask))

    def get_timing(self):
        """{kernel name: (total device ms, launches, frames worked on)} since enable_timing(True)."""
        k = self._lib.lfdmi_timing_slots()
        ms = (C.c_float * k)()
        n = (C.c_int32 * k)()
        u = (C.c_int64 * k)()
        self._chk(self._lib.lfdmi_get_timing(self._h, ms, n, u))
        return {self._lib.lfdmi_timing_name(i).decode(): (float(ms[i]), int(n[i]), int(u[i])) for i in range(k)}

    # -- helpers --------------------------------------------------------------------------
    @staticmethod
    def _batch(img, ndim_item=2):
        """-> (array with leading batch axis, n, h, w, squeeze?)"""
        if _is_dev(img):
            shp = tuple(img.shape)
            if not img.is_contiguous():
                raise ValueError("device tensors must be contiguous")
        else:
            img = np.ascontiguousarray(img)
            shp = img.shape
        if len(shp) == ndim_item:
            return img, 1, shp[0], shp[1], True
        if len(shp) == ndim_item + 1:
            return img, shp[0], shp[1], shp[2], False
        raise ValueError(f"expected a 2-d image or a 3-d batch, got shape {shp}")

    @staticmethod
    def _out_like(img, n, h, w, squeeze, dtype=np.uint8):
        if _is_dev(img):
            import torch
            tdt = {np.uint8: torch.uint8, np.int32: torch.int32, np.float32: torch.float32}[dtype]
            out = torch.empty((n, h, w), dtype=tdt, device=img.device)
        else:
            out = np.empty((n, h, w), dtype)
        return out

    # -- per-operator entry points ------------------------------------------------------------
    def prep_u8(self, img, mode, flip=False, minFlux=0.0, addFlux=0.0, want_hist=False):
        img, n, h, w, sq = self._batch(img)
        loc = DEVICE if _is_dev(img) else HOST
        out = self._out_like(img, n, h, w, sq)
        hist = None
        if want_hist:
            if loc == DEVICE:
                import torch
                hist = torch.empty((n, 256), dtype=torch.int32, device=img.device)
            else:
                hist = np.empty((n, 256), np.int32)
        self._chk(self._lib.lfdmi_prep_u8(self._h, _ptr(img), _dtype_code(img), n, h, w, int(flip),
                                          int(mode), C.c_double(minFlux), C.c_double(addFlux),
                                          _ptr(out), _ptr(hist), loc))
        out = out[0] if sq else out
        if want_hist:
            return out, (hist[0] if sq else hist)
        return out

    def equalize_hist(self, img):
        img, n, h, w, sq = self._batch(img)
        loc = DEVICE if _is_dev(img) else HOST
        out = self._out_like(img, n, h, w, sq)
        self._chk(self._lib.lfdmi_equalize_hist(self._h, _ptr(img), n, h, w, _ptr(out), loc))
        return out[0] if sq else out

    def _morph(self, fn, img, kernel):
        img, n, h, w, sq = self._batch(img)
        loc = DEVICE if _is_dev(img) else HOST
        k = np.ascontiguousarray(kernel, np.uint8)
        if k.ndim != 2:
            raise ValueError("kernel must be 2-d")
        out = self._out_like(img, n, h, w, sq)
        self._chk(fn(self._h, _ptr(img), n, h, w, _ptr(k), k.shape[0], k.shape[1], _ptr(out), loc))
        return out[0] if sq else out

    def dilate(self, img, kernel):
        return self._morph(self._lib.lfdmi_dilate, img, kernel)

    def erode(self, img, kernel):
        return self._morph(self._lib.lfdmi_erode, img, kernel)

    def gaussian_blur(self, img, ksize, sigma=0.0):
        img, n, h, w, sq = self._batch(img)
        loc = DEVICE if _is_dev(img) else HOST
        out = self._out_like(img, n, h, w, sq)
        self._chk(self._lib.lfdmi_gaussian_blur(self._h, _ptr(img), n, h, w, int(ksize), C.c_double(sigma), _ptr(out), loc))
        return out[0] if sq else out

    def canny(self, img, low=0.0, high=255.0):
        img, n, h, w, sq = self._batch(img)
        loc = DEVICE if _is_dev(img) else HOST
        out = self._out_like(img, n, h, w, sq)
        self._chk(self._lib.lfdmi_canny(self._h, _ptr(img), n, h, w, C.c_double(low), C.c_double(high),
                                        _ptr(out), loc))
        return out[0] if sq else out

    def fit_min_area_rect(self, img, contoursMode=1, contoursMethod=1, minAreaRectMinLen=1, lwTresh=5,
                          want_box=True):
        """-> (detection bool(s), box_img(s), n_boxes)"""
        img, n, h, w, sq = self._batch(img)
        if _is_dev(img):
            raise TypeError("fit_min_area_rect: pass numpy arrays (device outputs not wired in Python)")
        box = np.empty((n, h, w), np.uint8) if want_box else None
        det = np.zeros(n, np.int32)
        nb = np.zeros(n, np.int32)
        self._chk(self._lib.lfdmi_fit_min_area_rect(self._h, _ptr(img), n, h, w, int(contoursMode),
                                                    int(contoursMethod), C.c_double(minAreaRectMinLen),
                                                    C.c_double(lwTresh), _ptr(box), _ptr(det), _ptr(nb),
                                                    HOST))
        if sq:
            return bool(det[0]), (box[0] if want_box else None), int(nb[0])
        return det.astype(bool), box, nb

    def hough_dims(self, h, w, rho, theta=np.pi / 180):
        na, nr = C.c_int(), C.c_int()
        self._lib.lfdmi_hough_dims(int(h), int(w), C.c_double(rho), C.c_double(theta), C.byref(na), C.byref(nr))
        return na.value, nr.value

    def hough_lines(self, img, rho, theta=np.pi / 180, threshold=1, max_lines=None):
        """cv2.HoughLines layout per image: (n_lines, 1, 2) float32 or None; also total count."""
        img, n, h, w, sq = self._batch(img)
        if _is_dev(img):
            raise TypeError("hough_lines: pass numpy arrays")
        na, nr = self.hough_dims(h, w, rho, theta)
        cap = na * nr if max_lines is None else int(max_lines)
        lines = np.zeros((n, max(cap, 1), 2), np.float32)
        cnt = np.zeros(n, np.int32)
        self._chk(self._lib.lfdmi_hough_lines(self._h, _ptr(img), n, h, w, C.c_double(rho), C.c_double(theta),
                                              int(threshold), cap, _ptr(lines), _ptr(cnt), HOST))
        outs = []
        for i in range(n):
            k = min(int(cnt[i]), cap)
            outs.append(lines[i, :k].reshape(k, 1, 2).copy() if k else None)
        return (outs[0], int(cnt[0])) if sq else (outs, cnt)

    def hough_accum(self, img, rho, theta=np.pi / 180):
        img, n, h, w, sq = self._batch(img)
        na, nr = self.hough_dims(h, w, rho, theta)
        acc = np.zeros((n, na + 2, nr + 2), np.int32)
        self._chk(self._lib.lfdmi_hough_accum(self._h, _ptr(img), n, h, w, C.c_double(rho), C.c_double(theta),
                                              _ptr(acc), HOST))
        return acc[0] if sq else acc

    @staticmethod
    def _catalog(cat):
        """dict of stacked arrays (see synth.pack_catalogs) -> (Catalog struct, keep-alive list)"""
        if cat is None:
            return None, []
        dev = _is_dev(cat["ROWC"])
        if dev:
            arrs = {k: cat[k] for k in ("count", "ROWC", "COLC", "PSFMAG", "PETROTH90", "NOBSERVE", "NDETECT")}
        else:
            arrs = {"count": np.ascontiguousarray(cat["count"], np.int32)}
            for k in ("ROWC", "COLC", "PSFMAG", "PETROTH90"):
                arrs[k] = np.ascontiguousarray(cat[k], np.float32)
            for k in ("NOBSERVE", "NDETECT"):
                arrs[k] = np.ascontiguousarray(cat[k], np.int32)
        c = Catalog()
        c.max_obj = int(arrs["NOBSERVE"].shape[1])
        c.count = _ptr(arrs["count"]).value
        c.rowc = _ptr(arrs["ROWC"]).value
        c.colc = _ptr(arrs["COLC"]).value
        c.psfmag = _ptr(arrs["PSFMAG"]).value
        c.petro90 = _ptr(arrs["PETROTH90"]).value
        c.nobserve = _ptr(arrs["NOBSERVE"]).value
        c.ndetect = _ptr(arrs["NDETECT"]).value
        c.loc = DEVICE if dev else HOST
        return c, list(arrs.values())

    def remove_stars(self, img, cat, rs):
        """float32 frame(s), mutated in place like removestars.py:231."""
        b, n, h, w, sq = self._batch(img)
        if _dtype_code(b) != F32:
            raise TypeError("remove_stars needs float32 frames")
        if not _is_dev(img) and b is not img and not np.shares_memory(b, img):
            raise ValueError("remove_stars needs a C-contiguous array (it is modified in place)")
        c, keep = self._catalog(cat)
        self._chk(self._lib.lfdmi_remove_stars(self._h, _ptr(b), n, h, w, C.byref(c), C.byref(rs),
                                               DEVICE if _is_dev(b) else HOST))
        return img

    # -- whole passes -------------------------------------------------------------------------
    def _pass(self, fn, img, params, dim, flip, extra):
        img, n, h, w, sq = self._batch(img)
        loc = DEVICE if _is_dev(img) else HOST
        p, keep = make_params(params, dim=dim)
        res = np.zeros(n, RESULT_DTYPE)
        K = p.nlinesInSet
        le = np.zeros((n, K, 2), np.float32)
        lb = np.zeros((n, K, 2), np.float32)
        self._chk(fn(self._h, _ptr(img), _dtype_code(img), n, h, w, int(flip), *extra, C.byref(p), _ptr(res),
                     _ptr(le), _ptr(lb), loc))
        return (res[0], le[0], lb[0]) if sq else (res, le, lb)

    def process_bright(self, img, params, flip=False):
        return self._pass(self._lib.lfdmi_process_bright, img, params, False, flip, ())

    def process_dim(self, img, params, flip=False, after_bright=False):
        return self._pass(self._lib.lfdmi_process_dim, img, params, True, flip, (int(after_bright),))

    def process_multiscale(self, img, params, rhos, dim=True, flip=False, after_bright=False):
        """One pass with HoughLines evaluated at every rho of ``rhos``; returns a structured array
        [len(rhos), n] (or [len(rhos)] for a single image): row s == process_dim/bright with houghMethod=rhos[s]."""
        img, n, h, w, sq = self._batch(img)
        loc = DEVICE if _is_dev(img) else HOST
        p, keep = make_params(params, dim=dim)
        rh = (C.c_double * len(rhos))(*[float(r) for r in rhos])
        res = np.zeros((len(rhos), n), RESULT_DTYPE)
        self._chk(self._lib.lfdmi_process_multiscale(self._h, _ptr(img), _dtype_code(img), n, h, w, int(flip), int(dim),
                                                     int(after_bright), C.byref(p), len(rhos), rh, _ptr(res), loc))
        return res[:, 0] if sq else res

    def detect_batch(self, frames, params_bright, params_dim, cat=None, rs=None, pinned=False):
        """frames: float32 (n,h,w) numpy or torch-CUDA; returns a structured array of n results.

        numpy frames of dtype '>f4' (the raw data unit of a FITS image) are accepted as they are and byte-swapped on the
        device.  ``pinned=True``: the array lives in memory from ``PinnedBuffer`` (DMA'd in place, no staging copy)."""
        if isinstance(frames, DeviceFrames):                  # decompressed on the device (Bz2Decoder.frames): swapped in place there
            n, h, w = frames.shape
            code = F32_BE
        elif not _is_dev(frames) and isinstance(frames, np.ndarray) and frames.dtype == np.dtype(">f4"):
            if not frames.flags.c_contiguous:
                raise ValueError("big-endian frames must be C-contiguous")
            shp = frames.shape
            n, h, w = (1, *shp) if len(shp) == 2 else shp
            code = F32_BE
        else:
            frames, n, h, w, sq = self._batch(frames)
            code = _dtype_code(frames)
            if code != F32:
                raise TypeError("detect_batch needs float32 frames")
        if pinned and _is_dev(frames):
            raise ValueError("pinned=True is for host arrays")
        pb, k1 = make_params(params_bright)
        pd, k2 = make_params(params_dim, dim=True)
        c, k3 = self._catalog(cat)
        res = np.zeros(n, RESULT_DTYPE)
        loc = DEVICE if _is_dev(frames) else (HOST_PINNED if pinned else HOST)
        self._chk(self._lib.lfdmi_detect_batch_raw(self._h, _ptr(frames), code, n, h, w,
                                                   C.byref(c) if c is not None else None,
                                                   C.byref(rs) if rs is not None else None,
                                                   C.byref(pb), C.byref(pd), _ptr(res), loc))
        return res

    def pinned_buffer(self, nbytes):
        """Page-locked host memory next to this context's GPU (lfdmi_host_alloc) as a ``PinnedBuffer``."""
        return PinnedBuffer(self, nbytes)

    def get_counters(self, slot0=0, n=None):
        """Work counters ([n, 20] int32, see LFDMI_COUNTERS) the last pass left for in-flight slots slot0 .. slot0+n-1."""
        n = self.max_inflight - slot0 if n is None else int(n)
        out = np.zeros((n, 20), np.int32)
        self._chk(self._lib.lfdmi_get_counters(self._h, int(slot0), n, _ptr(out)))
        return out

    def get_stage(self, slot, which, h, w):
        out = np.empty((h, w), np.uint8)
        self._chk(self._lib.lfdmi_get_stage(self._h, int(slot), int(which), int(h), int(w), _ptr(out), HOST))
        return out
