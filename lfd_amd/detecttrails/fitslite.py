"""Dependency-free reader for the two FITS layouts the detection path needs.

The reference reads frames and photoObj catalogues with the third-party ``fitsio`` package
(detecttrails.py:113-114, removestars.py:96); it is not installed here, so this module reads
  * the primary-HDU image (any BITPIX, BSCALE/BZERO applied) plus its header cards, and
  * named columns of a binary-table extension (TFORM codes L B I J K E D A with repeat counts)
straight from the 2880-byte block structure of the FITS standard.  Host-side IO, not hot path.
"""
import bz2

import numpy as np

BLOCK = 2880


def _quoted(text):
    """Contents of the quoted string that starts ``text`` ('' is an escaped quote; trailing blanks are not significant)."""
    s = text.lstrip()[1:]
    out, i = [], 0
    while i < len(s):
        if s[i] == "'":
            if i + 1 < len(s) and s[i + 1] == "'":
                out.append("'")
                i += 2
                continue
            break
        out.append(s[i])
        i += 1
    return "".join(out).rstrip()


def _parse_card(card):
    key = card[:8].strip()
    if key == "CONTINUE" and card[8:].lstrip().startswith("'"):
        return key, _quoted(card[8:])          # long-string convention: continues the previous card's value
    if card[8:10] != "= ":
        return key, None
    val = card[10:]
    if val.lstrip().startswith("'"):
        return key, _quoted(val)
    val = val.split("/")[0].strip()
    if val in ("T", "F"):
        return key, val == "T"
    try:
        return key, int(val)
    except ValueError:
        try:
            return key, float(val.replace("D", "E"))
        except ValueError:
            return key, val


def _read_header(buf, off):
    hdr = {}
    last_str = None
    while True:
        block = buf[off:off + BLOCK]
        if len(block) < BLOCK:
            raise ValueError("truncated FITS header")
        off += BLOCK
        text = block.decode("ascii", "replace")
        done = False
        for i in range(0, BLOCK, 80):
            card = text[i:i + 80]
            key, val = _parse_card(card)
            if key == "END":
                done = True
                break
            if key == "CONTINUE":
                # a string value ending in '&' goes on in the following CONTINUE card(s)
                if last_str is not None and isinstance(hdr.get(last_str), str) and hdr[last_str].endswith("&") and val is not None:
                    hdr[last_str] = hdr[last_str][:-1] + val
                continue
            if key and val is not None and key not in hdr:
                hdr[key] = val
                last_str = key if isinstance(val, str) else None
        if done:
            for k, v in hdr.items():           # (a final '&' with nothing after it is dropped, as cfitsio does)
                if isinstance(v, str) and v.endswith("&") and k == last_str:
                    hdr[k] = v[:-1]
            return hdr, off


def _data_size(hdr):
    naxis = hdr.get("NAXIS", 0)
    if naxis == 0:
        return 0
    n = abs(hdr["BITPIX"]) // 8
    for i in range(1, naxis + 1):
        n *= hdr[f"NAXIS{i}"]
    n = (n + hdr.get("PCOUNT", 0)) * hdr.get("GCOUNT", 1)
    return n


def _load(path):
    with open(path, "rb") as f:
        raw = f.read()
    if str(path).endswith(".bz2"):
        from . import bz2blocks
        raw = bz2blocks.decompress(raw, bz2blocks.shared_pool())   # (the blocks of the stream side by side)
    return raw


_BITPIX = {8: ">u1", 16: ">i2", 32: ">i4", 64: ">i8", -32: ">f4", -64: ">f8"}


def read_image(path, with_header=True):
    """Primary-HDU image as a native-endian numpy array (float32 for SDSS frames) + header dict."""
    return read_image_bytes(_load(path), path, with_header)


def read_image_bytes(buf, path="<bytes>", with_header=True):
    """read_image on the (decompressed) bytes of a FITS file."""
    hdr, off = _read_header(buf, 0)
    if hdr.get("NAXIS", 0) < 2:
        raise ValueError(f"{path}: primary HDU holds no image")
    shape = tuple(hdr[f"NAXIS{i}"] for i in range(hdr["NAXIS"], 0, -1))
    dt = np.dtype(_BITPIX[hdr["BITPIX"]])
    count = int(np.prod(shape))
    arr = np.frombuffer(buf, dt, count, off).reshape(shape)
    arr = arr.astype(dt.newbyteorder("="))
    bscale, bzero = hdr.get("BSCALE", 1), hdr.get("BZERO", 0)
    if bscale != 1 or bzero != 0:
        unsigned = {(16, 32768): np.uint16, (32, 2147483648): np.uint32, (8, -128): np.int8}
        if bscale == 1 and float(bzero).is_integer() and (hdr["BITPIX"], int(bzero)) in unsigned:
            arr = (arr.astype(np.int64) + int(bzero)).astype(unsigned[(hdr["BITPIX"], int(bzero))])  # the FITS unsigned convention
        else:
            arr = arr.astype(np.float64) * float(bscale) + float(bzero)
    return (arr, hdr) if with_header else arr


def read_header(path, ext=0):
    buf = _load(path)
    off = 0
    for _ in range(ext + 1):
        hdr, off = _read_header(buf, off)
        off_next = off + (_data_size(hdr) + BLOCK - 1) // BLOCK * BLOCK
        last = hdr
        off = off_next
    return last


_TFORM = {"L": ("u1", 1), "B": ("u1", 1), "I": (">i2", 2), "J": (">i4", 4), "K": (">i8", 8),
          "E": (">f4", 4), "D": (">f8", 8), "A": ("S1", 1)}


def read_table(path, columns, ext=1):
    """dict name -> array for the requested columns of binary-table extension ``ext``."""
    buf = _load(path)
    off = 0
    hdr = None
    for _ in range(ext + 1):
        hdr, off = _read_header(buf, off)
        data_off = off
        off = off + (_data_size(hdr) + BLOCK - 1) // BLOCK * BLOCK
    if hdr.get("XTENSION", "").strip() != "BINTABLE":
        raise ValueError(f"{path}: extension {ext} is not a binary table")
    row_bytes, nrows = hdr["NAXIS1"], hdr["NAXIS2"]
    want = {c.upper() for c in columns}
    out, pos = {}, 0
    for i in range(1, hdr["TFIELDS"] + 1):
        form = hdr[f"TFORM{i}"].strip()
        j = 0
        while j < len(form) and form[j].isdigit():
            j += 1
        rep = int(form[:j]) if j else 1
        code = form[j]
        if code in ("P", "Q"):
            width, dt = (8 if code == "P" else 16) * rep, None
        elif code == "X":
            width, dt = (rep + 7) // 8, None
        elif code in ("C", "M"):
            width, dt = (8 if code == "C" else 16) * rep, None
        else:
            dt, size = _TFORM[code]
            width = size * rep
        name = str(hdr.get(f"TTYPE{i}", f"COL{i}")).strip().upper()
        if name in want:
            if dt is None:
                raise ValueError(f"column {name}: TFORM {form} not supported")
            col = np.ndarray((nrows, rep), dt, buf, data_off + pos, (row_bytes, np.dtype(dt).itemsize))
            col = col.astype(np.dtype(dt).newbyteorder("=")) if code != "A" else col.copy()
            if code == "L":
                col = (col == ord("T"))
            tscal, tzero = hdr.get(f"TSCAL{i}", 1), hdr.get(f"TZERO{i}", 0)
            if (tscal != 1 or tzero != 0) and code in "BIJKED":
                if tscal == 1 and float(tzero).is_integer() and code in "BIJK":
                    col = col.astype(np.int64) + int(tzero)      # unsigned integers stored with an offset
                else:
                    col = col * tscal + tzero
            out[name] = col[:, 0] if rep == 1 else col
        pos += width
    missing = want - set(out)
    if missing:
        raise KeyError(f"{path}: missing columns {sorted(missing)}")
    return out


# ---- writers (tests and synthetic data trees) -------------------------------------------------
def _card(key, val, comment=""):
    if isinstance(val, bool):
        v = f"{'T' if val else 'F':>20}"
    elif isinstance(val, (int, np.integer)):
        v = f"{int(val):>20}"
    elif isinstance(val, (float, np.floating)):
        t = repr(float(val)).upper()          # keeps a decimal point or exponent: stays a float on read
        v = f"{t:>20}"
    else:
        s = "'" + str(val).replace("'", "''").ljust(8) + "'"
        v = f"{s:<20}"
    return f"{key:<8}= {v} / {comment}"[:80].ljust(80)


def _finish_header(cards):
    text = "".join(cards) + "END".ljust(80)
    text += " " * (-len(text) % BLOCK)
    return text.encode("ascii")


def _pad(b):
    return b + b"\0" * (-len(b) % BLOCK)


def write_image(path, arr, header=None):
    arr = np.asarray(arr)
    bitpix = {"u1": 8, "i2": 16, "i4": 32, "i8": 64, "f4": -32, "f8": -64}[arr.dtype.str[1:]]
    cards = [_card("SIMPLE", True), _card("BITPIX", bitpix), _card("NAXIS", arr.ndim)]
    for i, n in enumerate(arr.shape[::-1], 1):
        cards.append(_card(f"NAXIS{i}", n))
    for k, v in (header or {}).items():
        cards.append(_card(k, v))
    data = arr.astype(arr.dtype.newbyteorder(">")).tobytes()
    with open(path, "wb") as f:
        f.write(_finish_header(cards))
        f.write(_pad(data))


def write_table(path, columns):
    """columns: dict name -> (nrows,) or (nrows, rep) array of dtype u1/i2/i4/i8/f4/f8."""
    codes = {"u1": "B", "i2": "I", "i4": "J", "i8": "K", "f4": "E", "f8": "D"}
    names = list(columns)
    arrs = [np.atleast_2d(np.asarray(columns[n]).T).T if np.asarray(columns[n]).ndim == 1 else np.asarray(columns[n])
            for n in names]
    arrs = [a.reshape(len(a), -1) for a in arrs]
    nrows = len(arrs[0])
    row_bytes = sum(a.shape[1] * a.dtype.itemsize for a in arrs)
    cards = [_card("XTENSION", "BINTABLE"), _card("BITPIX", 8), _card("NAXIS", 2), _card("NAXIS1", row_bytes),
             _card("NAXIS2", nrows), _card("PCOUNT", 0), _card("GCOUNT", 1), _card("TFIELDS", len(names))]
    for i, (n, a) in enumerate(zip(names, arrs), 1):
        cards.append(_card(f"TTYPE{i}", n))
        cards.append(_card(f"TFORM{i}", f"{a.shape[1]}{codes[a.dtype.str[1:]]}"))
    rec = np.zeros((nrows, row_bytes), np.uint8)
    pos = 0
    for a in arrs:
        b = a.astype(a.dtype.newbyteorder(">")).view(np.uint8).reshape(nrows, -1)
        rec[:, pos:pos + b.shape[1]] = b
        pos += b.shape[1]
    primary = [_card("SIMPLE", True), _card("BITPIX", 8), _card("NAXIS", 0), _card("EXTEND", True)]
    with open(path, "wb") as f:
        f.write(_finish_header(primary))
        f.write(_finish_header(cards))
        f.write(_pad(rec.tobytes()))
