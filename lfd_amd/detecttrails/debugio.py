"""Debug image dumps without cv2: an 8-bit PNG writer and Hough-line overlays.

Stands in for the cv2.imwrite / cv2.cvtColor / cv2.line calls the reference uses only when
``debug=True`` (processfield.py:153-198, :349-378, :459-496).  Host-side, not on the hot path.
"""
import struct
import zlib

import numpy as np


def _chunk(tag, data):
    body = tag + data
    return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xffffffff)


def write_png(path, img, compression=3):
    """img: (h, w) uint8 grey or (h, w, 3) uint8 RGB."""
    img = np.ascontiguousarray(img, np.uint8)
    if img.ndim == 2:
        color, rows = 0, img
    elif img.ndim == 3 and img.shape[2] == 3:
        color, rows = 2, img.reshape(img.shape[0], -1)
    else:
        raise ValueError("write_png: expected (h,w) or (h,w,3) uint8")
    h, w = img.shape[:2]
    raw = np.empty((h, rows.shape[1] + 1), np.uint8)
    raw[:, 0] = 0  # filter type None
    raw[:, 1:] = rows
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0)))
        f.write(_chunk(b"IDAT", zlib.compress(raw.tobytes(), int(compression))))
        f.write(_chunk(b"IEND", b""))


def read_png(path):
    """Decoder for the PNGs this module writes (8-bit grey or RGB, filter type 0): (h, w) or (h, w, 3) uint8."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos, idat, w = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h, depth, color = struct.unpack(">IIBB", body[:10])
            if depth != 8 or color not in (0, 2):
                raise ValueError("unsupported PNG layout")
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    ch = 3 if color == 2 else 1
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * ch + 1)
    if raw[:, 0].any():
        raise ValueError("filtered scanlines are not supported")
    img = raw[:, 1:]
    return img.reshape(h, w, 3).copy() if ch == 3 else img.copy()


def draw_lines(hough, image, nlines, name, path, compression=0, color=(255, 0, 0)):
    """Overlay the first ``nlines`` Hough lines on a grey image and save it as ``<name>.png``
    (reference: processfield.py:153-198).  Like the reference, a line runs through the two end points of
    ``dictify_hough`` (coordinates truncated to int) and is 2 px wide; ``color`` is in OpenCV's B, G, R order
    (the default draws blue lines, the dim pass's box overlay red ones), the file holds R, G, B."""
    import os
    n_x, n_y = image.shape
    rgb = np.repeat(np.asarray(image, np.uint8)[:, :, None], 3, axis=2)
    yy, xx = np.mgrid[0:n_x, 0:n_y]
    reach = n_x + n_y
    for params in (hough[:nlines] if hough is not None else []):
        try:
            rho, theta = params[0]
            x0, y0 = np.cos(theta) * rho, np.sin(theta) * rho
            x1, y1 = int(x0 - reach * np.sin(theta)), int(y0 + reach * np.cos(theta))
            x2, y2 = int(x0 + reach * np.sin(theta)), int(y0 - reach * np.cos(theta))
        except Exception:  # noqa: BLE001 - the reference skips lines it cannot draw
            continue
        dx, dy = float(x2 - x1), float(y2 - y1)
        norm = (dx * dx + dy * dy) ** 0.5
        if norm == 0:
            continue
        dist = np.abs((xx - x1) * dy - (yy - y1) * dx) / norm
        rgb[dist <= 1.0] = tuple(color)[::-1]
    write_png(os.path.join(path, name + ".png"), rgb, compression)
