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


def draw_lines(hough, image, nlines, name, path, compression=0, color=(255, 0, 0)):
    """Overlay the first ``nlines`` Hough lines on a grey image and save it as PNG
    (reference: processfield.py:153-198; 2-px lines, end points as in dictify_hough)."""
    import os
    n_x, n_y = image.shape
    rgb = np.repeat(np.asarray(image, np.uint8)[:, :, None], 3, axis=2)
    yy, xx = np.mgrid[0:n_x, 0:n_y]
    for params in (hough[:nlines] if hough is not None else []):
        rho, theta = params[0]
        c, s = float(np.cos(theta)), float(np.sin(theta))
        dist = np.abs(xx * c + yy * s - float(rho))
        rgb[dist <= 1.0] = color
    write_png(os.path.join(path, name + ".png"), rgb, compression)
