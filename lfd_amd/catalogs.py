"""photoObj catalogues in the layout ``lfdmi_catalog`` takes (include/lfdmi.h): the per-frame columns that
removestars.py:96-104 reads, stacked into padded arrays [n_frames, max_obj, ...] plus a count per frame."""
import numpy as np

COLUMNS5 = ("ROWC", "COLC", "PSFMAG", "PETROTH90")   # [n_obj, 5] float32: one value per filter (ugriz)
COLUMNS1 = ("NOBSERVE", "NDETECT")                    # [n_obj] int32


def empty_packed(n_frames, max_obj):
    """Zeroed arrays for ``n_frames`` catalogues of up to ``max_obj`` objects (NDETECT = 1 != NOBSERVE = 0 in the padding:
    a padded row never passes remove_stars' NOBSERVE == NDETECT test, and rows beyond ``count`` are not looked at anyway)."""
    out = {k: np.zeros((n_frames, max_obj, 5), np.float32) for k in COLUMNS5}
    out["NOBSERVE"] = np.zeros((n_frames, max_obj), np.int32)
    out["NDETECT"] = np.ones((n_frames, max_obj), np.int32)
    out["count"] = np.zeros(n_frames, np.int32)
    return out


def pack_catalogs(cats):
    """Stack per-frame catalogues (dicts of columns; ``None`` or an empty table = no objects) into padded arrays
    [n_frames, max_obj, ...] + counts."""
    sizes = [0 if c is None else len(c["NOBSERVE"]) for c in cats]
    out = empty_packed(len(cats), max(sizes + [1]))
    for i, (c, k) in enumerate(zip(cats, sizes)):
        out["count"][i] = k
        if k:
            for key in COLUMNS5 + COLUMNS1:
                out[key][i, :k] = c[key]
    return out
