"""The text interface after the hot path: ``results.txt`` rows and CCD-border snapping.

Mirrors the pieces of ``lfd/results`` that touch the detection output directly (the database
layer itself is out of scope): the 17-column row format written by process_field
(detecttrails.py:115-131, parsed by results/utils.py:185-210) and ``Event.snap2ccd``
(results/event.py:280-353), which clips the far-off-image end points that dictify_hough
produces to the 2048 x 2048 CCD box (results/ccd_dimensions.py:58-68).
"""

COLUMNS = ("run", "camcol", "filter", "field", "tai", "crpix1", "crpix2", "crval1", "crval2",
           "cd11", "cd12", "cd21", "cd22", "x1", "y1", "x2", "y2")

W_CAMCOL = 2048.0
H_FILTER = 2048.0


def format_result_row(run, camcol, filter, field, header, res):
    """One results.txt line (without newline). ``header`` maps TAI/CRPIX1/... to values, ``res`` is
    the dict returned by process_field_bright/dim."""
    keys = ("TAI", "CRPIX1", "CRPIX2", "CRVAL1", "CRVAL2", "CD1_1", "CD1_2", "CD2_1", "CD2_2")
    head = " ".join(str(x) for x in (run, camcol, filter, field, *(header[k] for k in keys)))
    return f"{head} {res['x1']} {res['y1']} {res['x2']} {res['y2']}"


def parse_result_row(string):
    """Row -> dict keyed by COLUMNS (ints for ids, floats for the rest), like results/utils.py:185-210."""
    s = string.split(" ")
    if len(s) < 17:
        raise ValueError(f"expected 17 space separated columns, got {len(s)}")
    out = {"run": int(s[0]), "camcol": int(s[1]), "filter": str(s[2]), "field": int(s[3])}
    for name, tok in zip(COLUMNS[4:], s[4:17]):
        out[name] = float(tok)
    return out


def read_results(path):
    """All rows of a results file as a list of dicts."""
    with open(path) as f:
        return [parse_result_row(line.strip()) for line in f if line.strip()]


def _points_on_sides(m, b):
    """Intersections of y = m x + b with the four borders of the CCD box, in the reference's order
    (x = 0, x = W, y = 0, y = H); corner hits appear twice (results/event.py:288-325)."""
    xs, ys = [], []
    if 0 <= b <= H_FILTER:
        xs.append(0)
        ys.append(b)
    t = m * W_CAMCOL + b
    if 0 <= t <= W_CAMCOL:
        xs.append(W_CAMCOL)
        ys.append(t)
    t = -b / m
    if 0 <= t <= W_CAMCOL:
        xs.append(t)
        ys.append(0)
    t = (H_FILTER - b) / m
    if 0 <= t <= W_CAMCOL:
        xs.append(t)
        ys.append(H_FILTER)
    return xs, ys


def snap2ccd(x1, y1, x2, y2):
    """End points clipped to the CCD border, (x1, y1, x2, y2).  Like the reference it needs a line that
    is neither vertical nor horizontal and raises ValueError when it finds no two border points."""
    m = (y2 - y1) / (x2 - x1)
    b = -m * x1 + y1
    xs, ys = _points_on_sides(m, b)
    if len(xs) in (2, 4):
        return xs[0], ys[0], xs[1], ys[1]
    raise ValueError(f"Could not compute edge points, returned: P1{xs} and P2{ys}.")
