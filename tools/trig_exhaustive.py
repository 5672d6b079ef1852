"""Developer check behind DESIGN.md's "device libm == glibc on the rectangle path": EVERY integer edge vector (dy, dx) with
|dy|, |dx| <= R (default 2048: any hull edge of an SDSS frame) through minAreaRect's angle (atan2 in double -> float32 degrees)
and boxPoints' cos / sin of it, device vs host, bit for bit.  python tools/trig_exhaustive.py [R]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lfd_amd import _native
from oracle import lfd_oracle as O

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ctx = _native.Context(0, 64, 64, 1)
xs = np.arange(-R, R + 1, dtype=np.float64)
bad = total = 0
for y0 in range(-R, R + 1, 256):
    ys = np.arange(y0, min(y0 + 256, R + 1), dtype=np.float64)
    yy, xx = np.meshgrid(ys, xs, indexing="ij")
    y, x = yy.ravel(), xx.ravel()
    dev = ctx.debug_trig(y, x)
    host = O.debug_trig(y, x)
    for d, h in zip(dev, host):
        bad += int(np.count_nonzero(d.view(np.uint32) != h.view(np.uint32)))
    total += y.size
print("integer edge vectors |dy|, |dx| <= %d: %d pairs x 3 float32 results (angle, cos / 2, sin / 2), %d differ between the device's libm and glibc"
      % (R, total, bad))
sys.exit(1 if bad else 0)
