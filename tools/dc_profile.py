"""Developer tool: where the waves of k_dilate_canny_t spend their cycles (LFDMI_DC_PROFILE=1 builds s_memtime stamps in),
per stage, summed over all waves of a batch, plus the number of active tiles."""
import os, sys, ctypes as C
os.environ["LFDMI_DC_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lfd_amd import _native as Nv, synth
from lfd_amd.detecttrails import default_params
n = 128
pb, pd, prs = default_params()
frames, _ = synth.make_frames(0, n, with_catalog=False)
ctx = Nv.Context(0, 1489, 2048, n)
ctx._lib.lfdmi_debug_frame_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
names = ["input wait+stage", "masks", "hmax", "vmax+lut", "borders+equ bits", "sobel", "nms+store"]
for label, fn in (("bright", lambda: ctx.process_bright(frames, pb, flip=True)),
                  ("dim", lambda: ctx.process_dim(frames, pd, flip=True, after_bright=True))):
    fn(); fn()
    raw = np.zeros(n * 16, np.int64)                     # (the getter copies 16 values per slot; the tile kernel's clocks are the first 8 n)
    assert ctx._lib.lfdmi_debug_frame_profile(ctx._h, n, raw.ctypes.data) == 0
    out = raw[:n * 8].reshape(n, 8)
    tiles = out[:, 7].sum()
    cyc = out[:, :7].sum(axis=0)
    ntl = ctx.get_counters(0, n)[:, 17]
    print("%s: active tiles/frame median %d (of %d), tiles that ran the stages %.0f/frame; cycles per staged tile: total %.0f" %
          (label, np.median(ntl), 94 * 32, tiles / n, cyc.sum() / max(1, tiles)))
    print("   " + "  ".join("%s %.0f (%.0f%%)" % (nm, c / max(1, tiles), 100.0 * c / cyc.sum()) for nm, c in zip(names, cyc)))
