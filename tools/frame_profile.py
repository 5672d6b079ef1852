"""Developer tool: per-phase wall clock of k_frame_contours (LFDMI_FRAME_PROFILE=1), median / max over a batch."""
import os, sys, ctypes as C
os.environ["LFDMI_FRAME_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lfd_amd import _native as Nv, synth
from lfd_amd.detecttrails import default_params
n = 256
pb, pd, prs = default_params()
frames = np.stack([synth.make_frame(k, with_catalog=False)[0] for k in range(32)])
frames = np.concatenate([frames] * 8)
ctx = Nv.Context(0, 1489, 2048, n)
ctx._lib.lfdmi_debug_frame_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
fgnames = ["merge", "flatten+strong", "outer keys | edge bits", "edge+extremes | write-out"]
names = ["bg merge", "flatten", "hole extents", "write-out", "outer keys", "slot table", "hole keys", "extremes"]
for label, fn in (("bright", lambda: ctx.process_bright(frames, pb, flip=True)),
                  ("dim", lambda: ctx.process_dim(frames, pd, flip=True, after_bright=True))):
    fn(); fn()
    out = np.zeros((n, 16), np.int64)
    assert ctx._lib.lfdmi_debug_frame_profile(ctx._h, n, out.ctypes.data) == 0
    t = out[:, :8] / 100.0
    d = np.diff(np.concatenate([np.zeros((n, 1)), t], 1), axis=1)
    print(label, " ".join("%s %.0f/%.0f" % (nm, np.median(d[:, i]), d[:, i].max()) for i, nm in enumerate(names)), "| total %.0f/%.0f us (median/max)" % (np.median(t[:, 7]), t[:, 7].max()))
    tf = out[:, 8:12] / 100.0
    df = np.diff(np.concatenate([np.zeros((n, 1)), tf], 1), axis=1)
    print(label, "k_frame_fg:", " ".join("%s %.0f/%.0f" % (nm, np.median(df[:, i]), df[:, i].max()) for i, nm in enumerate(fgnames)), "| total %.0f/%.0f us" % (np.median(tf[:, 3]), tf[:, 3].max()))
