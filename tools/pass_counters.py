"""Developer tool: distribution of the per-frame work counters (runs, words, keys, slots) over a
synthetic batch, for the bright pass and for the dim pass of the frames the bright pass left."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lfd_amd import _native as Nv, synth
from lfd_amd.detecttrails import default_params

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pb, pd, prs = default_params()
rs = Nv.make_rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})
frames, cats = [], []
for k in range(n):
    img, cat, _ = synth.make_frame(k)
    frames.append(img); cats.append(cat)
frames = np.stack(frames)
ctx = Nv.Context(0, 1489, 2048, n)
ctx.remove_stars(frames, synth.pack_catalogs(cats), rs)
names = ["keys", "slots", "quads", "chunks_equ", "chunks_box", "peak_equ", "peak_box", "ovf", "detect", "big", "fgw", "bgw",
         "runf", "runb", "med", "nnz_equ", "nnz_box"]
for label, fn in (("bright", lambda: ctx.process_bright(frames, pb, flip=True)),
                  ("dim", lambda: ctx.process_dim(frames, pd, flip=True, after_bright=True))):
    fn()
    c = ctx.get_counters(0, n)
    print(label)
    for i, nm in enumerate(names):
        col = c[:, i]
        print("  %-8s min %7d  median %7d  p90 %7d  max %7d" % (nm, col.min(), np.median(col), np.percentile(col, 90), col.max()))
