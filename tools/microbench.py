"""Developer micro-benchmark: per-operator kernel times on device-resident batches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lfd_amd import _native as Nv, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = Nv.Context(0, 1489, 2048, n)
frames = np.stack([synth.make_frame(k, with_catalog=False)[0] for k in range(8)])
gray = ctx.prep_u8(frames, Nv.PREP_BRIGHT, flip=True)
equ = ctx.dilate(ctx.equalize_hist(gray), np.ones((4, 4), np.uint8))
cases = {"zeros": np.zeros((1489, 2048), np.uint8), "gray": gray[0], "equ": equ[0],
         "rand": np.random.default_rng(0).integers(0, 256, (1489, 2048), dtype=np.uint8)}
for name, img in cases.items():
    d = torch.from_numpy(np.repeat(img[None], n, 0)).cuda()
    torch.cuda.synchronize()
    for op, fn in (("dilate4", lambda: ctx.dilate(d, np.ones((4, 4), np.uint8))),
                   ("dilate9", lambda: ctx.dilate(d, np.ones((9, 9), np.uint8))),
                   ("erode3", lambda: ctx.erode(d, np.ones((3, 3), np.uint8))),
                   ("canny", lambda: ctx.canny(d))):
        fn()
        ctx.enable_timing(True)
        for _ in range(3):
            fn()
        t = ctx.get_timing()
        ctx.enable_timing(False)
        parts = {k: round(v[0] / 3 / n * 1e3, 2) for k, v in t.items() if v[1]}
        print(f"{name:6s} {op:8s} us/frame:", parts, flush=True)
