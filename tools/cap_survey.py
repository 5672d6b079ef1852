"""Developer tool: how much of each per-frame table a pass really uses (runs, keys, row slots, Hough chunks, peaks),
for the SDSS batch workload and for LSST-size dim passes -- the numbers behind the default lfdmi_caps.
Usage: python tools/cap_survey.py [n_sdss] [n_lsst]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from lfd_amd import _native as Nv, synth
from lfd_amd.detecttrails import default_params

NAMES = ["keys", "slots", "quads", "chunks_equ", "chunks_box", "peak_equ", "peak_box", "ovf", "detect", "big", "fgw", "bgw",
         "runf", "runb", "med", "nnz_equ", "nnz_box"]


def show(label, c, N):
    print(label, flush=True)
    for i, nm in enumerate(NAMES):
        col = c[:, i]
        print("  %-10s median %8d  p99 %8d  max %8d   (N / max = %.0f)" % (nm, np.median(col), np.percentile(col, 99), col.max(), N / max(1, col.max())))


def gen(shape, ks):
    with ThreadPoolExecutor(16) as ex:
        out = list(ex.map(lambda k: synth.make_frame(k, shape)[:2], ks))
    return np.stack([o[0] for o in out]), [o[1] for o in out]


def main():
    n_sdss = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n_lsst = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    pb, pd, prs = default_params()
    rs = Nv.make_rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})
    if n_sdss:
        frames, cats = gen(synth.SDSS_SHAPE, range(n_sdss))
        with Nv.Context(0, 1489, 2048, min(64, n_sdss)) as ctx:
            ctx.remove_stars(frames, synth.pack_catalogs(cats), rs)
            acc = {"bright": [], "dim": []}
            for c0 in range(0, n_sdss, 64):
                sub = frames[c0:c0 + 64]
                ctx.process_bright(sub, pb, flip=True)
                acc["bright"].append(ctx.get_counters(0, len(sub)))
                ctx.process_dim(sub, pd, flip=True, after_bright=True)
                acc["dim"].append(ctx.get_counters(0, len(sub)))
            for k, v in acc.items():
                show("SDSS %s pass, %d frames (after remove_stars)" % (k, n_sdss), np.concatenate(v), 1489 * 2048)
    if n_lsst:
        frames, cats = gen(synth.LSST_SHAPE, range(n_lsst))
        blot = frames.copy()
        with Nv.Context(0, 4096, 4096, 2) as ctx:
            ctx.remove_stars(blot, synth.pack_catalogs(cats), rs)
            for label, src, ek in (("dim 9x9 erosion, no remove_stars", frames, 9), ("dim 9x9 erosion after remove_stars", blot, 9),
                                   ("dim 3x3 erosion, no remove_stars", frames, 3), ("bright", frames, 0)):
                p = dict(pd)
                if ek:
                    p["erodeKernel"] = np.ones((ek, ek), np.uint8)
                out = []
                t0 = time.time()
                for c0 in range(0, n_lsst, 2):
                    if ek:
                        res, _, _ = ctx.process_dim(src[c0:c0 + 2], p, flip=True)
                    else:
                        res, _, _ = ctx.process_bright(src[c0:c0 + 2].copy(), pb, flip=True)
                    out.append(ctx.get_counters(0, 2))
                show("LSST %s, %d frames (%.2f s)" % (label, n_lsst, time.time() - t0), np.concatenate(out), 4096 * 4096)


if __name__ == "__main__":
    main()
