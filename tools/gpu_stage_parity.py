"""Developer script: stage-by-stage GPU vs oracle comparison (run on the GPU box via gpurun)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import lfd_oracle as O
from lfd_amd import _native as Nv, synth

PB = dict(lwTresh=5, thetaTresh=0.15, dilateKernel=np.ones((4, 4), np.uint8), contoursMode=1, contoursMethod=1,
          minAreaRectMinLen=1, houghMethod=20, nlinesInSet=3, lineSetTresh=0.15, dro=25)
PD = dict(minFlux=0.02, addFlux=0.5, lwTresh=5, thetaTresh=0.15, erodeKernel=np.ones((3, 3), np.uint8),
          dilateKernel=np.ones((9, 9), np.uint8), contoursMode=1, contoursMethod=1, minAreaRectMinLen=1,
          houghMethod=20, nlinesInSet=3, lineSetTresh=0.15, dro=20)


def eq(name, a, b):
    ok = np.array_equal(a, b)
    print(f"  {name}: {'OK' if ok else 'MISMATCH'}" + ("" if ok else f" ({np.count_nonzero(np.asarray(a) != np.asarray(b))} differ)"), flush=True)
    return ok


def stages(ctx, img, params, dim, tag):
    print(tag, img.shape, flush=True)
    mode = O.PREP_BRIGHT_THEN_DIM if dim else O.PREP_BRIGHT
    g_o = O.prep(img, mode, flip=True, minFlux=params.get("minFlux", 0), addFlux=params.get("addFlux", 0))
    g_g, hist = ctx.prep_u8(img, mode, flip=True, minFlux=params.get("minFlux", 0), addFlux=params.get("addFlux", 0), want_hist=True)
    ok = eq("prep", g_g, g_o)
    ok &= eq("hist", hist, np.bincount(g_o.ravel(), minlength=256))
    e_o = O.equalize_hist(g_o)
    ok &= eq("equalize", ctx.equalize_hist(g_o), e_o)
    if dim:
        er_o = O.erode(e_o, params["erodeKernel"])
        ok &= eq("erode", ctx.erode(e_o, params["erodeKernel"]), er_o)
        d_o = O.dilate(er_o, params["dilateKernel"])
        ok &= eq("dilate", ctx.dilate(er_o, params["dilateKernel"]), d_o)
    else:
        d_o = O.dilate(e_o, params["dilateKernel"])
        ok &= eq("dilate", ctx.dilate(e_o, params["dilateKernel"]), d_o)
    c_o = O.canny(d_o)
    ok &= eq("canny", ctx.canny(d_o), c_o)
    det_o, box_o, nb_o = O.fit_min_area_rect(d_o)
    det_g, box_g, nb_g = ctx.fit_min_area_rect(d_o)
    print("  detection", det_o, det_g, "n_boxes", nb_o, nb_g, flush=True)
    ok &= (det_o == det_g) and (nb_o == nb_g)
    ok &= eq("box_img", box_g, box_o)
    for nm, im in (("equ", d_o), ("box", box_o)):
        acc_o = O.hough_accum(im, 20)
        acc_g = ctx.hough_accum(im, 20)
        ok &= eq(f"hough accum {nm}", acc_g, acc_o)
        l_o, n_o = O.hough_lines(im, 20)
        l_g, n_g = ctx.hough_lines(im, 20)
        same = (n_o == n_g) and ((l_o is None and l_g is None) or np.array_equal(l_o, l_g))
        print(f"  hough lines {nm}: n {n_o} {n_g} {'OK' if same else 'MISMATCH'}", flush=True)
        ok &= same
    return ok


def main():
    small = "--small" in sys.argv
    t0 = time.time()
    ctx = Nv.Context(0, 1489, 2048, 4)
    print("ctx created in %.1fs" % (time.time() - t0), flush=True)
    rng = np.random.default_rng(3)
    ok = True
    # small synthetic crop
    img, cat, truth = synth.make_frame(0)
    crop = np.ascontiguousarray(img[200:456, 300:812])
    ok &= stages(ctx, crop, PB, False, "crop bright")
    ok &= stages(ctx, crop, PD, True, "crop dim")
    odd = np.ascontiguousarray(img[100:231, 50:50 + 333])  # W not a multiple of 4 / 64
    ok &= stages(ctx, odd, PB, False, "odd-shape bright")
    if not small:
        ok &= stages(ctx, img, PB, False, "frame0 bright")
        img1, _, _ = synth.make_frame(1)
        ok &= stages(ctx, img1, PD, True, "frame1 dim")
    print("ALL OK" if ok else "SOME MISMATCH", flush=True)
    # whole passes
    rs_o = O.rs_params("r")
    rs_g = Nv.make_rs_params("r", 20, {'u': 22.0, 'g': 22.2, 'r': 22.2, 'i': 21.3, 'z': 20.5}, 60, 0.396, 3, 3)
    ks = list(range(2)) if small else list(range(8))
    frames, cats = [], []
    for k in ks:
        f, c, _ = synth.make_frame(k)
        frames.append(f); cats.append(c)
    batch = np.stack(frames)
    pc = synth.pack_catalogs(cats)
    t0 = time.time()
    res = ctx.detect_batch(batch.copy(), PB, PD, pc, rs_g)
    print("detect_batch %.3fs" % (time.time() - t0), flush=True)
    allok = True
    for i, k in enumerate(ks):
        r_o = O.detect_frame(frames[i].copy(), PB, PD, cats[i], rs_o)
        r_g = {n: res[i][n].item() for n in res.dtype.names}
        same = all(r_o[n] == r_g[n] for n in r_o)
        allok &= same
        print(k, "OK" if same else "MISMATCH", r_g if same else (r_o, r_g), flush=True)
    print("DETECT ALL OK" if allok else "DETECT MISMATCH", flush=True)
    return 0 if (ok and allok) else 1


if __name__ == "__main__":
    sys.exit(main())
