"""Bulk parity: N synthetic SDSS frames through the HIP pipe and through the CPU oracle (fanned out over
host processes), record by record.  Usage: python tools/bulk_parity.py [n_frames] [first_k]
       python tools/bulk_parity.py lsst [n_frames] [first_k]   (4096 x 4096 frames, dim pass with a 9 x 9 erosion, rho 20 / 10 / 5)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiprocessing as mp
import numpy as np


def _oracle(k):
    from lfd_amd import synth
    from lfd_amd.detecttrails import default_params
    from oracle import lfd_oracle as O
    pb, pd, prs = default_params()
    rs = O.rs_params("r", **{a: b for a, b in prs.items() if a != "debug"})
    img, cat, truth = synth.make_frame(k)
    return k, truth["streak"], O.detect_frame(img, pb, pd, cat, rs)


RHOS = (20.0, 10.0, 5.0)


def _lsst_params():
    from lfd_amd.detecttrails import default_params
    _, pd, _ = default_params()
    return dict(pd, erodeKernel=np.ones((9, 9), np.uint8))


def _oracle_lsst(k):
    from lfd_amd import synth
    from oracle import lfd_oracle as O
    pd = _lsst_params()
    img = synth.make_frame(k, shape=synth.LSST_SHAPE, with_catalog=False)[0]
    return k, [O.process_dim(img.copy(), dict(pd, houghMethod=r), flip=True) for r in RHOS]


def main_lsst(n, k0):
    workers = min(16, os.cpu_count() or 8)
    t0 = time.time()
    with mp.get_context("spawn").Pool(workers) as pool:
        want = dict(pool.map(_oracle_lsst, range(k0, k0 + n), chunksize=1))
    t_cpu = time.time() - t0
    from lfd_amd import synth
    from lfd_amd.batch import BatchDetector
    pd = _lsst_params()
    det = BatchDetector(0, synth.LSST_SHAPE, 8)
    bad = 0
    try:
        for c0 in range(0, n, 8):
            ks = list(range(k0 + c0, k0 + min(n, c0 + 8)))
            frames = np.stack([synth.make_frame(k, shape=synth.LSST_SHAPE, with_catalog=False)[0] for k in ks])
            res = det.multiscale(frames, pd, list(RHOS), dim=True, flip=True)
            for i, k in enumerate(ks):
                for s_, rec in enumerate(want[k]):
                    if not all(res[s_][i][f].item() == v for f, v in rec.items()):
                        bad += 1
                        print("MISMATCH", k, RHOS[s_], rec, {f: res[s_][i][f].item() for f in rec}, flush=True)
    finally:
        det.close()
    found = sum(1 for k in want for r in want[k] if r["found"])
    print(f"lsst frames {n} (k0={k0}) x {len(RHOS)} scales: identical {n * len(RHOS) - bad}/{n * len(RHOS)}; oracle {t_cpu:.1f}s on "
          f"{workers} procs; {found} of them with a line", flush=True)
    return 1 if bad else 0


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "lsst":
        return main_lsst(int(sys.argv[2]) if len(sys.argv) > 2 else 16, int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    k0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    workers = min(16, os.cpu_count() or 8)
    t0 = time.time()
    with mp.get_context("spawn").Pool(workers) as pool:      # fresh interpreters: nothing GPU-related is inherited
        want = pool.map(_oracle, range(k0, k0 + n), chunksize=2)
    t_cpu = time.time() - t0
    from lfd_amd import _native, synth
    from lfd_amd.detecttrails import default_params
    pb, pd, prs = default_params()
    rs = _native.make_rs_params("r", **{a: b for a, b in prs.items() if a != "debug"})
    bad, kinds = 0, {}
    with _native.Context(0, 1489, 2048, 64) as ctx:
        for c0 in range(0, n, 64):
            ks = list(range(k0 + c0, k0 + min(n, c0 + 64)))
            frames, cats = zip(*[synth.make_frame(k)[:2] for k in ks])
            res = ctx.detect_batch(np.stack(frames), pb, pd, synth.pack_catalogs(list(cats)), rs)
            for i, k in enumerate(ks):
                kk, streak, rec = want[k - k0]
                same = all(res[i][f].item() == v for f, v in rec.items())
                kinds[(streak, rec["found"])] = kinds.get((streak, rec["found"]), 0) + 1
                if not same:
                    bad += 1
                    print("MISMATCH", k, rec, {f: res[i][f].item() for f in rec}, flush=True)
    print(f"frames {n} (k0={k0}): identical {n - bad}/{n}; oracle {t_cpu:.1f}s on {workers} procs; "
          f"(truth, found) counts {sorted(kinds.items())}", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
