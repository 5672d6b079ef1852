"""Host-side readiness for eight ranks on one node, measured WITHOUT the GPU: R loader processes, each
bound to its own 1/R of the CPUs this process may run on (as the library binds a rank's feed threads to the CPUs next to its
GPU), each filling 64-frame (0.8 GB) buffers from a synthetic $BOSS tree in /dev/shm with the native readers
(lfdmi_fits_read_frames + lfdmi_fits_read_photoobj: the host side of DetectTrails.process), all at the same time; then the
same with a copy thread per process moving every filled buffer once more (the traffic a pinned upload's DMA reads add to the
host's memory system; a memcpy reads AND writes, so it overstates it).  Prints frames/s and GB/s per process and in total.
usage: tools/eight_loaders.py [ranks = 8] [seconds = 6] [distinct frames = 64]"""
import ctypes as C
import multiprocessing as mp
import os
import shutil
import sys
import tempfile
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def worker(rank, ranks, cpus, threads, root, seconds, with_copy, nfields, q, barrier):
    os.sched_setaffinity(0, cpus)
    os.environ["PHOTO_REDUX"] = os.path.join(root, "photo", "redux")
    os.environ["BOSS_PHOTOOBJ"] = os.path.join(root, "photoObj")
    from lfd_amd import _native
    from lfd_amd.detecttrails import loader, sdssfiles
    lib = _native.lib()
    h, w, slots, max_obj = 1489, 2048, 64, loader.MAX_OBJ
    fb = h * w * 4
    bufs = [np.zeros(slots * fb, np.uint8) for _ in range(2)]
    spare = np.zeros(slots * fb, np.uint8) if with_copy else None
    cats = {k: np.zeros((slots, max_obj, 5), np.float32) for k in ("ROWC", "COLC", "PSFMAG", "PETROTH90")}
    cats.update({k: np.zeros((slots, max_obj), np.int32) for k in ("NOBSERVE", "NDETECT")})
    cnt, st, pst, hl = (np.zeros(slots, np.int32) for _ in range(4))
    hdr = np.zeros((slots, loader.HDR_CAP), np.uint8)
    P = _native._ptr
    fields = [100 + (rank * 997 + i) % nfields for i in range(nfields)]
    copied = [0]
    stop = threading.Event()
    ready = threading.Semaphore(0)
    done_q = []

    def copier():                                       # stands in for the DMA's reads of the filled buffer
        while not stop.is_set():
            if not ready.acquire(timeout=0.05):
                continue
            b = done_q.pop(0)
            np.copyto(spare, b)
            copied[0] += 1

    for b in bufs + ([spare] if with_copy else []):      # touch every page before the clock starts (the library's pinned buffers are)
        b.fill(1)
    th = threading.Thread(target=copier, daemon=True) if with_copy else None
    if th:
        th.start()
    barrier.wait()
    t0 = time.perf_counter()
    frames = 0
    k = 0
    while time.perf_counter() - t0 < seconds:
        sel = [fields[(k * slots + i) % nfields] for i in range(slots)]
        fp = loader._paths([sdssfiles.filename("frame", run=94, camcol=1, field=f, filter="r") for f in sel])
        pp = loader._paths([sdssfiles.filename("photoObj", run=94, camcol=1, field=f) for f in sel])
        buf = bufs[k & 1]
        rc = lib.lfdmi_fits_read_frames(fp, slots, h, w, P(buf), threads, P(st), P(hdr), loader.HDR_CAP, P(hl))
        rc2 = lib.lfdmi_fits_read_photoobj(pp, slots, max_obj, P(cats["ROWC"]), P(cats["COLC"]), P(cats["PSFMAG"]), P(cats["PETROTH90"]),
                                           P(cats["NOBSERVE"]), P(cats["NDETECT"]), P(cnt), threads, P(pst))
        assert rc == 0 and rc2 == 0 and (st == 0).all() and (pst == 0).all(), (rc, rc2, st, pst)
        frames += slots
        if with_copy:
            done_q.append(buf)
            ready.release()
        k += 1
    dt = time.perf_counter() - t0
    stop.set()
    q.put((rank, frames / dt, frames * fb / dt / 1e9, copied[0], threads))


def main():
    ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
    nd = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    from lfd_amd import synth, usable_cores
    cpus = sorted(os.sched_getaffinity(0))
    usable = usable_cores()
    print("host: %d CPUs in this process's affinity mask (os.cpu_count() = %d); cgroup cpu.max: %s -> %d cores' worth of CPU time usable" % (
        len(cpus), os.cpu_count(), (open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a"), usable), flush=True)
    frames, cats = synth.make_frames(0, nd, synth.SDSS_SHAPE)
    root = tempfile.mkdtemp(prefix="lfd_8load_", dir="/dev/shm")
    nfields = 1024
    try:
        synth.write_boss_tree(root, frames, cats, field0=100, link_to=nfields)
        del frames
        for r in sorted({1, ranks}):
            for with_copy in (False, True):
                per = max(1, len(cpus) // r)
                threads = max(1, usable // r)             # (reader threads per process: the usable cores shared out, never more than the quota)
                q = mp.Queue()
                barrier = mp.Barrier(r)
                procs = [mp.Process(target=worker, args=(i, r, set(cpus[i * per:(i + 1) * per]), threads, root, seconds, with_copy, nfields, q, barrier)) for i in range(r)]
                for p in procs:
                    p.start()
                res = sorted(q.get() for _ in procs)
                for p in procs:
                    p.join()
                tot_f, tot_g = sum(x[1] for x in res), sum(x[2] for x in res)
                print("%d loader process%s x %d threads%s: %s frames/s each, %.0f frames/s = %.1f GB/s in total (%.0f frames/s per reader thread)" % (
                    r, "es" if r > 1 else "", res[0][4], " + a copy of every filled buffer" if with_copy else "",
                    "/".join("%.0f" % x[1] for x in res), tot_f, tot_g, tot_f / (r * res[0][4])), flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    mp.set_start_method("spawn")
    main()
