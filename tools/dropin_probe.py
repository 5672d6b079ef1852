"""Developer tool: where DetectTrails.process spends its time on this box -- context + pinned buffers, the loader pool alone
at several thread counts, the GPU call on a loaded chunk, and process() end to end over a tree of N frames (hard links of a
few distinct files).  python tools/dropin_probe.py [frames=1024] [distinct=64]"""
import os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lfd_amd import synth
from lfd_amd.detecttrails import DetectTrails, default_params, loader, sdssfiles
from lfd_amd.detecttrails.detecttrails import process_loaded
from lfd_amd.detecttrails.processfield import use_context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 64
frames, cats = synth.make_frames(0, nd, synth.SDSS_SHAPE)
root = tempfile.mkdtemp(prefix="lfd_probe_", dir="/dev/shm")
try:
    t0 = time.perf_counter()
    synth.write_boss_tree(root, frames, cats, field0=100, link_to=n)
    print("tree: %d files (%d distinct) in %.1f s" % (n, nd, time.perf_counter() - t0), flush=True)
    keys = [(94, 1, "r", 100 + i) for i in range(n)]
    t0 = time.perf_counter()
    with use_context(1489, 2048, inflight=256) as ctx:
        t1 = time.perf_counter()
        print("context (256 slots): %.2f s" % (t1 - t0), flush=True)
        for thr in (4, 8, 16, 32, 64):
            t0 = time.perf_counter()
            ld = loader.FrameLoader(ctx, (1489, 2048), 256, threads=thr)
            t1 = time.perf_counter()
            ld.load(keys[:256], 0)
            t2 = time.perf_counter()
            out = ld.load(keys[256:512] if n >= 512 else keys[:256], 1)
            t3 = time.perf_counter()
            print("loader %2d threads: pinned buffers %.2f s, first chunk %.0f frames/s, second %.0f frames/s (%.1f GB/s)" %
                  (thr, t1 - t0, 256 / (t2 - t1), 256 / (t3 - t2), 256 * 12.2e-3 / (t3 - t2)), flush=True)
            if thr == 32:
                pb, pd, prs = default_params()
                import io
                t4 = time.perf_counter()
                process_loaded(io.StringIO(), io.StringIO(), out, pb, pd, prs)
                t5 = time.perf_counter()
                process_loaded(io.StringIO(), io.StringIO(), out, pb, pd, prs)
                t6 = time.perf_counter()
                print("   GPU call + rows on a loaded chunk: %.3f s, again %.3f s (%.0f frames/s)" % (t5 - t4, t6 - t5, 256 / (t6 - t5)), flush=True)
            ld.close()
    for thr in (8, 16):
        save = os.path.join(root, "out%d" % thr)
        os.makedirs(save)
        dt = DetectTrails(run=94, camcol=1, filter="r", savepath=save)
        t0 = time.perf_counter()
        dt.process(batch=256, loader_threads=thr)
        el = time.perf_counter() - t0
        st = dt.last_stats
        d = st["chunk_done_s"]
        print("process(batch=256, %d threads): %d frames in %.2f s = %.0f frames/s end to end (set-up %.2f s), %.0f frames/s from the "
              "first finished chunk on (%d frames per GPU call), %d rows" %
              (thr, n, el, n / el, st["setup_s"], (n - st["chunk_frames"]) / (d[-1] - d[0]), st["chunk_frames"], sum(1 for _ in open(dt.results))), flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
