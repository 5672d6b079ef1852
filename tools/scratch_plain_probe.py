import os, sys, time, tempfile, shutil
sys.path.insert(0, "/root/repo")
import numpy as np
from lfd_amd import synth
from lfd_amd.detecttrails import DetectTrails
n = 256
host, cats = synth.make_frames(0, n, synth.SDSS_SHAPE, 16, True)
root = tempfile.mkdtemp(prefix="lfd_boss_", dir="/dev/shm")
try:
    hdr = synth.write_boss_tree(root, host, cats, run=94, camcol=1, filter="r", field0=100, link_to=8192)
    for slots in (64, 128, 256, 64, 256):
        os.environ["LFD_LOADER_SLOTS"] = str(slots)
        save = os.path.join(root, "out%d_%f" % (slots, time.time()))
        os.makedirs(save)
        dt = DetectTrails(run=94, camcol=1, filter="r", savepath=save)
        t0 = time.perf_counter(); dt.process(batch=256); el = time.perf_counter() - t0
        st = dt.last_stats; done = st["chunk_done_s"]
        print("slots %d: %.0f frames/s end to end, %.0f steady, setup %.2f s" % (slots, 8192 / el, (8192 - st["chunk_frames"]) / (done[-1] - done[0]), st["setup_s"]), flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
