"""Developer tool (GPU box): the full pipe on the stress workloads of lfd_amd.synth.STRESS -- frames/s against the benchmark's
sky in the same process, what the context had to do besides the fast path (Context.stats), records against the CPU oracle on a
sample.  usage: tools/stress_probe.py [distinct frames per workload = 16] [batch = 256] [oracle frames = 2] [names ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lfd_amd import _native as Nv, synth
from lfd_amd.batch import BatchDetector
from lfd_amd.detecttrails import default_params

nd = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n_or = int(sys.argv[3]) if len(sys.argv) > 3 else 2
names = sys.argv[4:] or ["baseline"] + list(synth.STRESS)
pb, pd, prs = default_params()
rs = Nv.make_rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})
dev = torch.device("cuda", 0)
O = None
if n_or:
    from oracle import lfd_oracle as O
    O.build()
    rs_o = O.rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})
base = None
for name in names:
    t0 = time.time()
    recipes = None if name == "baseline" else synth.stress_recipes(name, nd)
    host, cats = synth.make_frames(0, nd, synth.SDSS_SHAPE, min(16, os.cpu_count() or 1), True, recipes)
    tg = time.time() - t0
    idx = torch.arange(n, device=dev) % nd
    frames = torch.from_numpy(host).to(dev)[idx].contiguous()
    packed = synth.pack_catalogs([cats[i % nd] for i in range(n)])
    cat = {k: torch.from_numpy(v).to(dev) for k, v in packed.items()}
    det = BatchDetector(0, synth.SDSS_SHAPE, n, stream=torch.cuda.current_stream().cuda_stream)
    work = frames.clone()
    res = det.detect(work, pb, pd, cat, rs)                      # first call: tables grow / kernels switch here
    torch.cuda.synchronize()
    st0 = det.stats()
    times = []
    for _ in range(5):
        work.copy_(frames)
        torch.cuda.synchronize()
        t = time.perf_counter()
        res = det.detect(work, pb, pd, cat, rs)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t)
    st = det.stats()
    fps = n / min(times)
    det.enable_timing(True)
    work.copy_(frames)
    det.detect(work, pb, pd, cat, rs)
    torch.cuda.synchronize()
    tab = det.get_timing()
    det.enable_timing(False)
    top = " ".join("%s %.2f" % (k.replace("k_", ""), v[0]) for k, v in sorted(tab.items(), key=lambda kv: -kv[1][0])[:9] if v[0] > 0)
    if name == "baseline":
        base = fps
    cnt = det.get_counters()[:n]
    ok = ""
    if O is not None:
        good = 0
        for i in range(min(n_or, nd)):
            j = (5 if name == "one_crowded" else 0) + i if (5 + i) < nd else i
            want = O.detect_frame(host[j].copy(), pb, pd, cats[j], rs_o)
            good += all(res[j][k].item() == v for k, v in want.items())
        ok = " oracle %d/%d" % (good, min(n_or, nd))
    print("%-12s %8.0f frames/s (%5.1f %% of the benchmark's sky) first call %6.1f ms, then %6.1f ms | found %3d | runs f/b max %6d/%6d | steady: %s | first call: %s |%s gen %.0fs" % (
        name, fps, 100.0 * fps / base if base else 100.0, 0.0, 1e3 * min(times), int((res["found"] > 0).sum()), cnt[:, 12].max(), cnt[:, 13].max(),
        {k: st[k] - st0[k] for k in st if k != "scan_fused_on" and st[k] != st0[k]} or "fast path only",
        {k: v for k, v in st0.items() if v and k != "scan_fused_on" and k != "chunks"} or "fast path only", ok, tg), flush=True)
    print("             ms per call: " + top, flush=True)
    det.close()
    del frames, work, cat
    torch.cuda.empty_cache()
