#!/usr/bin/env python3
"""Headline benchmark: frames/s of the detecttrails hot path on N MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 1                      # BASELINE configs[2] (default) + secondary legs
    python bench.py --workload lsst                                     # BASELINE configs[4] as the headline of the run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W           # what the driver's scaling step runs

workload sdss (BASELINE.json configs[2], the configuration the metric is quoted on): one "step" = one pass of the
full pipe (remove_stars -> flip -> bright pass -> dim pass where the bright pass found nothing;
detecttrails.py:119-131) over this rank's batch of 256 synthetic 2048x1489 float32 frames, resident in HBM.
workload lsst (configs[4]): one step = the dim pass with a 9x9 erosion (processfield.py:453-506) over this rank's
batch of 4096x4096 float32 frames, HoughLines evaluated at rho = 20, 10 and 5 ("multi-scale Hough": SURVEY.md 8d).
Weak scaling: every rank owns --frames-per-gpu frames, no data-path collective; one barrier-bracketed timed
region, max over ranks.  Rank 0 prints ONE JSON line.

roofline: the library brackets launches with HIP events on the launch stream.  Bracketing all ~40 launches of a
step costs ~6 % of it, so inside the timed region only the few largest kernels are bracketed (nominated by the
fully bracketed warm-up steps); the dominant one is priced with ITS share of SURVEY.md 8(d)'s algorithmic bytes
(KERNEL_BYTES_PER_PX: every stage's bytes are charged exactly once across its kernels) x the frames its launches
worked on.  `stages` and `canny_hough` come from one extra, fully bracketed step run after the timed region.

Secondary fields of the default (N = 1, sdss) invocation -- never `value`:
  host_resident  the same step with the frames handed over as host buffers (PCIe-inclusive);
  sustained      the same device-resident step repeated for >= 3 s, GPU clock before / after;
  dropin         DetectTrails(run=...).process(batch=256) over a synthetic $BOSS tree of FITS files in /dev/shm (plain and
                 .bz2), rows compared with the device-resident run (the drop-in itself, file reading included);
  lsst           the configs[4] step (value, roofline, canny_hough) -- the driver never passes --workload lsst;
  cpu_baseline   the C oracle (oracle/, a port of the reference's algorithm) on a bounded sample of the same frames: one
                 host thread, and all of the rank's host cores; `opencv`: the reference's own cv2 call sequence
                 (oracle/cv2_path.py) where `import cv2` works, "unavailable" otherwise.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# SURVEY 8(d)'s secondary bound for HoughLines (180 votes per non-zero pixel): one LDS atomic add per vote.  A wave64
# ds_add costs 4 LDS cycles (MI355X_MICROARCH.md, LDS: ds_write_b32-class, 2 x 32 lanes + the address/data transfer), i.e.
# 16 votes per clock per CU; 256 CUs at 2.4 GHz.  The vector units would allow more: >= 4 lane-operations per vote
# (x cos, + y sin, round, address) on 4 SIMD-32 per CU.
CU_COUNT, CLOCK_HZ = 256, 2.4e9
LDS_ATOMIC_VOTES_PER_S = CU_COUNT * 16 * CLOCK_HZ
VALU_VOTES_PER_S = CU_COUNT * 4 * 32 * CLOCK_HZ / 4.0

# algorithmic bytes per pixel of every stage of a pass, and the kernels (timing slots of the library)
# that make up the stage.  A kernel that spans two stages appears with a share of each (bytes, fraction of its time).
#   prep 5N | erode 2N | dilate 2N | Canny 2N | contours+rect+fill 2N | Hough 1N per image
# The fused tile kernel's time is split between dilate and Canny by the committed stage-clock profile (dc_split()).


def dc_split():
    """(dilate share, Canny share, source) of k_dilate_canny's time from the newest profiles/r*_dc_stage_clocks.txt
    (tools/dc_profile.py: s_memtime stamps per stage): hmax + vmax+lut + borders are the dilation, sobel + nms+store are
    Canny, the input wait and the activity masks serve both and are split in the same proportion."""
    import glob
    import re
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_dc_stage_clocks.txt")), reverse=True):
        try:
            dil = can = 0.0
            for line in open(path):
                if "hmax" not in line:
                    continue
                v = {k.strip(): float(x) for k, x in re.findall(r"([a-z+ ]+?)\s+(\d+)\s+\(\d+%\)", line)}
                dil += v["hmax"] + v["vmax+lut"] + v["borders+equ bits"]
                can += v["sobel"] + v["nms+store"]
            if dil > 0 and can > 0:
                return dil / (dil + can), can / (dil + can), os.path.basename(path)
        except (OSError, KeyError, ValueError):
            continue
    return 0.5, 0.5, None


def fg_split():
    """(hysteresis share, contour-key share, source) of k_frame_fg's time from the newest profiles/r*_frame_profile.txt
    (tools/frame_profile.py: wall-clock stamps per phase).  Since round 4 the kernel that does Canny's hysteresis also makes the
    outer-border keys of the contour stage from its label table (k_frame.h): its merge / flatten / edge-bit phases are Canny,
    its "outer keys" phase belongs to contours+rect+fill (the extremes themselves ride on the edge-bit pass and stay with Canny:
    a lower bound for the contour share)."""
    import glob
    import re
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_frame_profile.txt")), reverse=True):
        try:
            keys = tot = 0.0
            for line in open(path):
                m = re.search(r"k_frame_fg: merge (\d+)/\d+ flatten\+strong (\d+)/\d+ outer keys \| edge bits (\d+)/\d+ edge\+extremes \| write-out (\d+)/\d+", line)
                if m:
                    a, b, c, d = (float(x) for x in m.groups())
                    keys += c
                    tot += a + b + c + d
            if tot > 0 and keys > 0:
                return 1.0 - keys / tot, keys / tot, os.path.basename(path)
        except (OSError, ValueError):
            continue
    return 1.0, 0.0, None


DC_DILATE, DC_CANNY, DC_SRC = dc_split()
FG_CANNY, FG_KEYS, FG_SRC = fg_split()
STAGES = {
    "prep": (5.0, {"k_prep_hist": 1.0, "k_lut": 1.0, "k_removestars": 1.0}),
    "prep+erode": (7.0, {"k_prep_erode": 1.0}),
    # both passes' front ends in one sweep over the float frames: 5N for every frame (bright image) + 7N for the frames the dim
    # pass then works on (priced in main(): the split depends on how many frames the bright pass accepts)
    "prep(bright)+prep+erode(dim)": (None, {"k_prep_dual": 1.0}),
    # (k_bits_erode: the dim pass of lfdmi_detect_batch erodes from the planes the bright pass's sweep left -- the dim
    # conversion, 5N by this table, is part of k_prep_hist's one sweep there and is not charged a second time)
    "erode": (2.0, {"k_morph(erode)": 1.0, "k_bits_erode": 1.0}),
    "dilate": (2.0, {"k_morph(dilate)": 1.0, "k_dilate_canny": DC_DILATE}),
    # Canny = NMS (the Sobel / NMS stages of the fused tile kernel) + hysteresis (candidate-run scan, per-frame union-find,
    # general fallback kernels)
    "canny": (2.0, {"k_dilate_canny": DC_CANNY, "k_canny_nms": 1.0, "k_runs_init(fg)": 1.0, "k_frame_fg": FG_CANNY, "k_runs_merge8": 1.0,
                    "k_runs_flatten(fg)": 1.0, "k_edge_from_cand": 1.0}),
    "contours+rect+fill": (2.0, {"k_runs_init(bg)": 1.0, "k_frame_bg": 1.0, "k_frame_fg": FG_KEYS, "k_frame_keys": 1.0, "k_runs_merge4_bg": 1.0,
                                 "k_runs_flatten(bg)": 1.0, "k_keys": 1.0, "k_extremes": 1.0, "k_rects": 1.0, "k_fill_quads": 1.0}),
    # two images per frame with a detected rectangle, 1N each
    "hough": (2.0, {"k_pixlist": 1.0, "k_hough_vote": 1.0, "k_hough_peaks": 1.0, "k_hough_topk": 1.0, "k_hough_sort": 1.0}),
}
# bytes per pixel charged to ONE kernel when it is the dominant one: its stage's bytes split over the stage's kernels
# so that nothing is counted twice (the fused tile kernel: dilate 2N + the image read of the Canny stage, 1N; the other
# 1N of Canny -- its edge-map output -- belongs to the hysteresis kernels)
KERNEL_BYTES_PER_PX = {
    "k_prep_hist": 5.0, "k_prep_erode": 7.0, "k_prep_dual": None, "k_morph(erode)": 2.0, "k_bits_erode": 2.0, "k_morph(dilate)": 2.0, "k_dilate_canny": 3.0, "k_canny_nms": 1.0,
    "k_runs_init(fg)": 0.25, "k_frame_fg": 0.75, "k_runs_init(bg)": 0.25, "k_frame_bg": 1.0, "k_rects": 0.5, "k_fill_quads": 0.25,
    "k_pixlist": 0.5, "k_hough_vote": 1.25, "k_hough_peaks": 0.25,
}
# timing slot of the library -> the kernels it brackets, as rocprofv3 names them (template variants are averaged, the kernels of a slot summed)
TRAFFIC_KEYS = {"k_morph(dilate)": ["k_morph_rect_v<0"], "k_morph(erode)": ["k_erode_cand", ("k_morph_rect_rows<1", "k_morph_rect_v<1")], "k_canny_nms": ["k_canny_nms_v"],
                "k_dilate_canny": ["k_dc_tiles", "k_dilate_canny_t"], "k_frame_bg": ["k_frame_contours"], "k_prep_dual": ["k_prep_erode<true"],
                "k_prep_erode": ["k_prep_erode<false"]}


def load_traffic(name, cfg):
    """HBM bytes per launch of kernel `name` from the committed PMC passes (profiles/r*_traffic*.json), for this config only."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                tj = json.load(f)
            c = tj["config"]
            if (c.get("workload", "sdss"), c["frames_per_gpu"], c["inflight"], c["lanes"], c["shape"]) != cfg:
                continue
            total = 0
            for keys in TRAFFIC_KEYS.get(name, [name.split("(")[0]]):
                hits = []
                for key in (keys if isinstance(keys, tuple) else (keys,)):   # a tuple: alternative kernels of one step, first present wins
                    hits = [v["hbm_bytes_per_launch"] for k, v in tj["kernels"].items() if k == key or k.startswith(key + "<") or k.startswith(key + ",")
                            or (key.endswith(("<0", "<1", "<false", "<true")) and k.startswith(key))]
                    if hits:
                        break
                if not hits:
                    raise KeyError(keys)
                total += sum(hits) / len(hits)
            return int(total), os.path.basename(path)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def gpu_clock_mhz(index=0):
    """Current shader clock of HIP device `index` in MHz: hwmon's sclk (freq1_input) of the PCI function torch reports for the
    device -- a GPU box shows the whole node's cards in sysfs, so the card is found by its bus id, not by its position --, else
    the active level of its pp_dpm_sclk; None where neither can be read."""
    import glob
    import re
    dev_dir = None
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        bus = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        if os.path.isdir("/sys/bus/pci/devices/" + bus):
            dev_dir = "/sys/bus/pci/devices/" + bus
    except Exception:  # noqa: BLE001
        pass
    if dev_dir is None:
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        if len(cards) != 1:
            return None                            # several cards and no way to tell which one is ours
        dev_dir = cards[0]
    try:
        for lab in glob.glob(dev_dir + "/hwmon/hwmon*/freq*_label"):
            if open(lab).read().strip() == "sclk":
                return int(open(lab.replace("_label", "_input")).read()) // 1000000
    except (OSError, ValueError):
        pass
    try:
        for line in open(dev_dir + "/pp_dpm_sclk"):
            if line.rstrip().endswith("*"):
                return int(re.search(r"(\d+)\s*[Mm][Hh]z", line).group(1))
    except (OSError, AttributeError, ValueError):
        pass
    return None


def host_cores(world):
    """(threads the CPU legs may use, cores the process may run on): this rank's share of the node -- an MI355X node has 8
    GPUs, so 1/8 of its cores per rank (256 / 8 = 32 on the pool's hosts; $LFD_CORES_PER_GPU overrides)."""
    from lfd_amd import usable_cores
    avail = usable_cores()   # (affinity mask cut down to the cgroup's CPU quota: the pool's one-GPU boxes show 256 CPUs and grant 16 cores)
    per_gpu = int(os.environ.get("LFD_CORES_PER_GPU", 0)) or max(1, (os.cpu_count() or avail) // 8)
    return max(1, min(avail, per_gpu * max(1, world))), avail


def detection_quality(res, truths, h):
    """Detections against what lfd_amd.synth injected (the third value of make_frame): a kernel change that stays oracle-equal
    cannot move these, a change of the synthetic recipe that breaks the workload does.  Lines are compared in the flipped
    frame the pipe works on (detecttrails.py:124): a streak through (x0, y0) at angle_deg has theta = 90 deg - angle (mod 180)."""
    with_streak = [i for i, t in enumerate(truths) if t["streak"] != "none"]
    without = [i for i, t in enumerate(truths) if t["streak"] == "none"]
    found = res["found"] > 0
    dth, dist = [], []
    for i in with_streak:
        if not found[i]:
            continue
        t = truths[i]
        th = float(res["theta"][i])
        want = np.deg2rad((90.0 - t["angle_deg"]) % 180.0)
        d = abs(th - want) % np.pi
        dth.append(np.rad2deg(min(d, np.pi - d)))
        dist.append(abs(t["x0"] * np.cos(th) + (h - 1 - t["y0"]) * np.sin(th) - float(res["rho"][i])))
    out = {"frames_with_streak": len(with_streak), "of_those_found": int(found[with_streak].sum()) if with_streak else 0,
           "bright_streaks_found_by_bright_pass": int(sum(1 for i in with_streak if truths[i]["streak"] == "bright" and res["found"][i] == 1)),
           "frames_without_streak": len(without), "of_those_found_false_positives": int(found[without].sum()) if without else 0}
    if dth:
        out["median_abs_dtheta_deg"] = round(float(np.median(dth)), 3)
        out["median_distance_of_injected_point_from_line_px"] = round(float(np.median(dist)), 2)
        out["note"] = "HoughLines at rho = 20 px, theta = 1 deg: a line is known to +-10 px / +-0.5 deg by construction"
    return out


def load_util(workload):
    """Per-kernel utilisation figures derived from the committed SQ / GRBM / TCC counters of the newest profiles/r*_util_<workload>.json
    (tools/make_util.py; formulas in that file): {timing slot: {...}}, the file name, or ({}, None)."""
    import glob
    best = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_util_%s.json" % workload)))
    if not best:
        return {}, None, None
    with open(best[-1]) as f:
        d = json.load(f)
    return d.get("slots", {}), os.path.basename(best[-1]), d.get("top_kernel")


def stress_leg(args, env, headline):
    """The full pipe on the stress workloads of lfd_amd.synth.STRESS (crowded fields, noisier sky, a saturated star, one crowded
    frame in a quiet chunk): frames/s for a 256-frame batch (16 distinct frames, each 16 times), the share of the headline rate,
    and what the context did besides the fast path.  tests/test_gpu_stress.py checks the same workloads against the oracle."""
    import torch
    from lfd_amd import _native, synth
    from lfd_amd.batch import BatchDetector
    from lfd_amd.detecttrails import default_params
    dev, dev_index = env["dev"], env["dev_index"]
    pb, pd, prs = default_params()
    rs = _native.make_rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})
    n, nd = 256, 16
    out = {"batch": n, "distinct_frames": nd, "note": "secondary leg, never `value`; synchronous calls; percent = of this run's rate on the benchmark's sky with synchronous calls (`one_call_in_flight`, the headline when calls_in_flight = 1)"}
    for name in synth.STRESS:
        host, cats = synth.make_frames(0, nd, synth.SDSS_SHAPE, min(16, host_cores(1)[0]), True, synth.stress_recipes(name, nd))
        idx = torch.arange(n, device=dev) % nd
        frames = torch.from_numpy(host).to(dev)[idx].contiguous()
        packed = synth.pack_catalogs([cats[i % nd] for i in range(n)])
        cat = {k: torch.from_numpy(v).to(dev) for k, v in packed.items()}
        det = BatchDetector(dev_index, synth.SDSS_SHAPE, n, stream=torch.cuda.current_stream().cuda_stream)
        work = frames.clone()
        res = det.detect(work, pb, pd, cat, rs)                  # first call: tables grow, kernels switch
        torch.cuda.synchronize()
        first = det.stats()
        best = None
        for _ in range(3):
            work.copy_(frames)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = det.detect(work, pb, pd, cat, rs)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        st = det.stats()
        cnt = det.get_counters()[:n]
        out[name] = {"frames_per_s": round(n / best, 1), "percent_of_headline": round(100.0 * (n / best) / headline, 1),
                     "ms_per_batch": round(1e3 * best, 2), "found": int((res["found"] > 0).sum()),
                     "max_candidate_runs_per_frame": int(cnt[:, 12].max()), "catalogue_objects_per_frame": int(packed["count"].max()),
                     "first_call": {k: v for k, v in first.items() if v and k not in ("chunks", "scan_fused_on")},
                     "per_call_after_that": {k: (st[k] - first[k]) // 3 for k in st if k not in ("chunks", "scan_fused_on") and st[k] != first[k]}}
        det.close()
        del frames, work, cat
        torch.cuda.empty_cache()
    return out


def measure(args, workload, n, steps, warmup, env, distinct=None, host_leg=True, sustained_s=0.0):
    """One workload on this rank's GPU: generation, upload, warm-up, the timed region, the per-kernel table.  Returns
    (out dict for rank 0 or None, state for the follow-up legs)."""
    import torch
    import torch.distributed as dist
    from lfd_amd import _native, synth
    from lfd_amd.batch import BatchDetector
    from lfd_amd.detecttrails import default_params
    rank, world, dev, dev_index, use_dist, red_dev, share = (env[k] for k in ("rank", "world", "dev", "dev_index", "use_dist", "red_dev", "share"))
    lsst = workload == "lsst"
    shape = synth.LSST_SHAPE if lsst else synth.SDSS_SHAPE
    inflight = args.inflight or n
    k0 = rank * n
    workers = args.gen_workers
    if workers < 0:
        workers = max(1, min(32, host_cores(world)[0] // max(1, world)))
    t0 = time.time()
    # child processes (never forks of this one: safe under a profiler's preloaded library), frames into shared memory
    nd = min(n, distinct or n)
    host, cats, truths = synth.make_frames(k0 if nd == n else 0, nd, shape, workers, with_catalog=not lsst, with_truth=True)
    t_gen = time.time() - t0

    pb, pd, prs = default_params()
    rs = _native.make_rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})
    h, w = shape
    rhos = [20.0, 10.0, 5.0]
    if lsst:
        pd = dict(pd, erodeKernel=np.ones((9, 9), np.uint8))
    dframes = torch.from_numpy(host).to(dev)
    if nd < n:                                     # frame i of the batch is distinct frame i % nd (said in config.workload)
        idx = torch.arange(n, device=dev) % nd
        dframes = dframes[idx].contiguous()
        cats = [cats[i % nd] for i in range(n)]
        truths = [truths[i % nd] for i in range(n)]
    cat = packed = None
    if not lsst and not args.no_removestars:
        packed = synth.pack_catalogs(cats)
        cat = {k: torch.from_numpy(v).to(dev) for k, v in packed.items()}
    stream = torch.cuda.current_stream().cuda_stream
    cif = max(1, args.calls_in_flight) if args.lanes == 1 and not args.host_frames else 1
    det = BatchDetector(dev_index, (h, w), inflight, stream=stream, lanes=args.lanes, calls_in_flight=cif)

    if lsst:
        def run(frames, _cat):
            return det.multiscale(frames, pd, rhos, dim=True, flip=True)
    else:
        def run(frames, c):
            return det.detect(frames, pb, pd, c, rs)

    def step_dev():
        return run(dframes, cat)

    def step_async():                              # calls_in_flight > 1: the call as a future (BatchDetector.detect_async)
        if lsst:
            return det.multiscale_async(dframes, pd, rhos, dim=True, flip=True)
        return det.detect_async(dframes, pb, pd, cat, rs)

    def run_steps(k):
        # k steps; with several calls in flight they are all submitted (a host thread per context takes its own in order) and
        # collected in order: step i + 1 is queued on the GPU while step i runs
        if cif > 1:
            futs = [step_async() for _ in range(k)]
            return [f.result() for f in futs][-1]
        r = None
        for _ in range(k):
            r = step()
        return r

    def step_host():
        return run(host, None if (lsst or args.no_removestars) else packed)

    host_ok = nd == n
    step = step_host if (args.host_frames and host_ok) else step_dev

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # Per-launch HIP events cost ~6 % of a step when every one of its ~40 launches is bracketed, so inside
    # the timed region only the few largest kernels are (those that move pixels; the warm-up steps, fully
    # bracketed, nominate two -- a cold first launch can distort a single winner, and four candidates' brackets measured
    # 0.06-0.09 ms of a 4.2 ms step against an unbracketed run, profiles/README.md round-4 log); the dominant one is then
    # chosen from the timed region's own sums.  The per-kernel table of the JSON line comes from one extra,
    # fully bracketed step AFTER the timed region.
    res = None
    det.enable_timing(not args.no_kernel_timing)
    for c in det.ctxs[1:] if cif > 1 else ():      # (every further context's first call -- lazy allocations -- before the warm-up)
        if lsst:
            c.process_multiscale(dframes, pd, rhos, True, True, False)
        else:
            c.detect_batch(dframes, pb, pd, cat, rs)
    res = run_steps(warmup)
    torch.cuda.synchronize()
    warm = det.get_timing()
    cands = sorted((k for k, v in warm.items() if v[1] and KERNEL_BYTES_PER_PX.get(k, 0.0) != 0.0), key=lambda k: -warm[k][0])[:2]
    det.timing_select(cands or ["k_dilate_canny", "k_prep_hist"])
    det.enable_timing(not args.no_kernel_timing)  # (re-arms and clears the sums)
    fence()
    t0 = time.perf_counter()
    res = run_steps(steps)
    fence()
    elapsed = time.perf_counter() - t0
    timing = det.get_timing()
    one_call = None
    if cif > 1:                                   # beside it: the same steps one synchronous call after the other
        det.enable_timing(False)
        fence()
        t1 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        one_call = time.perf_counter() - t1
    det.timing_select(None)
    det.enable_timing(not args.no_kernel_timing)
    step()                                        # untimed: the per-kernel table
    torch.cuda.synchronize()
    table = det.get_timing()
    det.enable_timing(False)
    cnt = det.get_counters()[:n]  # last pass of every slot: Hough cost is 180 votes per non-zero pixel
    sustained = None
    if sustained_s > 0:                           # the same step back to back for >= sustained_s seconds: the clock the chip holds
        import threading
        clocks, stop_sampling = [], threading.Event()

        def sample():                             # (read while the loop runs: the clock falls back within milliseconds of an idle GPU)
            while not stop_sampling.wait(0.2):
                c = gpu_clock_mhz(dev_index)
                if c:
                    clocks.append(c)

        sampler = threading.Thread(target=sample, daemon=True)
        fence()
        sampler.start()
        t0 = time.perf_counter()
        m = 0
        while time.perf_counter() - t0 < sustained_s:
            run_steps(10)
            torch.cuda.synchronize()
            m += 10
        dt = time.perf_counter() - t0
        stop_sampling.set()
        sampler.join()
        sustained = {"value": round(world * n * m / dt, 2), "unit": "frames/s", "steps": m, "seconds": round(dt, 2),
                     "ms_per_step": round(1e3 * dt / m, 3),
                     "sclk_mhz_during": {"samples": len(clocks), "min": min(clocks), "median": int(np.median(clocks)), "max": max(clocks)} if clocks else None,
                     "note": "the timed step repeated back to back (this rank); sclk = hwmon freq1_input of this device's PCI function, sampled every 0.2 s while the loop runs"}
    two_calls = None
    if sustained_s > 0 and cif == 1 and args.lanes == 1 and not args.host_frames:
        # secondary: the same step with two calls in flight (no HIP events: with a queue that never drains they no longer bracket
        # single kernels, which is why the headline's timed region keeps synchronous calls)
        det2 = BatchDetector(dev_index, (h, w), inflight, stream=stream, calls_in_flight=2)
        try:
            def sub2():
                return det2.multiscale_async(dframes, pd, rhos, dim=True, flip=True) if lsst else det2.detect_async(dframes, pb, pd, cat, rs)
            for f in [sub2() for _ in range(4)]:
                r2 = f.result()
            m2 = max(steps, 40)
            fence()
            t0 = time.perf_counter()
            for f in [sub2() for _ in range(m2)]:
                r2 = f.result()
            fence()
            dt2 = time.perf_counter() - t0
            fence()
            t0 = time.perf_counter()
            for _ in range(m2):
                step()
            fence()
            dt1 = time.perf_counter() - t0
            two_calls = {"value": round(world * n * m2 / dt2, 2), "unit": "frames/s", "ms_per_step": round(1e3 * dt2 / m2, 3), "steps": m2,
                         "synchronous_calls_beside_it": {"value": round(world * n * m2 / dt1, 2), "ms_per_step": round(1e3 * dt1 / m2, 3)},
                         "records_equal_the_synchronous_call": bool(r2.tobytes() == res.tobytes()),
                         "note": "BatchDetector(calls_in_flight=2).detect_async: two contexts (a workspace each) launching into one stream, a host "
                                 "thread each; this rank only; secondary, never `value`"}
        finally:
            det2.close()
    host_res = None
    if host_leg and host_ok and not args.host_frames and not args.no_host_leg:  # secondary: frames handed over as host buffers (PCIe-inclusive; never `value`)
        step_host()
        fence()
        t0 = time.perf_counter()
        m = max(1, min(steps, 3))
        for _ in range(m):
            step_host()
        fence()
        host_res = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([host_res], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            host_res = float(t.item())
        host_res = (world * n * m / host_res, m)
    if use_dist:
        t = torch.tensor([elapsed, one_call or 0.0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        one_call = float(t[1].item()) or None
    one_call_el = one_call

    state = {"host": host, "cats": cats, "res": res, "pb": pb, "pd": pd, "prs": prs, "rhos": rhos, "lsst": lsst, "n": n, "nd": nd}
    out = None
    if rank == 0:
        nnz_equ, nnz_box = cnt[:, 15][cnt[:, 8] > 0], cnt[:, 16][cnt[:, 8] > 0]
        res0 = res[0] if lsst else res       # lsst: [scale, frame]
        found_b = int((res0["found"] == 1).sum())
        found_d = int((res0["found"] == 2).sum())
        errors = int((res["status"] != 0).sum())
        total_frames = world * n * steps
        value = total_frames / elapsed
        N = h * w
        # dominant kernel by device time inside the timed region
        if not any(v[1] for v in timing.values()):
            timing = {"misc": (1e-9, 1, 1)}
        name, (ms, launches, units) = max(timing.items(), key=lambda kv: kv[1][0])
        dim_share = 1.0 - found_b / float(n) if not lsst else 1.0   # frames of a chunk the dim pass works on
        dual_bpp = 5.0 + 7.0 * dim_share                            # k_prep_dual: 5N per frame + 7N per frame that goes on to the dim pass
        bpp_dom = KERNEL_BYTES_PER_PX.get(name, 0.0)
        if bpp_dom is None:
            bpp_dom = dual_bpp
        bytes_per_frame = bpp_dom * N
        achieved = (bytes_per_frame * units) / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        avg_ms = ms / max(1, launches)
        traffic, traffic_src = load_traffic(name, (workload, n, inflight, args.lanes, [h, w]))
        kern = {k: {"ms_per_step": round(v[0], 4), "launches_per_step": v[1], "frames_per_step": v[2]}
                for k, v in table.items() if v[1]}  # one fully bracketed step after the timed region
        # what bounds each kernel, from the committed SQ / GRBM / TCC counter passes of the same command (tools/make_util.py): shares of
        # the vector pipe, the scalar unit, the LDS and the HBM peak, occupancy, parked / issue-stalled shares of a wave's life
        util, util_src, top_rocprof = load_util(workload)
        for k in kern:
            u = util.get("k_scan_fused" if k.startswith("k_runs_init") else k)
            if u:
                kern[k]["util"] = u
        # per-stage view of that step: every stage's algorithmic bytes once, over the summed time of its kernels
        stages = {}
        for sname, (bpp, ks) in STAGES.items():
            t_ms = sum(table[k][0] * sh for k, sh in ks.items() if k in table and table[k][1])
            if t_ms <= 0:
                continue
            lead = max((k for k in ks if k in table and table[k][1]), key=lambda k: table[k][0] * ks[k])
            frames_st = table[lead][2]  # (Hough: frames with a rectangle, summed over the launches of every scale)
            gb = (dual_bpp if bpp is None else bpp) * N * frames_st / 1e9
            frac = gb / (t_ms * 1e-3) / HBM_PEAK_GBPS
            stages[sname] = {"ms_per_step": round(t_ms, 4), "algorithmic_GB": round(gb, 4), "GBps": round(gb / (t_ms * 1e-3), 1),
                             "frac_of_peak": round(min(frac, 1.0), 4)}
            if frac > 1.0:  # the stage's kernels skip empty tiles / bands: the algorithmic bytes were never moved, so this is no bandwidth figure
                stages[sname]["sparse"] = "algorithmic bytes not moved (activity-driven kernels skip empty regions): uncapped %.2f" % frac
            if sname == "prep" and "k_removestars" in table and table["k_removestars"][1] >= 2:
                # (two timed spans per step: k_rs_boxes in front of the sweep, the zero fill on a side stream beside k_frame_bg)
                stages[sname]["overlapped"] = ("k_removestars' zero fill runs on a side stream beside k_frame_bg (the sweep masks the squares on load); "
                                               "its time is still charged to this stage, and stretches k_frame_bg's")
        ch = None
        if "canny" in stages and "hough" in stages:  # what north_star asks for: Canny (2N) + Hough (1N per image) over their kernels
            gb = stages["canny"]["algorithmic_GB"] + stages["hough"]["algorithmic_GB"]
            t_ms = stages["canny"]["ms_per_step"] + stages["hough"]["ms_per_step"]
            t_lo = t_ms + sum(table[k][0] * DC_DILATE for k in ("k_dilate_canny",) if k in table)  # the whole fused tile kernel charged to Canny
            ch = {"algorithmic_GB": round(gb, 4), "ms_per_step": round(t_ms, 4), "GBps": round(gb / (t_ms * 1e-3), 1),
                  "frac": round(gb / (t_ms * 1e-3) / HBM_PEAK_GBPS, 4),
                  "frac_if_fused_tile_kernel_is_all_canny": round(gb / (t_lo * 1e-3) / HBM_PEAK_GBPS, 4),
                  "fused_tile_kernel_split": {"dilate": round(DC_DILATE, 3), "canny": round(DC_CANNY, 3), "source": DC_SRC},
                  "k_frame_fg_split": {"hysteresis": round(FG_CANNY, 3), "contour_keys": round(FG_KEYS, 3), "source": FG_SRC},
                  "target": 0.5}
        hv = None
        if "k_hough_vote" in table and table["k_hough_vote"][1]:
            # SURVEY 8(d)'s secondary bound: 180 votes per non-zero pixel of each image against the LDS-atomic / vector-issue peaks.
            # The counters hold the LAST pass of every slot, so the votes are priced over the vote launches of that pass only:
            # scale the time by the share of the launches' frames the counted frames make up.
            numangle = 180
            votes = float(numangle) * float(nnz_equ.sum() + nnz_box.sum()) * (len(rhos) if lsst else 1)
            t_all, _, fr_all = table["k_hough_vote"]
            per_scale = max(1, len(nnz_equ))
            t_cnt = t_all * min(1.0, per_scale * (len(rhos) if lsst else 1) / max(1.0, float(fr_all)))
            if votes > 0 and t_cnt > 0:
                vps = votes / (t_cnt * 1e-3)
                hv = {"votes_per_s": float("%.4g" % vps), "votes": int(votes), "ms": round(t_cnt, 4),
                      "lds_atomic_peak_votes_per_s": LDS_ATOMIC_VOTES_PER_S, "frac_of_lds_atomic_peak": round(vps / LDS_ATOMIC_VOTES_PER_S, 4),
                      "valu_peak_votes_per_s": VALU_VOTES_PER_S, "frac_of_valu_peak": round(vps / VALU_VOTES_PER_S, 4),
                      "note": "180 votes per non-zero pixel of equ and box_img (frames whose LAST pass reached HoughLines) over k_hough_vote's "
                              "time for those frames; peaks: one ds_add per vote at 16 per clock per CU, >= 4 vector lane-operations per vote on "
                              "4 x SIMD-32 per CU, 256 CUs at 2.4 GHz.  The kernel votes per run of pixels (two adds per run and angle), which is "
                              "how it can exceed the per-vote LDS bound"}
        if lsst:
            metric = "LSST-scale frames/sec (4096x4096) dim pass, 9x9 erosion, multi-scale Hough"
            workload_s = ("configs[4]: dim pass with 9x9 erosion + HoughLines at rho 20/10/5, batch=%d synthetic 4096x4096 float32 frames "
                          "per GPU%s, %s" % (n, "" if nd == n else " (%d distinct frames, each %d times)" % (nd, n // nd),
                                             "HOST-resident (PCIe-inclusive)" if args.host_frames else "device-resident"))
        else:
            metric = "SDSS frames/sec (2048x1489) full detecttrails pipe"
            workload_s = ("configs[2]: full removestars+bright+dim pipe, batch=%d synthetic SDSS 2048x1489 float32 frames per GPU%s, %s"
                          % (n, "" if nd == n else " (%d distinct frames, each %d times)" % (nd, n // nd),
                             "HOST-resident (PCIe-inclusive)" if args.host_frames else "device-resident"))
        out = {
            "metric": metric, "value": round(value, 2),
            "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(1e3 * elapsed / steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_s,
                       "frames_per_gpu": n, "inflight": inflight, "lanes": args.lanes, "shape": [h, w],
                       "calls_in_flight": cif,
                       "calls_in_flight_note": ("BatchDetector(calls_in_flight=%d): %d contexts (a workspace each) launching into one stream, a host thread "
                                                "each; step i + 1 is queued while step i runs, the steps still execute one after the other and "
                                                "return the same records; `one_call_in_flight` is the same loop with synchronous calls" % (cif, cif)) if cif > 1 else None,
                       "removestars": (not lsst) and not args.no_removestars, "parallelism": "frame-parallel x%d" % world + (" (rehearsal: ranks share %d GPU(s), gloo)" % torch.cuda.device_count() if share else ""),
                       "found_bright": found_b, "found_dim": found_d, "frame_errors": errors,
                       "hough_rhos": rhos if lsst else [20.0],
                       "hough_nnz_equ_median": int(np.median(nnz_equ)) if len(nnz_equ) else 0,
                       "hough_nnz_box_median": int(np.median(nnz_box)) if len(nnz_box) else 0,
                       "workspace_GB": round(det.workspace_bytes() / 1e9, 2), "frames_spilled_to_worst_case_workspace": det.spill_count(),
                       "beside_the_fast_path": {k: v for k, v in det.stats().items() if k != "spilled_frames"},
                       "detection_vs_injected_truth": detection_quality(res0, truths, h),
                       "gen_s": round(t_gen, 1)},
            "one_call_in_flight": ({"value": round(total_frames / one_call_el, 2), "unit": "frames/s", "ms_per_step": round(1e3 * one_call_el / steps, 3),
                                    "steps": steps} if one_call_el else None),
            "two_calls_in_flight": two_calls,
            "roofline": {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                         "frac_kind": "ALGORITHMIC bytes (SURVEY 8d) / kernel time / peak -- not HBM utilisation; see measured_traffic_frac",
                         "traffic": traffic,
                         "traffic_source": traffic_src,
                         "measured_traffic_frac": round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5) if traffic and avg_ms > 0 else None,
                         "frac_of_measured_copy_6290": round(achieved / 6290.0, 5),
                         "avg_launch_ms": round(avg_ms, 4),
                         "frames_per_launch": round(units / max(1, launches), 2),
                         "algorithmic_bytes_per_frame": bytes_per_frame,
                         "algorithmic_bytes_per_px": round(bpp_dom, 3),
                         "canny_hough": ch,
                         "hough_votes": hv,
                         "kernel_is": "a timing SLOT of the library (the fused tile kernel's three launches); the largest single rocprof kernel is kernel_rocprof",
                         "kernel_rocprof": top_rocprof,
                         "util_source": util_src,
                         "note": "algorithmic bytes as SURVEY 8(d) prescribes, each stage charged once across its kernels; "
                                 "measured_traffic_frac = PMC HBM bytes per launch / launch time / peak: a kernel whose measured "
                                 "fraction is far below 1 is limited by vector-unit issue / LDS round trips, not by HBM"},
            "stages": stages,
            "kernels": kern,
        }
        if host_res:
            out["host_resident"] = {"value": round(host_res[0], 2), "unit": "frames/s", "steps": host_res[1],
                                    "note": "same step with the frames handed over as host buffers (PCIe-inclusive); never `value`"}
        if sustained:
            out["sustained"] = sustained
    det.close()
    del dframes
    torch.cuda.empty_cache()
    return out, state


def cpu_baseline(state, args, world):
    """The oracle (kind "port") on a bounded sample of rank 0's frames: one thread, all of the rank's cores; plus the
    reference's own cv2 call sequence where OpenCV can be imported."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import cv2_path, lfd_oracle as O
    host, cats, res, pb, pd, prs, rhos, lsst, n = (state[k] for k in ("host", "cats", "res", "pb", "pd", "prs", "rhos", "lsst", "n"))
    n = min(n, state["nd"])
    cpu_sample = args.cpu_sample if args.cpu_sample is not None else (2 if lsst else 24)
    if cpu_sample <= 0:
        return None
    rs_o = O.rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})

    def cpu_frame(i):
        if lsst:
            rec = [O.process_dim(host[i].copy(), dict(pd, houghMethod=r), flip=True) for r in rhos]
            return all(all(rr[k] == res[s][i][k].item() for k in rr) for s, rr in enumerate(rec))
        r = O.detect_frame(host[i].copy(), pb, pd, None if args.no_removestars else cats[i], rs_o)
        return all(r[k] == res[i][k].item() for k in r)

    m = min(cpu_sample, n)
    t0 = time.perf_counter()
    agree = sum(cpu_frame(i) for i in range(m))
    dt = time.perf_counter() - t0
    cores, avail = host_cores(world)
    cores = max(1, min(cores, n))
    what = "3 x process_dim (rho 20/10/5)" if lsst else "detect_frame"
    out = {"value": round(m / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "first %d frames of rank 0's batch through oracle/ (C, -O2, 1 thread; %s), "
                     "%d/%d identical to the GPU records" % (m, what, agree, m),
           "host_cores_available": avail}
    # all of the rank's cores: one oracle call per thread (ctypes releases the GIL), a bounded sample again
    per = max(1, int(round(m / dt * 8.0)))               # ~8 s of work per core
    ma = min(n, cores * per)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        agree_a = sum(ex.map(cpu_frame, range(ma)))
    dta = time.perf_counter() - t0
    out["all_cores"] = {"value": round(ma / dta, 3), "unit": "frames/s", "cores": cores,
                        "sample": "%d frames on %d threads (this rank's share of the host: 1/8 of its cores per GPU), "
                                  "%d/%d identical to the GPU records" % (ma, cores, agree_a, ma)}
    # the reference's own OpenCV path, where OpenCV exists
    cv2 = cv2_path.load()
    if cv2 is None or lsst:
        out["opencv"] = "unavailable" if cv2 is None else "not timed for this workload"
        if cv2 is None:
            out["opencv_note"] = "`import cv2` fails on this host (tried at run time): the reference's OpenCV path cannot be timed; kind stays 'port'"
    else:
        from lfd_amd.detecttrails import check_theta
        mo = min(8, n)
        t0 = time.perf_counter()
        same = 0
        for i in range(mo):
            found, rho, theta = cv2_path.detect_frame(cv2, host[i].copy(), pb, pd, None if args.no_removestars else cats[i], rs_o,
                                                      O.remove_stars, check_theta)
            same += int(found == res[i]["found"].item() and np.float32(rho) == res[i]["rho"] and np.float32(theta) == res[i]["theta"])
        dto = time.perf_counter() - t0
        out["opencv"] = {"value": round(mo / dto, 3), "unit": "frames/s", "kind": "reference-equivalent cv2 call sequence (oracle/cv2_path.py)",
                         "version": cv2.__version__, "threads": int(cv2.getNumThreads()),
                         "sample": "%d frames, %d/%d with the GPU's (found, rho, theta)" % (mo, same, mo)}
    return out


def dropin_leg(state, args, dev_index):
    """DetectTrails(run=...).process(batch=256) over a synthetic $BOSS tree in /dev/shm: the drop-in itself, FITS reading
    included.  Plain .fits: the batch's 256 frames written once and hard-linked to a run of --dropin-frames fields (so that
    the one-off set-up -- context, staging buffers -- does not dominate); then a bounded sample as .fits.bz2.  Rows compared
    with the device-resident records."""
    import shutil
    import tempfile
    from lfd_amd import results as results_io, synth
    from lfd_amd.detecttrails import DetectTrails
    host, cats, res, n = state["host"], state["cats"], state["res"], min(state["n"], state["nd"])
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    root = tempfile.mkdtemp(prefix="lfd_boss_", dir=base)
    out = {}
    old_env = {k: os.environ.get(k) for k in ("BOSS_PHOTOOBJ", "PHOTO_REDUX", "LFD_DEVICE")}
    try:
        os.environ["LFD_DEVICE"] = str(dev_index)
        for label, distinct, total, bz in (("plain", n, max(n, args.dropin_frames), False),
                                           ("bz2", min(n, args.dropin_bz2), args.dropin_bz2_frames, True)):
            if distinct <= 0:
                continue
            total = max(total, distinct)
            tree = os.path.join(root, label)
            t0 = time.perf_counter()
            hdr = synth.write_boss_tree(tree, host[:distinct], cats[:distinct], run=94, camcol=1, filter="r", field0=100, bz2_all=bz,
                                        link_to=total)
            t_write = time.perf_counter() - t0
            save = os.path.join(tree, "out")
            os.makedirs(save)
            dt = DetectTrails(run=94, camcol=1, filter="r", savepath=save)
            t0 = time.perf_counter()
            dt.process(batch=256)
            el = time.perf_counter() - t0
            st = dt.last_stats
            rows = [ln.strip() for ln in open(dt.results) if ln.strip()]
            want = [results_io.format_result_row(94, 1, "r", 100 + i, hdr, res[i % distinct]) for i in range(total) if res[i % distinct]["found"]]
            errs = open(dt.errors).read().count("\n\n")
            done = st["chunk_done_s"]
            steady = None
            if len(done) >= 3:                     # chunks after the first, from the moment the first one was done
                steady = round((total - st["chunk_frames"]) / (done[-1] - done[0]), 1)
            out[label] = {"value": round(total / el, 1), "unit": "frames/s", "frames": total, "distinct_frames": distinct, "seconds": round(el, 3),
                          "setup_s": round(st["setup_s"], 3), "steady_state_frames_per_s": steady, "frames_per_gpu_call": st["chunk_frames"],
                          "rows": len(rows), "rows_equal_device_resident_run": rows == want, "errors_logged": errs,
                          "tree_write_s": round(t_write, 1)}
            if bz:
                # .fits.bz2 frames are decompressed on the GPU, a chunk's files at once (lfdmi_bz2_decode_batch); beside it: what the
                # host's cores do with the same files (the reference's way: one bunzip2 per frame), measured on a bounded sample
                from lfd_amd import usable_cores
                from lfd_amd.detecttrails import bz2blocks, sdssfiles
                cores = usable_cores()
                out[label]["decoded_on"] = st.get("bz2", {})
                blob = open(sdssfiles.filename("frame", run=94, camcol=1, field=100, filter="r") + ".bz2", "rb").read()
                out[label]["compressed_MB_per_frame"] = round(len(blob) / 1e6, 2)
                import bz2 as _bz2
                t0 = time.perf_counter(); a = _bz2.decompress(blob); t_whole = time.perf_counter() - t0
                bz2blocks.decompress(blob, bz2blocks.shared_pool())
                t0 = time.perf_counter(); b = bz2blocks.decompress(blob, bz2blocks.shared_pool()); t_blocks = time.perf_counter() - t0
                out[label]["host_decoder"] = {"usable_cores": cores, "frames_per_s_per_core": round(1.0 / t_whole, 2),
                                              "frames_per_s_all_cores_estimate": round(cores / t_whole, 1),
                                              "one_frame_latency_s": {"whole_file_one_core": round(t_whole, 3),
                                                                      "blocks_side_by_side": round(t_blocks, 3), "equal": a == b,
                                                                      "threads": min(16, cores)}}
                # the decoder alone, the same files already in host memory (no FITS parsing, no detection)
                try:
                    from lfd_amd import _native as Nv
                    import numpy as np
                    m = min(256, total)
                    blobs = [open(sdssfiles.filename("frame", run=94, camcol=1, field=100 + (i % distinct), filter="r") + ".bz2", "rb").read()
                             for i in range(distinct)]
                    offs, cur = [], 0
                    for i in range(m):
                        offs.append(cur)
                        cur += (len(blobs[i % distinct]) + 255) & ~255
                    srcb = np.zeros(cur, np.uint8)
                    for i in range(m):
                        srcb[offs[i]:offs[i] + len(blobs[i % distinct])] = np.frombuffer(blobs[i % distinct], np.uint8)
                    with Nv.Bz2Decoder(dev_index) as z:
                        lens = [len(blobs[i % distinct]) for i in range(m)]
                        z.decode(srcb, offs, lens, len(a) + (1 << 20))
                        t0 = time.perf_counter()
                        ol, stt, _ = z.decode(srcb, offs, lens, len(a) + (1 << 20))
                        td = time.perf_counter() - t0
                        same = z.fetch(m - 1, 0, int(ol[m - 1])).tobytes() == _bz2.decompress(blobs[(m - 1) % distinct])
                        out[label]["device_decoder_alone"] = {"frames": m, "frames_per_s": round(m / td, 1), "ms": {k: round(v, 1) for k, v in z.timings().items()},
                                                              "all_ok": bool((stt == 0).all()), "last_file_equals_python_bz2": bool(same)}
                except Exception as e:  # noqa: BLE001
                    out[label]["device_decoder_alone"] = {"error": repr(e)}
            shutil.rmtree(tree, ignore_errors=True)
    finally:
        for k, v in old_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        shutil.rmtree(root, ignore_errors=True)
    out["note"] = ("DetectTrails(run=94, camcol=1, filter='r').process(batch=256), end to end (value) and without the one-off set-up "
                   "and first chunk (steady_state): frame FITS + photoObj FITS read from /dev/shm by the loader pool straight into pinned "
                   "staging memory, big-endian floats swapped on the device; PCIe-inclusive by nature; never `value` of the bench line")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=("sdss", "lsst"), default="sdss")
    ap.add_argument("--frames-per-gpu", type=int, default=None, help="default 256 (configs[2]: 256 per GPU; configs[4]: 2048 over 8 GPUs)")
    ap.add_argument("--inflight", type=int, default=None, help="frames per launch (default: the whole batch)")
    ap.add_argument("--lanes", type=int, default=1, help="concurrent half-batches (streams) per GPU")
    ap.add_argument("--calls-in-flight", type=int, default=1,
                    help="BatchDetector(calls_in_flight=): contexts launching into one stream, a host thread each, step i + 1 queued while step i "
                         "runs.  Default 1 (synchronous calls: HIP events bracket single kernels only then); the default run reports the rate with 2 "
                         "as the secondary `two_calls_in_flight`")
    ap.add_argument("--cpu-sample", type=int, default=None, help="frames timed through the CPU oracle on one thread (0 = skip)")
    ap.add_argument("--gen-workers", type=int, default=-1)
    ap.add_argument("--no-removestars", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="developer: leave the per-launch HIP events off (no roofline entry) to see what they cost")
    ap.add_argument("--host-frames", action="store_true",
                    help="developer: time ONLY the host-resident (PCIe-inclusive) path as the headline value of this run")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the secondary PCIe-inclusive measurement")
    ap.add_argument("--no-secondary", action="store_true", help="skip the sustained / dropin / lsst legs of the default invocation (profiling runs)")
    ap.add_argument("--sustained-s", type=float, default=3.0)
    ap.add_argument("--dropin-bz2", type=int, default=64, help="distinct frames of the .bz2 sample of the dropin leg (0 = skip)")
    ap.add_argument("--dropin-bz2-frames", type=int, default=4096, help="fields of the .bz2 run (hard links to the distinct files)")
    ap.add_argument("--dropin-frames", type=int, default=4096, help="fields of the plain-FITS run of the dropin leg (hard links of the batch's frames)")
    ap.add_argument("--lsst-distinct", type=int, default=64, help="distinct frames of the secondary lsst leg (each used 256 / this times)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        args.gpus = world

    import torch
    import torch.distributed as dist

    # One rank per GPU.  Rehearsal on a box with fewer GPUs than ranks (LFD_BENCH_SHARE_GPU=1): ranks share devices round-robin
    # and synchronise over gloo (RCCL refuses two ranks on one device) -- the sharding, timing and reduction code is the same.
    share = os.environ.get("LFD_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % max(1, torch.cuda.device_count()) if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or os.environ.get("LFD_BENCH_FORCE_DIST") == "1"  # the env switch lets a 1-GPU box rehearse the RCCL path
    red_dev = torch.device("cpu") if share else dev
    ranks_seen = None
    if use_dist:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        # every rank reports (rank, device it runs on): rank 0 prints what the collective backend really connected
        mine = torch.tensor([rank, dev_index, torch.cuda.device_count()], dtype=torch.int64, device=red_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        seen = [tuple(int(x) for x in t.cpu()) for t in allr]
        ranks_seen = {"backend": dist.get_backend(), "world_size": world, "ranks": sorted(s[0] for s in seen),
                      "devices": [s[1] for s in sorted(seen)], "visible_devices_per_rank": sorted({s[2] for s in seen})}
    env = {"rank": rank, "world": world, "dev": dev, "dev_index": dev_index, "use_dist": use_dist, "red_dev": red_dev, "share": share}

    n = args.frames_per_gpu or 256
    primary = world == 1 and args.workload == "sdss" and not args.host_frames and not args.no_secondary
    out, state = measure(args, args.workload, n, args.steps, args.warmup, env,
                         sustained_s=args.sustained_s if primary else 0.0)
    if rank == 0:
        if ranks_seen:
            out["rccl_ranks_seen" if ranks_seen["backend"] == "nccl" else "ranks_seen"] = ranks_seen
        if world > 1:
            out["cpu_baseline"] = None  # (the host baseline is timed at N = 1 only: rank 0 would keep the other ranks waiting)
        else:
            out["cpu_baseline"] = cpu_baseline(state, args, world)
        if primary:
            try:
                out["dropin"] = dropin_leg(state, args, dev_index)
            except Exception as e:  # noqa: BLE001 - a secondary leg must not cost the headline line
                out["dropin"] = {"error": "%s: %s" % (type(e).__name__, e)}
            del state
            try:
                out["stress"] = stress_leg(args, env, (out.get("one_call_in_flight") or out)["value"])
            except Exception as e:  # noqa: BLE001
                out["stress"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                a2 = argparse.Namespace(**vars(args))
                a2.inflight = None
                a2.cpu_sample = 0
                nl = 256
                sec, _ = measure(a2, "lsst", nl, max(1, min(args.steps, 5)), 1, env, distinct=max(1, min(nl, args.lsst_distinct)), host_leg=False)
                out["lsst"] = {k: sec[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "config", "roofline", "stages", "kernels")}
            except Exception as e:  # noqa: BLE001
                out["lsst"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
