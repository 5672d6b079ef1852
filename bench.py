#!/usr/bin/env python3
"""Headline benchmark: frames/s of the detecttrails hot path on N MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 1                      # BASELINE configs[2] (default)
    python bench.py --workload lsst                                     # BASELINE configs[4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

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
cpu_baseline: the C oracle (oracle/, a port of the reference's algorithm; the reference's own OpenCV path cannot
run here or on the GPU box) on a bounded sample of the same frames: one host thread, and all host cores.
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

# algorithmic bytes per pixel of every stage of a pass, and the kernels (timing slots of the library)
# that make up the stage.  A kernel that spans two stages appears with a share of each (bytes, fraction of its time).
#   prep 5N | erode 2N | dilate 2N | Canny 2N | contours+rect+fill 2N | Hough 1N per image
STAGES = {
    "prep": (5.0, {"k_prep_hist": 1.0, "k_lut": 1.0, "k_removestars": 1.0}),
    "prep+erode": (7.0, {"k_prep_erode": 1.0}),
    # both passes' front ends in one sweep over the float frames: 5N for every frame (bright image) + 7N for the frames the dim
    # pass then works on (priced in main(): the split depends on how many frames the bright pass accepts)
    "prep(bright)+prep+erode(dim)": (None, {"k_prep_dual": 1.0}),
    # (k_bits_erode: the dim pass of lfdmi_detect_batch erodes from the planes the bright pass's sweep left -- the dim
    # conversion, 5N by this table, is part of k_prep_hist's one sweep there and is not charged a second time)
    "erode": (2.0, {"k_morph(erode)": 1.0, "k_bits_erode": 1.0}),
    "dilate": (2.0, {"k_morph(dilate)": 1.0, "k_dilate_canny": 0.55}),
    # Canny = NMS (the Sobel / NMS stages are ~45 % of the fused tile kernel: stage ablation in profiles/README.md)
    # + hysteresis (candidate-run scan, per-frame union-find, general fallback kernels)
    "canny": (2.0, {"k_dilate_canny": 0.45, "k_canny_nms": 1.0, "k_runs_init(fg)": 1.0, "k_frame_fg": 1.0, "k_runs_merge8": 1.0,
                    "k_runs_flatten(fg)": 1.0, "k_edge_from_cand": 1.0}),
    "contours+rect+fill": (2.0, {"k_runs_init(bg)": 1.0, "k_frame_bg": 1.0, "k_frame_keys": 1.0, "k_runs_merge4_bg": 1.0,
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=("sdss", "lsst"), default="sdss")
    ap.add_argument("--frames-per-gpu", type=int, default=None, help="default 256 (configs[2]: 256 per GPU; configs[4]: 2048 over 8 GPUs)")
    ap.add_argument("--inflight", type=int, default=None, help="frames per launch (default: the whole batch)")
    ap.add_argument("--lanes", type=int, default=1, help="concurrent half-batches (streams) per GPU")
    ap.add_argument("--cpu-sample", type=int, default=None, help="frames timed through the CPU oracle on one thread (0 = skip)")
    ap.add_argument("--gen-workers", type=int, default=-1)
    ap.add_argument("--no-removestars", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="developer: leave the per-launch HIP events off (no roofline entry) to see what they cost")
    ap.add_argument("--host-frames", action="store_true",
                    help="developer: time ONLY the host-resident (PCIe-inclusive) path as the headline value of this run")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the secondary PCIe-inclusive measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        args.gpus = world

    from lfd_amd import synth
    lsst = args.workload == "lsst"
    shape = synth.LSST_SHAPE if lsst else synth.SDSS_SHAPE
    n = args.frames_per_gpu or 256
    inflight = args.inflight or n
    cpu_sample = args.cpu_sample if args.cpu_sample is not None else (2 if lsst else 40)
    k0 = rank * n
    workers = args.gen_workers
    if workers < 0:
        workers = max(1, min(16, (os.cpu_count() or 8) // max(1, world)))
    t0 = time.time()
    # child processes (never forks of this one: safe under a profiler's preloaded library), frames into shared memory
    host, cats = synth.make_frames(k0, n, shape, workers, with_catalog=not lsst)
    t_gen = time.time() - t0

    import torch
    import torch.distributed as dist
    from lfd_amd import _native
    from lfd_amd.batch import BatchDetector
    from lfd_amd.detecttrails import default_params

    # One rank per GPU.  Rehearsal on a box with fewer GPUs than ranks (LFD_BENCH_SHARE_GPU=1): ranks share devices round-robin
    # and synchronise over gloo (RCCL refuses two ranks on one device) -- the sharding, timing and reduction code is the same.
    share = os.environ.get("LFD_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % max(1, torch.cuda.device_count()) if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or os.environ.get("LFD_BENCH_FORCE_DIST") == "1"  # the env switch lets a 1-GPU box rehearse the RCCL path
    red_dev = torch.device("cpu") if share else dev
    if use_dist:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    pb, pd, prs = default_params()
    rs = _native.make_rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})
    h, w = shape
    rhos = [20.0, 10.0, 5.0]
    if lsst:
        pd = dict(pd, erodeKernel=np.ones((9, 9), np.uint8))
    dframes = torch.from_numpy(host).to(dev)
    cat = packed = None
    if not lsst and not args.no_removestars:
        packed = synth.pack_catalogs(cats)
        cat = {k: torch.from_numpy(v).to(dev) for k, v in packed.items()}
    stream = torch.cuda.current_stream().cuda_stream
    det = BatchDetector(dev_index, (h, w), inflight, stream=stream, lanes=args.lanes)

    if lsst:
        def run(frames, _cat):
            return det.multiscale(frames, pd, rhos, dim=True, flip=True)
    else:
        def run(frames, c):
            return det.detect(frames, pb, pd, c, rs)

    def step_dev():
        return run(dframes, cat)

    def step_host():
        return run(host, None if (lsst or args.no_removestars) else packed)

    step = step_host if args.host_frames else step_dev

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # Per-launch HIP events cost ~6 % of a step when every one of its ~40 launches is bracketed, so inside
    # the timed region only the few largest kernels are (those that move pixels; the warm-up steps, fully
    # bracketed, nominate four -- a cold first launch can distort a single winner); the dominant one is then
    # chosen from the timed region's own sums.  The per-kernel table of the JSON line comes from one extra,
    # fully bracketed step AFTER the timed region.
    res = None
    det.enable_timing(not args.no_kernel_timing)
    for _ in range(args.warmup):
        res = step()
    torch.cuda.synchronize()
    warm = det.get_timing()
    cands = sorted((k for k, v in warm.items() if v[1] and KERNEL_BYTES_PER_PX.get(k, 0.0) != 0.0), key=lambda k: -warm[k][0])[:4]
    det.timing_select(cands or ["k_dilate_canny", "k_prep_hist", "k_hough_vote", "k_prep_erode"])
    det.enable_timing(not args.no_kernel_timing)  # (re-arms and clears the sums)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    timing = det.get_timing()
    det.timing_select(None)
    det.enable_timing(not args.no_kernel_timing)
    step()                                        # untimed: the per-kernel table
    torch.cuda.synchronize()
    table = det.get_timing()
    det.enable_timing(False)
    host_leg = None
    if not args.host_frames and not args.no_host_leg:  # secondary: frames handed over as host buffers (PCIe-inclusive; never `value`)
        step_host()
        fence()
        t0 = time.perf_counter()
        m = max(1, min(args.steps, 3))
        for _ in range(m):
            step_host()
        fence()
        host_leg = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([host_leg], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            host_leg = float(t.item())
        host_leg = (world * n * m / host_leg, m)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    cnt = det.get_counters()[:n]  # last pass of every slot: Hough cost is 180 votes per non-zero pixel
    nnz_equ, nnz_box = cnt[:, 15][cnt[:, 8] > 0], cnt[:, 16][cnt[:, 8] > 0]
    res0 = res[0] if lsst else res       # lsst: [scale, frame]
    found_b = int((res0["found"] == 1).sum())
    found_d = int((res0["found"] == 2).sum())
    errors = int((res["status"] != 0).sum())

    if rank == 0:
        total_frames = world * n * args.steps
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
        traffic, traffic_src = load_traffic(name, (args.workload, n, inflight, args.lanes, [h, w]))
        kern = {k: {"ms_per_step": round(v[0], 4), "launches_per_step": v[1], "frames_per_step": v[2]}
                for k, v in table.items() if v[1]}  # one fully bracketed step after the timed region
        # per-stage view of that step: every stage's algorithmic bytes once, over the summed time of its kernels
        stages = {}
        for sname, (bpp, ks) in STAGES.items():
            t_ms = sum(table[k][0] * share for k, share in ks.items() if k in table and table[k][1])
            if t_ms <= 0:
                continue
            lead = max((k for k in ks if k in table and table[k][1]), key=lambda k: table[k][0] * ks[k])
            frames_st = table[lead][2]  # (Hough: frames with a rectangle, summed over the launches of every scale)
            gb = (dual_bpp if bpp is None else bpp) * N * frames_st / 1e9
            stages[sname] = {"ms_per_step": round(t_ms, 4), "algorithmic_GB": round(gb, 4), "GBps": round(gb / (t_ms * 1e-3), 1),
                             "frac_of_peak": round(gb / (t_ms * 1e-3) / HBM_PEAK_GBPS, 4)}
        ch = None
        if "canny" in stages and "hough" in stages:  # what north_star asks for: Canny (2N) + Hough (1N per image) over their kernels
            gb = stages["canny"]["algorithmic_GB"] + stages["hough"]["algorithmic_GB"]
            t_ms = stages["canny"]["ms_per_step"] + stages["hough"]["ms_per_step"]
            t_lo = t_ms + sum(table[k][0] * 0.55 for k in ("k_dilate_canny",) if k in table)  # the whole fused tile kernel charged to Canny
            ch = {"algorithmic_GB": round(gb, 4), "ms_per_step": round(t_ms, 4), "GBps": round(gb / (t_ms * 1e-3), 1),
                  "frac": round(gb / (t_ms * 1e-3) / HBM_PEAK_GBPS, 4),
                  "frac_if_fused_tile_kernel_is_all_canny": round(gb / (t_lo * 1e-3) / HBM_PEAK_GBPS, 4),
                  "target": 0.5}
        if lsst:
            metric = "LSST-scale frames/sec (4096x4096) dim pass, 9x9 erosion, multi-scale Hough"
            workload = ("configs[4]: dim pass with 9x9 erosion + HoughLines at rho 20/10/5, batch=%d synthetic 4096x4096 float32 frames "
                        "per GPU, %s" % (n, "HOST-resident (PCIe-inclusive)" if args.host_frames else "device-resident"))
        else:
            metric = "SDSS frames/sec (2048x1489) full detecttrails pipe"
            workload = ("configs[2]: full removestars+bright+dim pipe, batch=%d synthetic SDSS 2048x1489 float32 frames per GPU, %s"
                        % (n, "HOST-resident (PCIe-inclusive)" if args.host_frames else "device-resident"))
        out = {
            "metric": metric, "value": round(value, 2),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "frames_per_gpu": n, "inflight": inflight, "lanes": args.lanes, "shape": [h, w],
                       "removestars": (not lsst) and not args.no_removestars, "parallelism": "frame-parallel x%d" % world + (" (rehearsal: ranks share %d GPU(s), gloo)" % torch.cuda.device_count() if share else ""),
                       "found_bright": found_b, "found_dim": found_d, "frame_errors": errors,
                       "hough_rhos": rhos if lsst else [20.0],
                       "hough_nnz_equ_median": int(np.median(nnz_equ)) if len(nnz_equ) else 0,
                       "hough_nnz_box_median": int(np.median(nnz_box)) if len(nnz_box) else 0,
                       "workspace_GB": round(det.workspace_bytes() / 1e9, 2), "frames_spilled_to_worst_case_workspace": det.spill_count(),
                       "gen_s": round(t_gen, 1)},
            "roofline": {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "measured_traffic_frac": round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5) if traffic and avg_ms > 0 else None,
                         "frac_of_measured_copy_6290": round(achieved / 6290.0, 5),
                         "avg_launch_ms": round(avg_ms, 4),
                         "frames_per_launch": round(units / max(1, launches), 2),
                         "algorithmic_bytes_per_frame": bytes_per_frame,
                         "algorithmic_bytes_per_px": round(bpp_dom, 3),
                         "canny_hough": ch,
                         "note": "algorithmic bytes as SURVEY 8(d) prescribes, each stage charged once across its kernels; "
                                 "measured_traffic_frac = PMC HBM bytes per launch / launch time / peak: a kernel whose measured "
                                 "fraction is far below 1 is limited by vector-unit issue / LDS round trips, not by HBM"},
            "stages": stages,
            "kernels": kern,
        }
        if host_leg:
            out["host_resident"] = {"value": round(host_leg[0], 2), "unit": "frames/s", "steps": host_leg[1],
                                    "note": "same step with the frames handed over as host buffers (PCIe-inclusive); never `value`"}
        if world > 1:
            out["cpu_baseline"] = None  # (the host baseline is timed at N = 1 only: rank 0 would keep the other ranks waiting)
        elif cpu_sample > 0:
            from concurrent.futures import ThreadPoolExecutor
            from oracle import lfd_oracle as O
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
            try:
                avail = len(os.sched_getaffinity(0))
            except (AttributeError, OSError):
                avail = os.cpu_count() or 1
            cores = max(1, min(avail, 16 * max(1, world), n))     # this process's share of the host (16 cores per GPU on the pool)
            what = "3 x process_dim (rho 20/10/5)" if lsst else "detect_frame"
            out["cpu_baseline"] = {"value": round(m / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": "first %d frames of rank 0's batch through oracle/ (C, -O2, 1 thread; %s), "
                                             "%d/%d identical to the GPU records" % (m, what, agree, m),
                                   "host_cores_available": avail}
            # all host cores: one oracle call per thread (ctypes releases the GIL), a bounded sample again
            per = max(1, int(round(m / dt * 12.0)))               # ~12 s of work per core
            ma = min(n, cores * per)
            t0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                agree_a = sum(ex.map(cpu_frame, range(ma)))
            dta = time.perf_counter() - t0
            out["cpu_baseline"]["all_cores"] = {"value": round(ma / dta, 3), "unit": "frames/s", "cores": cores,
                                                "sample": "%d frames on %d threads, %d/%d identical to the GPU records" % (ma, cores, agree_a, ma)}
        print(json.dumps(out), flush=True)
    det.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
