"""Per-kernel utilisation figures from rocprofv3 counter passes (tools/collect_profiles.sh) -> profiles/<tag>_util_<workload>.json,
read by bench.py (`kernels{}.util`, `roofline.kernel_rocprof`).

usage: make_util.py <out.json> <workload> <kernel_stats.csv> <traffic.json> <counter_collection.csv> [...]

Units and formulas (MI355X_MICROARCH.md: 256 CUs, 4 SIMD-32 per CU, SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles,
GRBM_GUI_ACTIVE is summed over the 8 XCDs), all over the profiled run's launches of a kernel:
  clocks          = GRBM_GUI_ACTIVE / 8                       shader clocks the kernel's launches lasted
  valu_pipe       = SQ_INSTS_VALU * 2 / (1024 * clocks)       a wave64 vector instruction holds a SIMD-32's pipe for 2 clocks
  valu_wave_active= SQ_ACTIVE_INST_VALU * 4 / (1024 * clocks) wave-clocks inside vector instructions per SIMD-clock (one wave alone
                                                              issues one per 4 clocks, so this can be up to 2 x valu_pipe)
  salu            = SQ_INSTS_SALU / (256 * clocks)            one scalar instruction per CU and clock
  lds             = SQ_INSTS_LDS * 4 / (256 * clocks)         a 64-lane 32-bit LDS instruction: 4 LDS clocks
  lds_conflict    = SQ_LDS_BANK_CONFLICT / (256 * clocks)     extra LDS clocks lost to bank conflicts
  waves_per_simd  = SQ_WAVE_CYCLES * 4 / (1024 * clocks)      average occupancy (8 = full)
  parked          = SQ_WAIT_ANY / SQ_WAVE_CYCLES              share of a wave's life in s_waitcnt / barriers
  issue_stalled   = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES         ... waiting to issue (dependencies, busy pipes)
  hbm             = (FETCH_SIZE * factor + WRITE_SIZE bytes per launch, from the traffic file) / launch time / 8 TB/s
"""
import collections
import csv
import hashlib
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SLOT = {  # rocprof kernel (name before its template / argument list) -> timing slot of the library (bench.py `kernels{}`)
    "k_prep_hist": "k_prep_hist", "k_prep_erode": "k_prep_erode", "k_bits_erode": "k_bits_erode", "k_lut": "k_lut",
    "k_dilate_canny_t": "k_dilate_canny", "k_dc_tiles": "k_dilate_canny", "k_tile_perm": "k_dilate_canny",
    "k_scan_fused": "k_scan_fused", "k_frame_fg": "k_frame_fg", "k_frame_contours": "k_frame_bg",
    "k_rects": "k_rects", "k_rects_big": "k_rects", "k_fill_quads": "k_fill_quads", "k_pixlist": "k_pixlist",
    "k_hough_vote": "k_hough_vote", "k_hough_peaks": "k_hough_peaks", "k_hough_topk": "k_hough_topk", "k_finalize": "k_finalize",
    "k_rs_boxes": "k_removestars", "k_rs_fill": "k_removestars", "k_morph_rect_rows": "k_morph(erode)", "k_erode_cand": "k_morph(erode)",
    "k_fill_around": "k_morph(erode)", "k_hough_reset": "k_hough_vote",
}


def base(name):
    n = name.strip().strip('"')
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0].split("(")[0].strip()


def csrc_sha():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "lfd_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "lfd_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "lfd_amd", "csrc", "*.inc"))):
        if os.path.basename(f) in ("k_bz2.h", "bz2_core.h", "bz2dev.hip"):   # the bzip2 decoder: its own translation unit, none of the profiled kernels
            continue
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def figures(c):
    g = c.get
    clocks = g("GRBM_GUI_ACTIVE", 0.0) / 8.0
    out = {}
    if clocks <= 0:
        return out
    r = lambda v: float("%.3g" % v)
    if "SQ_INSTS_VALU" in c: out["valu_pipe"] = r(c["SQ_INSTS_VALU"] * 2 / (1024 * clocks))
    if "SQ_ACTIVE_INST_VALU" in c: out["valu_wave_active"] = r(c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * clocks))
    if "SQ_INSTS_SALU" in c: out["salu"] = r(c["SQ_INSTS_SALU"] / (256 * clocks))
    if "SQ_INSTS_LDS" in c: out["lds"] = r(c["SQ_INSTS_LDS"] * 4 / (256 * clocks))
    if "SQ_LDS_BANK_CONFLICT" in c: out["lds_conflict"] = r(c["SQ_LDS_BANK_CONFLICT"] / (256 * clocks))
    if "SQ_WAVE_CYCLES" in c:
        out["waves_per_simd"] = r(c["SQ_WAVE_CYCLES"] * 4 / (1024 * clocks))
        if "SQ_WAIT_ANY" in c: out["parked"] = r(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"])
        if "SQ_WAIT_INST_ANY" in c: out["issue_stalled"] = r(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"])
    return out


def main():
    out_path, workload, stats_csv, traffic_json = sys.argv[1:5]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in sys.argv[5:]:
        for row in csv.DictReader(open(f)):
            acc[base(row["Kernel_Name"])][row["Counter_Name"]] += float(row["Counter_Value"])
    stats, stats_full = {}, {}
    for row in csv.DictReader(open(stats_csv)):
        full = row["Name"].strip().strip('"').replace("void ", "").split("(")[0]
        stats_full[full] = (int(row["Calls"]), float(row["TotalDurationNs"]))
        s = stats.setdefault(base(row["Name"]), [0, 0.0])
        s[0] += int(row["Calls"]); s[1] += float(row["TotalDurationNs"])
    traffic = json.load(open(traffic_json)).get("kernels", {}) if os.path.exists(traffic_json) else {}
    hbm = collections.defaultdict(lambda: [0.0, 0.0])  # base kernel -> [bytes, ns] over the stats run's launches
    for full, v in traffic.items():
        if full in stats_full:
            calls, tns = stats_full[full]
            hbm[base(full)][0] += v["hbm_bytes_per_launch"] * calls
            hbm[base(full)][1] += tns
    kernels, slots = {}, collections.defaultdict(lambda: [collections.defaultdict(float), 0.0, 0.0])
    for b_, c in acc.items():
        calls, tns = stats.get(b_, [0, 0.0])
        kernels[b_] = figures(c)
        kernels[b_]["avg_launch_us"] = round(tns / calls / 1e3, 1) if calls else None
        if hbm[b_][1] > 0:
            kernels[b_]["hbm"] = float("%.3g" % (hbm[b_][0] / (hbm[b_][1] * 1e-9) / 8e12))
        s = SLOT.get(b_)
        if s:
            for k, v in c.items():
                slots[s][0][k] += v
            slots[s][1] += hbm[b_][0]
            slots[s][2] += hbm[b_][1]
    top = max(stats.items(), key=lambda kv: kv[1][1]) if stats else None
    doc = {"workload": workload, "collected_unix": int(time.time()), "csrc_sha256": csrc_sha(),
           "formulas": __doc__.split("Units and formulas")[1].strip(),
           "kernels": kernels, "slots": {}}
    for s, v in slots.items():
        doc["slots"][s] = figures(v[0])
        if v[2] > 0:
            doc["slots"][s]["hbm"] = float("%.3g" % (v[1] / (v[2] * 1e-9) / 8e12))
    if top:
        doc["top_kernel"] = {"name": top[0], "calls": top[1][0], "avg_launch_us": round(top[1][1] / top[1][0] / 1e3, 1),
                             "share_of_kernel_time": round(top[1][1] / sum(v[1] for v in stats.values()), 4), "util": kernels.get(top[0], {})}
    json.dump(doc, open(out_path, "w"), indent=1, sort_keys=True)
    for b in sorted(kernels, key=lambda b: -stats.get(b, [0, 0])[1])[:12]:
        print("%-20s %s" % (b, kernels[b]))


if __name__ == "__main__":
    main()
