"""Build profiles/rNN_traffic*.json from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of a bench.py command.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --cpu-sample 0 --steps 2 --no-host-leg
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --cpu-sample 0 --steps 2 --no-host-leg
    python tools/make_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [sdss|lsst] [frames_per_gpu]

(the two counters do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots; separate passes with --kernel-trace only.)
"""
import collections
import csv
import json
import sys

ABOUT = (
    "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (with --kernel-trace only) of "
    "`python3 bench.py --cpu-sample 0 --steps 2 --no-host-leg%s` on one MI355X.  Counters are in KB per dispatch, averaged over each "
    "kernel's dispatches (bright- and dim-pass launches mixed).  Corrections per MI355X_MICROARCH.md (HBM section): FETCH_SIZE "
    "tallies the 128-B requests of a wide (16 B/lane) coalesced row stream as 64 B, so it is doubled for the kernels that stream "
    "whole float rows that way: k_prep_hist and k_prep_erode (fetch_factor 2; check: 2 x FETCH ~ the float32 input bytes, halo rows "
    "of the band kernel included).  The tile kernels fetch 96-B row pieces; their raw FETCH_SIZE is left as reported (fetch_factor 1, "
    "uncalibrated width).  WRITE_SIZE is exact for 16 B/lane streaming stores; narrow stores (one u64 per lane in the bit-row "
    "writers) are counted per request and over-report.  hbm_bytes_per_launch = (fetch_factor*FETCH + WRITE) * 1024.")
FETCH_FACTOR = {"k_prep_hist": 2, "k_prep_erode": 2}


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[name].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    workload = sys.argv[4] if len(sys.argv) > 4 else "sdss"
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 256
    shape = [1489, 2048] if workload == "sdss" else [4096, 4096]
    out = {"_about": ABOUT % ("" if workload == "sdss" else " --workload lsst"),
           "config": {"workload": workload, "frames_per_gpu": n, "inflight": n, "lanes": 1, "shape": shape}, "kernels": {}}
    for k in sorted(fetch):
        if k.startswith("__amd") or k not in write:
            continue
        f = sum(fetch[k]) / len(fetch[k])
        w = sum(write[k]) / len(write[k])
        ff = FETCH_FACTOR.get(k.split("<")[0], 1)
        out["kernels"][k] = {"launches": len(fetch[k]), "FETCH_SIZE_KB_per_launch": round(f, 1), "WRITE_SIZE_KB_per_launch": round(w, 1),
                             "fetch_factor": ff, "hbm_bytes_per_launch": int((ff * f + w) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print("%-34s %8.1f MB/launch" % (k, v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
