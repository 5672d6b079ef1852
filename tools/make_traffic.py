"""Build profiles/r01_traffic.json from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of
`python3 bench.py --cpu-sample 0 --steps 2`.

    python tools/make_traffic.py gpurun_out/pmc_fetch/run_counter_collection.csv \
                                 gpurun_out/pmc_write/run_counter_collection.csv profiles/r01_traffic.json
"""
import collections
import csv
import json
import sys

ABOUT = (
    "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (with --kernel-trace only) of "
    "`python3 bench.py --cpu-sample 0 --steps 2` (256 frames/GPU, 256 in flight, 1 lane) on one MI355X, final "
    "round-1 code.  Counters are in KB per dispatch, averaged over each kernel's dispatches (bright- and dim-pass "
    "launches mixed).  Corrections per MI355X_MICROARCH.md: FETCH_SIZE tallies the 128-B requests of a wide "
    "(16 B/lane) coalesced row stream as 64 B, so it is doubled for k_prep_hist, the only kernel that streams "
    "whole rows that way (fetch_factor 2; check: 2 x FETCH ~ the float32 input bytes).  The tile kernels fetch "
    "96-B row pieces; their raw FETCH_SIZE already matches tile+halo bytes, so they and all other kernels keep "
    "fetch_factor 1 (uncalibrated widths left as reported).  WRITE_SIZE is exact for 16 B/lane streaming stores; "
    "narrow stores (one u64 per lane in the bit-row writers) are counted per request and over-report.  "
    "hbm_bytes_per_launch = (fetch_factor*FETCH + WRITE) * 1024.")
FETCH_FACTOR = {"k_prep_hist": 2}


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
    out = {"_about": ABOUT, "config": {"frames_per_gpu": 256, "inflight": 256, "lanes": 1, "shape": [1489, 2048]}, "kernels": {}}
    for k in sorted(fetch):
        if k.startswith("__amd") or k not in write:
            continue
        f = sum(fetch[k]) / len(fetch[k])
        w = sum(write[k]) / len(write[k])
        ff = FETCH_FACTOR.get(k, 1)
        out["kernels"][k] = {"launches": len(fetch[k]), "FETCH_SIZE_KB_per_launch": round(f, 1), "WRITE_SIZE_KB_per_launch": round(w, 1),
                             "fetch_factor": ff, "hbm_bytes_per_launch": int((ff * f + w) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print("%-28s %8.1f MB/launch" % (k, v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
