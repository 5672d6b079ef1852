#!/bin/bash
# Developer tool: registers, scratch, LDS and occupancy of every kernel of liblfdmi.so (hipcc -Rpass-analysis=kernel-resource-usage).
cd "$(dirname "$0")/../lfd_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -pthread -shared -Rpass-analysis=kernel-resource-usage lfdmi.hip -o /tmp/lfdmi_res.so 2> /tmp/lfdmi_res.txt
python3 - "$@" <<'PY'
import re, sys
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for b in open("/tmp/lfdmi_res.txt").read().split("Function Name: ")[1:]:
    name = b.split("\n")[0].split(" ")[0]
    if pat and pat not in name: continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [0, "?"])[1]
    print("%-72s VGPR %4s SGPR %4s scratch %4s occ %2s LDS %6s" % (name[:72], g("VGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
PY
