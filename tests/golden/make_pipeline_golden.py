"""Golden records for the full pipe (remove_stars -> flip -> bright -> dim) on portable
synthetic frames (lfd_amd.synth.make_portable_frame: integer RNG + IEEE arithmetic only, so the
GPU box regenerates the same inputs bit for bit).  Expected values come from the CPU oracle
(oracle/, parity with OpenCV unpinned -- see DESIGN.md); the GPU tests compare the HIP path
with these records AND with the oracle run live.

    python tests/golden/make_pipeline_golden.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from lfd_amd import synth  # noqa: E402
from lfd_amd.detecttrails import default_params  # noqa: E402
from oracle import lfd_oracle as O  # noqa: E402

CASES = [(k, (512, 768)) for k in range(12)] + [(100, (1489, 2048)), (101, (1489, 2048)), (102, (333, 517))]


def main():
    pb, pd, prs = default_params()
    rs = O.rs_params("r", **{k: v for k, v in prs.items() if k != "debug"})
    out = []
    for k, shape in CASES:
        img, cat, truth = synth.make_portable_frame(k, shape)
        sha = hashlib.sha256(img.tobytes()).hexdigest()
        rec = O.detect_frame(img.copy(), pb, pd, cat, rs)
        out.append({"k": k, "shape": list(shape), "image_sha256": sha, "truth": truth, "record": rec})
        print(k, shape, truth["streak"], rec["found"], rec["rho"], rec["theta"])
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pipeline_golden.json"), "w") as f:
        json.dump({"cases": out}, f, indent=1)


if __name__ == "__main__":
    main()
