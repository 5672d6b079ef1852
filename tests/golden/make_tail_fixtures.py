"""Generate golden vectors for the numpy-only tail of the reference hot path.

Runs ONLY in the build container (needs /root/reference).  Loads the reference's
``lfd/detecttrails/processfield.py`` by file path with an empty placeholder module
registered as ``cv2`` (cv2 is not installed; the two functions exercised here,
``check_theta`` (processfield.py:36-150) and ``dictify_hough`` (processfield.py:266-288),
never touch it) and records inputs and outputs as JSON.  Only data is written:
no reference source text is stored.

    python tests/golden/make_tail_fixtures.py
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/lfd/detecttrails/processfield.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tail_fixtures.json")


def load_ref():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    spec = importlib.util.spec_from_file_location("_ref_processfield", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_ref()
    rng = np.random.default_rng(12345)
    f32 = np.float32
    theta_step = f32(np.pi / 180)

    dictify = []
    shapes = [(1489, 2048), (4096, 4096), (512, 768)]
    fixed = [((1489, 2048), 10.0, 0.0), ((1489, 2048), 1010.0, 0.7853981852531433),
             ((1489, 2048), -350.0, 2.356194496154785), ((1489, 2048), 1490.0, 1.5707963705062866),
             ((4096, 4096), 2890.0, 0.767944872379303), ((1489, 2048), -2030.0, 3.1241393089294434)]
    cases = list(fixed)
    for _ in range(200):
        shape = shapes[int(rng.integers(0, len(shapes)))]
        n = int(rng.integers(0, 180))
        theta = float(f32(n) * theta_step)
        r = int(rng.integers(-170, 170))
        rho = float(f32(20 * r + 10))
        cases.append((shape, rho, theta))
    for shape, rho, theta in cases:
        out = ref.dictify_hough(shape, (f32(rho), f32(theta)))
        dictify.append({"shape": list(shape), "rho": float(f32(rho)), "theta": float(f32(theta)),
                        "out": {k: int(v) for k, v in out.items()}})

    def hl(pairs):
        return np.asarray(pairs, dtype=np.float32).reshape(-1, 1, 2)

    check = []

    def add(h1, h2, navg=3, dro=25, thetaTresh=0.15, lineSetTresh=0.15):
        a, b = hl(h1), hl(h2)
        res = ref.check_theta(a, b, navg, dro, thetaTresh, lineSetTresh, False)
        check.append({"h1": a.reshape(-1, 2).tolist(), "h2": b.reshape(-1, 2).tolist(),
                      "navg": navg, "dro": dro, "thetaTresh": thetaTresh,
                      "lineSetTresh": lineSetTresh,
                      "out": None if res is None else bool(res)})

    add([(1010, .7853982), (990, .7853982), (1030, .80285144)],
        [(1010, .7853982), (990, .7679449), (1030, .7853982)])
    add([(1010, .78), (990, .78), (1030, .78)], [(1100, .78), (1080, .78), (1120, .78)])
    add([(1010, .60), (990, .78), (1030, .78)], [(1010, .78), (990, .78), (1030, .78)])
    add([(1010, .78), (990, .78), (1030, .78)], [(1010, .60), (990, .78), (1030, .78)])
    add([(1010, .60)] * 3, [(1010, .78)] * 3)
    add([(10, .1)] * 3, [(10, .1)] * 2)
    add([(30, .1)], [(30, .1)])
    add([(10, 0), (-10, 3.1241393), (10, 0)], [(10, 0)] * 3)
    add([(1010, .78)] * 5, [(1010, .78)] * 4, navg=5, dro=20)
    for _ in range(300):
        n1 = int(rng.integers(1, 6))
        n2 = int(rng.integers(1, 6))
        base_n = int(rng.integers(0, 180))
        base_r = int(rng.integers(-100, 100))

        def mk(n):
            out = []
            for _i in range(n):
                dn = int(rng.integers(-6, 7)) if rng.random() < 0.5 else 0
                dr = int(rng.integers(-3, 4))
                out.append((float(f32(20 * (base_r + dr) + 10)),
                            float(f32(max(0, min(179, base_n + dn))) * theta_step)))
            return out
        add(mk(n1), mk(n2), navg=int(rng.integers(1, 5)), dro=int(rng.choice([20, 25])))

    # error behaviour: None input raises TypeError (processfield.py:97, not an IndexError)
    try:
        ref.check_theta(None, hl([(1, .1)]), 3, 25, .15, .15, False)
        none_raises = None
    except Exception as e:  # noqa: BLE001
        none_raises = type(e).__name__

    with open(OUT, "w") as f:
        json.dump({"numpy": np.__version__, "dictify_hough": dictify, "check_theta": check,
                   "check_theta_none_raises": none_raises}, f, indent=0)
    print("wrote", OUT, len(dictify), len(check), none_raises)


if __name__ == "__main__":
    main()
