"""The stress workloads of lfd_amd.synth.STRESS: what the benchmark's sky does not exercise -- crowded
fields (4 000 and 20 000 stars: more candidate runs than the per-frame kernels' LDS table holds, catalogues that take the sorted
remove_stars path), noisier sky (half of it survives the dim pass's minFlux at sigma 0.1), a saturated star with a full-height
bleed column, one crowded frame in an otherwise quiet chunk.  Every record equals the CPU oracle's (detecttrails.py:119-131:
every frame must come out, whatever it holds), nothing is sent to the worst-case workspace, and what the context had to do
besides the fast path is visible in its stats.  bench.py's `stress` leg times the same workloads."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def same(rec, want):
    return all(rec[k].item() == v for k, v in want.items())


@pytest.mark.parametrize("name", ["stars4000", "stars20000", "sky0.05", "sky0.1", "bleed", "one_crowded"])
def test_stress_workload_matches_the_oracle(oracle, name):
    from lfd_amd import _native, synth
    from lfd_amd.detecttrails import default_params
    pb, pd, prs = default_params()
    kw = {k: v for k, v in prs.items() if k != "debug"}
    rs_g, rs_o = _native.make_rs_params("r", **kw), oracle.rs_params("r", **kw)
    n = 8 if name == "one_crowded" else 4
    recipes = synth.stress_recipes(name, 16)[:n] if name != "one_crowded" else synth.stress_recipes(name, 16)[2:10]
    k0 = 2 if name == "one_crowded" else 0
    frames, cats = zip(*[synth.make_frame(k0 + i, **recipes[i])[:2] for i in range(n)])
    batch = np.stack(frames)
    with _native.Context(0, 1489, 2048, n) as ctx:
        res = ctx.detect_batch(batch.copy(), pb, pd, synth.pack_catalogs(list(cats)), rs_g)
        st = ctx.stats()
        assert st["spilled_frames"] == 0, st
        runs = ctx.get_counters(0, n)[:, 12:14].max()
        if name in ("stars4000", "stars20000", "one_crowded"):
            assert st["general_chunks"] >= 1 and st["general_reruns"] == 1, st      # beyond the per-frame kernels' LDS table ...
        res2 = ctx.detect_batch(batch.copy(), pb, pd, synth.pack_catalogs(list(cats)), rs_g)
        assert res2.tobytes() == res.tobytes()
        assert ctx.stats()["general_reruns"] == st["general_reruns"]                # ... and from then on without a rerun
    check = range(n) if name != "stars20000" else (0, 3)
    for i in check:
        want = oracle.detect_frame(frames[i].copy(), pb, pd, cats[i], rs_o)
        assert same(res[i], want), (name, i, want, res[i])
