"""Developer tool (GPU box): the device bzip2 decoder on synthetic data -- correctness against Python's bz2 and files/s.
usage: tools/bz2_probe.py [n_frames = 32] [level = 9]"""
import bz2, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lfd_amd import _native as Nv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
level = int(sys.argv[2]) if len(sys.argv) > 2 else 9
kind = sys.argv[3] if len(sys.argv) > 3 else "noise"       # noise: float32 Gaussian (barely compressible); counts: integer counts x a calibration vector (2 : 1)
rng = np.random.default_rng(5)
small = {
    "text": b"hello hello hello world" * 3,
    "runs": b"a" * 1000 + b"b" * 5 + bytes(300) + b"xyz" * 7 + b"\xfb" * 2000 + b"q" * 4 + b"r" * 259 + b"ssss",
    "allbytes": bytes(range(256)) * 50,
    "noise": rng.integers(0, 256, 300000, dtype=np.uint8).tobytes(),
    "one": b"z",
    "zeros": bytes(2_500_000),
    "fours": b"aaaa" + b"bbbb" * 3 + b"cccc",
    "periodic": b"abcd" * 5000,
}
hdr = b"".join(c.ljust(80) for c in (b"SIMPLE  =                    T", b"BITPIX  =                  -32", b"NAXIS   =                    2",
                                    b"NAXIS1  =                 2048", b"NAXIS2  =                 1489", b"END")).ljust(2880)
def frame(k):
    r = np.random.default_rng(k)
    if kind == "counts":
        counts = np.rint(r.normal(1100, 6, (1489, 2048))).astype(np.float32)
        calib = (0.005 * (1 + 0.02 * np.sin(np.arange(2048) / 300))).astype(np.float32)
        img = (counts * calib[None, :]).astype(">f4")
    else:
        img = r.normal(0.0, 0.025, (1489, 2048)).astype(">f4")
    img[200:300, 500:800] = 0.0
    return hdr + img.tobytes()
plains = list(small.values()) + [frame(k) for k in range(min(n, 4))]
while len(plains) < len(small) + n:
    plains.append(plains[len(small) + (len(plains) - len(small)) % min(n, 4)])
t0 = time.time()
comp = [bz2.compress(p, level) for p in plains[:len(small) + min(n, 4)]]
while len(comp) < len(plains):
    comp.append(comp[len(small) + (len(comp) - len(small)) % min(n, 4)])
print("compressed %d inputs in %.1f s; frame %.2f MB -> %.2f MB" % (len(comp), time.time() - t0, len(plains[-1]) / 1e6, len(comp[-1]) / 1e6), flush=True)
off, cur = [], 0
for c in comp:
    off.append(cur)
    cur += (len(c) + 7) & ~7
src = np.zeros(cur, np.uint8)
for o, c in zip(off, comp):
    src[o:o + len(c)] = np.frombuffer(c, np.uint8)
ln = [len(c) for c in comp]
cap = 13 * 1024 * 1024
with Nv.Bz2Decoder(0) as z:
    for rep in range(3):
        t = time.perf_counter()
        out_len, status, heads = z.decode(src, off, ln, cap, 2880)
        dt = time.perf_counter() - t
        print("decode %d files (%d frames): %.1f ms -> %.0f frames/s | %s" % (len(comp), n, 1e3 * dt, n / dt, {k: round(v, 1) for k, v in z.timings().items()}), flush=True)
    bad = 0
    for i, p in enumerate(plains):
        if i >= len(small) + 4 and i % 7:
            continue
        if status[i] != 0:
            print("file %d: status %d (%s), len %d" % (i, status[i], Nv.BZ2_STATUS.get(int(status[i])), len(p)))
            bad += 1
            continue
        got = z.fetch(i, 0, int(out_len[i])).tobytes()
        if got != p or (len(p) >= 2880 and heads[i].tobytes() != p[:2880]):
            k = next((j for j in range(min(len(got), len(p))) if got[j] != p[j]), -1)
            print("file %d: MISMATCH len %d vs %d first diff %d" % (i, len(got), len(p), k))
            bad += 1
    print("checked: %d bad" % bad)
    # declined inputs
    two = bz2.compress(b"abc") + bz2.compress(b"def")
    broken = bytearray(comp[3]); broken[len(broken) // 2] ^= 0x10
    srcs = [two, bytes(broken), b"not bzip2 at all", comp[0] + b"\0"]
    o2, c2 = [], 0
    for c in srcs:
        o2.append(c2); c2 += (len(c) + 7) & ~7
    s2 = np.zeros(c2, np.uint8)
    for o, c in zip(o2, srcs):
        s2[o:o + len(c)] = np.frombuffer(c, np.uint8)
    ol, st, _ = z.decode(s2, o2, [len(c) for c in srcs], 1 << 20)
    print("declined:", [(int(s), Nv.BZ2_STATUS.get(int(s))) for s in st])
