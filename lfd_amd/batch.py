"""Frame-parallel sharding of a batch over the GPUs of one node.

Frames are independent (process_field keeps no cross-frame state, detecttrails.py:30-143) and
the reference itself scales by handing disjoint run lists to separate PBS jobs
(createjobs/createjobs.py:173-202).  Here every rank (one process per GPU) takes a contiguous
block of frames, runs ``lfdmi_detect_batch`` on its own device, and the per-frame result
records (48 B each) are gathered once at the end.  There is no collective on the data path.
"""
import threading
import time

import numpy as np

from . import _native


def shard_bounds(n_frames, world_size):
    """Contiguous blocks of ceil(n/world) frames: [(start, stop)] per rank (config 4: 1024 each)."""
    per = -(-n_frames // world_size) if world_size > 0 else n_frames
    return [(min(r * per, n_frames), min((r + 1) * per, n_frames)) for r in range(world_size)]


def shard_range(n_frames, rank, world_size):
    return shard_bounds(n_frames, world_size)[rank]


def gather_results(local, n_frames, group=None):
    """All ranks contribute their structured result array; every rank gets the full batch back.

    Uses torch.distributed (gloo on CPU tensors, RCCL on device tensors); the payload is
    48 B x frames, so this is a latency-bound bookkeeping exchange, not data-path traffic."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    bounds = shard_bounds(n_frames, world)
    per = max(b - a for a, b in bounds)
    rec = _native.RESULT_DTYPE.itemsize
    buf = np.zeros(per * rec, np.uint8)
    raw = np.ascontiguousarray(local).view(np.uint8).reshape(-1)
    buf[:raw.size] = raw
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.from_numpy(buf).to(dev)
    out = torch.empty(world * per * rec, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, mine, group=group)
    allrec = out.cpu().numpy().reshape(world, per * rec)
    parts = [allrec[r, :(b - a) * rec].copy().view(_native.RESULT_DTYPE) for r, (a, b) in enumerate(bounds)]
    return np.concatenate(parts)


class BatchDetector:
    """Runs the full pipe over this rank's share of a batch on one GPU.

    ``lanes`` independent contexts (own HIP stream, own workspace for ``inflight // lanes``
    frames) work on disjoint slices of the batch concurrently, each driven by its own host
    thread (ctypes releases the GIL).  Most kernels of this path are latency-bound with long
    tails (a few tall contours, the largest Hough lists), so two or more streams keep the CUs
    busy where one stream would idle between dependent launches.  Frames are independent, so
    this is the same frame-parallel sharding as across GPUs, one level down.
    """

    def __init__(self, device=0, shape=(1489, 2048), inflight=32, stream=None, lanes=1, caps=None, stage_images=False,
                 calls_in_flight=1):
        from concurrent.futures import ThreadPoolExecutor
        self.lanes = max(1, int(lanes))
        self.calls_in_flight = max(1, int(calls_in_flight))
        if self.calls_in_flight > 1 and self.lanes > 1:
            raise ValueError("calls_in_flight > 1 needs lanes == 1")
        per = max(1, inflight // self.lanes)
        self.ctxs = [_native.Context(device, shape[0], shape[1], per, caps=caps) for _ in range(self.lanes * self.calls_in_flight)]
        for c in self.ctxs:  # a batch detector is for throughput: no 8-bit debug images per frame unless asked for
            c.set_stage_images(1 if stage_images else 0)
        self.ctx = self.ctxs[0]
        self._own_stream = None
        if self.calls_in_flight > 1 and stream is None:
            import torch                                   # (a stream for the contexts to share when the caller names none)
            self._own_stream = torch.cuda.Stream(device)
            stream = self._own_stream.cuda_stream
        if self.lanes == 1 and stream is not None:
            # several calls in flight: every context launches into the SAME stream (one after the other on the GPU, each in its
            # own workspace)
            for c in self.ctxs:
                c.set_stream(stream)
        self.pool = ThreadPoolExecutor(self.lanes) if self.lanes > 1 else None
        # detect_async / multiscale_async: a host thread per context, calls dealt out in turn
        self._turn_pool = [ThreadPoolExecutor(1) for _ in self.ctxs] if self.calls_in_flight > 1 else None
        self._turn = 0
        self._pending = []
        self._start_lock = threading.Lock()
        self._last_start = 0.0
        self._spacing = 0.001
        self.shape = shape

    def close(self):
        if self.pool is not None:
            self.pool.shutdown()
        for ex in self._turn_pool or ():
            ex.shutdown()
        for c in self.ctxs:
            c.close()

    # ---- several calls in flight ---------------------------------------------------------------------------------------------
    # A synchronous call returns when its records are on the host; the next call's first kernel then starts ~0.1 ms later (the
    # return, the caller, the re-entry, a launch on an idle queue: profiles/r04_gap_trace.txt).  With ``calls_in_flight=2`` the
    # detector owns two contexts (a workspace each) that launch into one stream, and a host thread per context: while call k runs,
    # call k + 1 is already queued behind it, and the GPU goes from one to the other without waiting for the host.  The calls
    # still run one after the other on the GPU and their results are what the synchronous call returns.
    def _in_turn(self, method, args):
        if self._turn_pool is None:
            raise RuntimeError("BatchDetector(calls_in_flight=1): use detect / multiscale")
        i = self._turn
        self._turn = (i + 1) % len(self.ctxs)
        fut = self._turn_pool[i].submit(self._call, i, method, args)
        self._pending = [f for f in self._pending if not f.done()] + [fut]
        return fut

    def _busy(self):
        """Calls still in flight (a context serves one host thread at a time: a synchronous call then takes its turn too)."""
        self._pending = [f for f in self._pending if not f.done()]
        return bool(self._pending)

    def _call(self, i, method, args):
        # two calls that start together interleave their launches and finish together -- and would keep doing so: the starts are
        # kept a quarter of a call apart (in the steady state they are half a call apart by themselves)
        with self._start_lock:
            wait = self._last_start + self._spacing - time.perf_counter()
            if wait > 0:
                time.sleep(wait)
            self._last_start = t0 = time.perf_counter()
        out = getattr(self.ctxs[i], method)(*args)
        self._spacing = min(0.05, max(0.0005, 0.25 * (time.perf_counter() - t0)))
        return out

    def detect_async(self, frames, params_bright, params_dim, catalogs=None, rs=None):
        """``detect`` as a ``concurrent.futures.Future`` (``calls_in_flight`` > 1): submit the next batch before asking for the
        previous one's ``result()``.  Device-resident frames must stay untouched until the result is there."""
        return self._in_turn("detect_batch", (frames, params_bright, params_dim, catalogs, rs))

    def multiscale_async(self, frames, params, rhos, dim=True, flip=True, after_bright=False):
        return self._in_turn("process_multiscale", (frames, params, rhos, dim, flip, after_bright))

    def enable_timing(self, on=True):
        for c in self.ctxs:
            c.enable_timing(on)

    def timing_select(self, names=None):
        for c in self.ctxs:
            c.timing_select(names)

    def get_timing(self):
        out = {}
        for c in self.ctxs:
            for k, (ms, n, u) in c.get_timing().items():
                a = out.get(k, (0.0, 0, 0))
                out[k] = (a[0] + ms, a[1] + n, a[2] + u)
        return out

    def workspace_bytes(self):
        return sum(c.workspace_bytes() for c in self.ctxs)

    def spill_count(self):
        return sum(c.spill_count() for c in self.ctxs)

    def stats(self):
        """Sum of every lane's ``Context.stats()``."""
        out = {}
        for c in self.ctxs:
            for k, v in c.stats().items():
                out[k] = out.get(k, 0) + v
        return out

    def get_counters(self):
        """Work counters ([slots, 20] int32, see lfdmi_get_counters) the last pass left, all lanes."""
        return np.concatenate([c.get_counters() for c in self.ctxs])

    @staticmethod
    def _slice_cat(cat, a, b):
        if cat is None:
            return None
        return {k: v[a:b] for k, v in cat.items()}

    def multiscale(self, frames, params, rhos, dim=True, flip=True, after_bright=False):
        """One pass (dim or bright) with HoughLines at every rho of ``rhos`` over the batch: records [len(rhos), n]."""
        if self.lanes == 1:
            if self._turn_pool is not None and self._busy():
                return self.multiscale_async(frames, params, rhos, dim, flip, after_bright).result()
            return self.ctx.process_multiscale(frames, params, rhos, dim=dim, flip=flip, after_bright=after_bright)
        n = frames.shape[0]
        bounds = shard_bounds(n, self.lanes)
        futs = [self.pool.submit(c.process_multiscale, frames[a:b], params, rhos, dim, flip, after_bright)
                for c, (a, b) in zip(self.ctxs, bounds) if b > a]
        return np.concatenate([f.result() for f in futs], axis=1)

    def detect(self, frames, params_bright, params_dim, catalogs=None, rs=None):
        """frames: (n, h, w) float32, numpy (staged through the library) or a torch CUDA tensor
        (used in place; must be complete on the device before the call when lanes > 1).
        catalogs: dict from synth.pack_catalogs (numpy or torch CUDA tensors)."""
        if self.lanes == 1:
            if self._turn_pool is not None and self._busy():
                return self.detect_async(frames, params_bright, params_dim, catalogs, rs).result()
            return self.ctx.detect_batch(frames, params_bright, params_dim, catalogs, rs)
        n = frames.shape[0]
        bounds = shard_bounds(n, self.lanes)
        futs = [self.pool.submit(c.detect_batch, frames[a:b], params_bright, params_dim,
                                 self._slice_cat(catalogs, a, b), rs)
                for c, (a, b) in zip(self.ctxs, bounds) if b > a]
        return np.concatenate([f.result() for f in futs])
