"""lfd_amd: the lfd.detecttrails per-frame hot path on MI355X (gfx950).

``lfd_amd.detecttrails`` mirrors the reference package's interface; ``lfd_amd._native`` is the
ctypes binding of the C-ABI library (lfd_amd/csrc/liblfdmi.so, include/lfdmi.h);
``lfd_amd.batch`` shards frame batches over the GPUs of a node; ``lfd_amd.synth`` generates the
deterministic synthetic workload of BASELINE.json.
"""
__version__ = "0.1.0"


def usable_cores():
    """CPUs this process can really keep busy: its affinity mask, cut down to the CPU quota of its cgroup (a container that sees
    256 CPUs but may use 16 cores' worth of time is throttled as a whole when 64 threads run: round 4's first .bz2 leg was
    slower with more threads).  cgroup v2 ``cpu.max`` / v1 ``cpu.cfs_quota_us``; no quota: the mask."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n
