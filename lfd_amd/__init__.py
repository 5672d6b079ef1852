"""lfd_amd: the lfd.detecttrails per-frame hot path on MI355X (gfx950).

``lfd_amd.detecttrails`` mirrors the reference package's interface; ``lfd_amd._native`` is the
ctypes binding of the C-ABI library (lfd_amd/csrc/liblfdmi.so, include/lfdmi.h);
``lfd_amd.batch`` shards frame batches over the GPUs of a node; ``lfd_amd.synth`` generates the
deterministic synthetic workload of BASELINE.json.
"""
__version__ = "0.1.0"
