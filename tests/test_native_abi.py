"""The C-ABI library loads on a machine without a GPU and exports every symbol the header
declares; struct layouts agree between the header's definitions and the ctypes mirror."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    with open(os.path.join(ROOT, "include", "lfdmi.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lfdmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from lfd_amd import _native
    lib = _native.lib()
    declared = header_symbols()
    assert len(declared) >= 20
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(_native.SYMBOLS) == declared
    assert lib.lfdmi_version() == 300
    # nothing is exported that the header does not declare
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("lfdmi_")})
    assert exported == declared, sorted(set(exported) ^ set(declared))


def test_struct_layouts():
    from lfd_amd import _native
    assert C.sizeof(_native.Result) == 48 == _native.RESULT_DTYPE.itemsize
    assert [n for n, _ in _native.Result._fields_] == list(_native.RESULT_DTYPE.names)
    assert C.sizeof(_native.Params) == 6 * 8 + 3 * 4 + 2 * 4 + 8 + 2 * 4 + 8 + 2 * 8 + 4 + 4 + 4 + 8  # with padding
    assert C.sizeof(_native.RsParams) == 48 and C.sizeof(_native.Catalog) == 72
    assert C.sizeof(_native.Caps) == 32
    c = _native.Caps()
    _native.lib().lfdmi_default_caps(1489, 2048, C.byref(c))
    n = 1489 * 2048
    assert (c.run_cap, c.key_cap, c.slot_cap, c.list_cap, c.peak_cap, c.min_rho) == (n // 16, n // 256, n // 16, n // 16, 65536, 5.0)


def test_hough_dims_without_gpu():
    from lfd_amd import _native
    na, nr = C.c_int(), C.c_int()
    _native.lib().lfdmi_hough_dims(1489, 2048, C.c_double(20), C.c_double(np.pi / 180), C.byref(na), C.byref(nr))
    assert (na.value, nr.value) == (180, 354)


def test_no_fallback_without_device():
    """On a box without a GPU the product path must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lfd_amd import _native
    with pytest.raises(_native.NativeError):
        _native.Context(0, 64, 64, 1)
    from lfd_amd.detecttrails import process_field_bright, default_params
    pb, _, _ = default_params()
    with pytest.raises(_native.NativeError):
        process_field_bright(np.zeros((32, 32), np.float32), **pb)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lfd_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                with open(os.path.join(dirpath, f)) as fh:
                    text = fh.read()
                assert "lfd_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
