"""The sequential core of the device bzip2 decoder (lfd_amd/csrc/bz2_core.h: block header, coding tables, Huffman + run-length +
move-to-front -> the block's BWT column) compiled for the CPU (tools/bz2_core_check.cpp, g++) and run on whole .bz2 files: every
block's CRC, the stream's CRC and the bytes against Python's bz2 module.  The device kernel shares this code for everything but the
symbol loop (k_bz2.h has its own, checked on the GPU in tests/test_gpu_bz2.py).  Reference flow: detecttrails.py:81-109."""
import bz2
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path_factory.mktemp("bz2core") / "bz2_core_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tools", "bz2_core_check.cpp")])
    return str(exe)


def cases():
    rng = np.random.default_rng(1)
    hdr = b"".join(c.ljust(80) for c in (b"SIMPLE  =                    T", b"BITPIX  =                  -32", b"END")).ljust(2880)
    img = rng.normal(0.0, 0.025, (300, 512)).astype(">f4")
    img[100:140, 50:300] = 0.0
    return {
        "text": b"hello hello hello world" * 3,
        "runs": b"a" * 1000 + b"b" * 5 + bytes(300) + b"xyz" * 7 + b"\xfb" * 2000 + b"q" * 4 + b"r" * 259 + b"ssss",
        "allbytes": bytes(range(256)) * 50,
        "noise": rng.integers(0, 256, 300000, dtype=np.uint8).tobytes(),
        "one": b"z",
        "zeros": bytes(1_200_000),
        "fits": hdr + img.tobytes(),
        "few_symbols": bytes(rng.integers(0, 3, 50000, dtype=np.uint8)),
    }


@pytest.mark.parametrize("level", [1, 9])
def test_block_core_against_python_bz2(checker, tmp_path, level):
    for name, plain in cases().items():
        src, dst = tmp_path / f"{name}.bz2", tmp_path / f"{name}.out"
        src.write_bytes(bz2.compress(plain, level))
        r = subprocess.run([checker, str(src), str(dst)], capture_output=True, text=True)
        assert r.returncode == 0, (name, r.stdout, r.stderr)
        assert dst.read_bytes() == plain, name


def test_a_damaged_block_is_noticed(checker, tmp_path):
    plain = cases()["noise"]
    data = bytearray(bz2.compress(plain, 9))
    data[len(data) // 2] ^= 0x04
    src = tmp_path / "bad.bz2"
    src.write_bytes(bytes(data))
    r = subprocess.run([checker, str(src), str(tmp_path / "bad.out")], capture_output=True, text=True)
    assert r.returncode != 0
    with pytest.raises((OSError, ValueError)):
        bz2.decompress(bytes(data))
