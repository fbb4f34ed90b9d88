"""CPU: PCD `DATA binary_compressed` (what pcl::io::savePCDFileBinaryCompressed writes and
loadPCDFile reads -- src/main.cpp:53,81,87 go through whatever mode a capture was stored in).
The LZF coder is checked against an independent decoder written here from the stream format,
a hand-assembled known-answer stream, and round trips; the PCD layer in both host languages
(cloud.py over the C ABI, pcl_compat.hpp compiled with g++) must read each other's files."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def py_lzf_decode(data, usize):
    """Independent LZF decoder (stream format: lzf.hpp header comment)."""
    out = bytearray()
    i = 0
    while i < len(data):
        ctrl = data[i]
        i += 1
        if ctrl < 32:
            out += data[i:i + ctrl + 1]
            i += ctrl + 1
        else:
            ln = ctrl >> 5
            if ln == 7:
                ln += data[i]
                i += 1
            dist = ((ctrl & 31) << 8 | data[i]) + 1
            i += 1
            for _ in range(ln + 2):
                out.append(out[-dist])
    assert len(out) == usize
    return bytes(out)


@pytest.fixture(scope="module")
def L(rs):
    from rsreg_amd import lib
    lib.build()
    return lib.lib()


def _enc(L, data):
    cap = L.rsreg_lzf_max_encoded_size(len(data))
    src = np.frombuffer(data, np.uint8)
    out = np.empty(max(cap, 1), np.uint8)
    n = L.rsreg_lzf_encode(src.ctypes.data if len(src) else None, len(src), out.ctypes.data, cap)
    return out[:n].tobytes()


def _dec(L, data, usize):
    src = np.frombuffer(data, np.uint8)
    out = np.empty(max(usize, 1), np.uint8)
    n = L.rsreg_lzf_decode(src.ctypes.data if len(src) else None, len(src), out.ctypes.data, usize)
    return out[:n].tobytes()


def test_lzf_known_answer_stream(L):
    # literal "abc" (ctrl 2), back reference len 6 = 4 + 2 at distance 3 (ctrl 4 << 5, low byte 2),
    # long back reference len 7 + 5 + 2 = 14 at distance 9 (ctrl 7 << 5, extra 5, low byte 8), literal "Z"
    stream = bytes([2]) + b"abc" + bytes([4 << 5, 2]) + bytes([7 << 5, 5, 8]) + bytes([0]) + b"Z"
    want = b"abc" + b"abcabc" + (b"abcabcabc" * 2)[:14] + b"Z"
    assert py_lzf_decode(stream, len(want)) == want
    assert _dec(L, stream, len(want)) == want
    assert _dec(L, stream, len(want) - 1) == b""            # capacity too small -> failure, not overflow
    assert _dec(L, bytes([1 << 5, 0]), 16) == b""           # reference before the start of the output
    assert _dec(L, bytes([5]) + b"ab", 16) == b""           # truncated literal run


@pytest.mark.parametrize("kind", ["empty", "one", "zeros", "random", "floats", "text", "long_match"])
def test_lzf_round_trip(L, kind):
    rng = np.random.default_rng(11)
    data = {
        "empty": b"", "one": b"x", "zeros": bytes(100000),
        "random": rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(),
        "floats": np.round(rng.random(30000).astype(np.float32) * 4, 3).tobytes(),
        "text": b"the quick brown fox jumps over the lazy dog. " * 300,
        "long_match": (bytes(range(256)) * 40) + b"tail",
    }[kind]
    enc = _enc(L, data)
    assert (len(enc) > 0) == (len(data) > 0)
    assert len(enc) <= L.rsreg_lzf_max_encoded_size(len(data))
    assert _dec(L, enc, len(data)) == data
    assert py_lzf_decode(enc, len(data)) == data            # the stream is standard LZF, not a private dialect
    if kind in ("zeros", "text", "long_match"):
        assert len(enc) < len(data) // 8
    if kind == "floats":
        assert len(enc) < len(data)


def _cloud(rs, w=250, h=200):
    c = rs.synth.render_frame(1, (w, h), "bench")
    c.points["x"][7] = np.nan                               # a non-finite point survives bit for bit
    return c


def _same(a, b):
    assert (a.width, a.height, len(a)) == (b.width, b.height, len(b))
    for f in ("x", "y", "z", "rgba"):
        np.testing.assert_array_equal(a.points[f].view(np.uint32), b.points[f].view(np.uint32))


def test_pcd_binary_compressed_round_trip(tmp_path, rs, L):
    c = _cloud(rs)
    p = str(tmp_path / "c.pcd")
    rs.save_pcd(p, c, compressed=True)
    raw = open(p, "rb").read()
    head, body = raw.split(b"DATA binary_compressed\n", 1)
    assert b"FIELDS x y z rgb" in head and b"WIDTH 250" in head and b"HEIGHT 200" in head
    csize, usize = np.frombuffer(body, "<u4", count=2)
    assert usize == 16 * len(c) and csize == len(body) - 8 and csize < usize
    soa = py_lzf_decode(body[8:], int(usize))               # fields one after the other
    np.testing.assert_array_equal(np.frombuffer(soa, "<f4", count=len(c), offset=4 * len(c)), c.points["y"])
    np.testing.assert_array_equal(np.frombuffer(soa, "<u4", count=len(c), offset=12 * len(c)), c.points["rgba"])
    back = rs.load_pcd(p)
    _same(back, c)
    assert not back.is_dense
    empty = rs.PointCloud()
    rs.save_pcd(str(tmp_path / "e.pcd"), empty, compressed=True)
    assert len(rs.load_pcd(str(tmp_path / "e.pcd"))) == 0
    open(str(tmp_path / "bad.pcd"), "wb").write(head + b"DATA binary_compressed\n" + body[:-40])
    with pytest.raises(ValueError):
        rs.load_pcd(str(tmp_path / "bad.pcd"))


def test_pcd_cross_language(tmp_path, rs, L):
    """cloud.py and pcl_compat.hpp read each other's binary and binary_compressed files."""
    from rsreg_amd import lib
    exe = str(tmp_path / "pcd_convert")
    pkg = os.path.dirname(lib.SO_PATH)
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "pcd_convert.cpp"),
                    "-o", exe, "-L", pkg, "-lrsreg", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"], check=True)
    c = _cloud(rs)
    a, b, d, e = (str(tmp_path / n) for n in ("py_c.pcd", "cpp_c.pcd", "cpp_b.pcd", "py_b.pcd"))
    rs.save_pcd(a, c, compressed=True)
    out = subprocess.run([exe, a, b, "binary_compressed"], check=True, stdout=subprocess.PIPE, text=True).stdout.split()
    assert [int(v) for v in out] == [len(c), c.width, c.height, 0]
    _same(rs.load_pcd(b), c)                                 # C++ wrote compressed, Python reads it
    subprocess.run([exe, b, d, "binary"], check=True, stdout=subprocess.PIPE)
    _same(rs.load_pcd(d), c)                                 # C++ read its own compressed file
    rs.save_pcd(e, c, binary=True)
    subprocess.run([exe, e, b, "binary_compressed"], check=True, stdout=subprocess.PIPE)
    _same(rs.load_pcd(b), c)
    assert subprocess.run([exe, str(tmp_path / "missing.pcd"), b, "binary"], stdout=subprocess.PIPE, stderr=subprocess.PIPE).returncode == 1
