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


def test_malformed_pcd_headers_are_rejected_under_asan(tmp_path, rs, L):
    """An untrusted header must never index past what it declares (the binary_compressed branch used to copy four
    bytes per field whatever SIZE said): the C++ reader, built with the CPU AddressSanitizer, rejects every malformed
    file below with a status instead of reading out of bounds; the Python reader raises."""
    from rsreg_amd import lib
    exe = str(tmp_path / "pcd_convert_asan")
    # header-only path: lzf.hpp + pcl_compat.hpp; the io functions need nothing from librsreg.so at run time, but the
    # header declares the C ABI, so the library is linked like in the cross-language test
    pkg = os.path.dirname(lib.SO_PATH)
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "pcd_convert.cpp"), "-o", exe, "-L", pkg, "-lrsreg", "-Wl,-rpath," + pkg,
                    "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"], check=True)
    c = _cloud(rs, 40, 30)
    good = str(tmp_path / "good.pcd")
    rs.save_pcd(good, c, compressed=True)
    raw = open(good, "rb").read()
    head, body = raw.split(b"DATA binary_compressed\n", 1)
    n = len(c)

    def hdr(fields="x y z rgb", size="4 4 4 4", typ="F F F F", points=n, mode="binary_compressed"):
        return ("# .PCD v0.7\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\n"
                "POINTS %d\nDATA %s\n" % (fields, size, typ, " ".join("1" for _ in fields.split()), points, points, mode)).encode()

    soa1 = bytes(3 * n)   # what SIZE 1 1 1 promises: three bytes per point
    cases = {
        "size1": hdr("x y z", "1 1 1", "U U U") + np.array([len(_enc(L, soa1)), len(soa1)], "<u4").tobytes() + _enc(L, soa1),
        "short_sizes": hdr(size="4 4") + body,
        "negative_size": hdr(size="4 4 -4 4") + body,
        "huge_csize": hdr() + np.array([0xfffffff0, 16 * n], "<u4").tobytes() + body[8:],
        "huge_usize": hdr() + np.array([len(body) - 8, 0xfffffff0], "<u4").tobytes() + body[8:],
        "truncated_body": hdr() + body[:-30],
        "no_sizes_at_all": hdr().replace(b"SIZE 4 4 4 4\n", b"") + body,
        "binary_too_short": hdr(mode="binary") + bytes(16 * n - 5),
        "points_beyond_file": hdr(points=1 << 30, mode="binary") + bytes(64),
    }
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    env.pop("LD_PRELOAD", None)
    for name, blob in cases.items():
        p = str(tmp_path / (name + ".pcd"))
        open(p, "wb").write(blob)
        r = subprocess.run([exe, p, str(tmp_path / "out.pcd"), "binary"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        assert r.returncode == 1 and "loadPCDFile" in r.stderr and "AddressSanitizer" not in r.stderr, (name, r.returncode, r.stderr[-600:])
        if name == "size1":   # a consistent file of one-byte fields: the Python reader handles any field width
            assert len(rs.load_pcd(p)) == n
        else:
            with pytest.raises((ValueError, KeyError, AssertionError)):
                rs.load_pcd(p)
    r = subprocess.run([exe, good, str(tmp_path / "out.pcd"), "binary"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert r.returncode == 0, r.stderr[-600:]
