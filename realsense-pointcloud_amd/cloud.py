"""Point-cloud container and PCD I/O mirroring what the reference hands to the hot path.

The reference's clouds are ``pcl::PointCloud<pcl::PointXYZRGB>`` (src/types.hpp:8-10) loaded
with ``pcl::io::loadPCDFile`` / saved with ``savePCDFileBinary`` (src/main.cpp:53,81,87).
``POINT_DTYPE`` is byte-compatible with ``pcl::PointXYZRGB`` (32 B).
"""
import numpy as np

POINT_DTYPE = np.dtype(
    {
        "names": ["x", "y", "z", "w", "rgba"],
        "formats": ["<f4", "<f4", "<f4", "<f4", "<u4"],
        "offsets": [0, 4, 8, 12, 16],
        "itemsize": 32,
    }
)


class PointCloud:
    """width/height/is_dense/points, like pcl::PointCloud<PointXYZRGB>."""

    def __init__(self, points=None, width=None, height=1, is_dense=True):
        if points is None:
            points = np.zeros(0, POINT_DTYPE)
        assert points.dtype == POINT_DTYPE
        self.points = points
        self.height = int(height)
        self.width = int(width) if width is not None else len(points) // max(self.height, 1)
        self.is_dense = bool(is_dense)

    def __len__(self):
        return len(self.points)

    @property
    def xyz(self):
        p = self.points
        return np.stack([p["x"], p["y"], p["z"]], axis=1)

    @classmethod
    def from_xyz(cls, xyz, rgba=None, width=None, height=1, is_dense=True):
        xyz = np.asarray(xyz, np.float32)
        pts = np.zeros(len(xyz), POINT_DTYPE)
        pts["x"], pts["y"], pts["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
        pts["w"] = 1.0
        pts["rgba"] = 0xFF000000 if rgba is None else rgba
        return cls(pts, width=width, height=height, is_dense=is_dense)

    def copy(self):
        return PointCloud(self.points.copy(), self.width, self.height, self.is_dense)

    def crop(self, u0, v0, w, h, step=1):
        """Window (and optional stride) of an organized cloud, still organized."""
        assert self.height > 1
        img = self.points.reshape(self.height, self.width)[v0:v0 + h:step, u0:u0 + w:step]
        return PointCloud(np.ascontiguousarray(img).reshape(-1), width=img.shape[1],
                          height=img.shape[0], is_dense=self.is_dense)

    def __add__(self, other):
        """pcl::PointCloud::operator+ (incremental_icp.hpp:64, icp_edge...hpp:57,119-120):
        concatenate, width = size, height = 1, is_dense = both dense."""
        na, nb = len(self.points), len(other.points)
        pts = np.zeros(na + nb, POINT_DTYPE)  # (np.concatenate would repack the padded record layout)
        pts[:na] = self.points
        pts[na:] = other.points
        return PointCloud(pts, width=len(pts), height=1, is_dense=self.is_dense and other.is_dense)


_PCD_TYPES = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 1): "u1", ("U", 2): "<u2", ("U", 4): "<u4",
              ("I", 1): "i1", ("I", 2): "<i2", ("I", 4): "<i4"}


def _lzf(fn, data, capacity):
    """LZF coder of the C ABI (include/rsreg/lzf.hpp through librsreg.so; host code, no GPU)."""
    import ctypes as C

    from . import lib as _l
    src = np.frombuffer(data, np.uint8)
    out = np.empty(max(capacity, 1), np.uint8)
    n = getattr(_l.lib(), fn)(src.ctypes.data if len(src) else None, len(src), out.ctypes.data, capacity)
    return out[:n].tobytes()


def load_pcd(path):
    """Read an ASCII, binary or binary_compressed .pcd with x y z [rgb|rgba] fields
    (pcl::io::loadPCDFile, src/main.cpp:81)."""
    with open(path, "rb") as f:
        raw = f.read()
    hdr = {}
    pos = 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        key, _, val = line.partition(" ")
        hdr[key.upper()] = val.split()
        if key.upper() == "DATA":
            break
    fields = hdr["FIELDS"]
    sizes = [int(s) for s in hdr["SIZE"]]
    types = hdr["TYPE"]
    counts = [int(c) for c in hdr.get("COUNT", ["1"] * len(fields))]
    width, height = int(hdr["WIDTH"][0]), int(hdr["HEIGHT"][0])
    n = int(hdr.get("POINTS", [width * height])[0])
    mode = hdr["DATA"][0].lower()
    assert all(c == 1 for c in counts), "COUNT != 1 unsupported"
    if len(sizes) != len(fields) or len(types) != len(fields) or any(s <= 0 for s in sizes) or n < 0:
        raise ValueError("malformed PCD header")
    if mode == "ascii":
        text = raw[pos:].decode("ascii").split()
        cols = len(fields)
        tab = np.array(text[: n * cols], dtype=object).reshape(n, cols)
        col = {}
        for k, name in enumerate(fields):
            dt = _PCD_TYPES[(types[k], sizes[k])]
            col[name] = np.array([float(v) if types[k] == "F" else int(v) for v in tab[:, k]]).astype(dt)
    elif mode == "binary":
        dt = np.dtype([(name, _PCD_TYPES[(types[k], sizes[k])]) for k, name in enumerate(fields)])
        arr = np.frombuffer(raw, dtype=dt, count=n, offset=pos)
        col = {name: arr[name] for name in fields}
    elif mode == "binary_compressed":
        # u32 compressed size, u32 uncompressed size, one LZF stream over the fields laid out
        # one after the other (all x, all y, ...)
        if len(raw) < pos + 8:
            raise ValueError("truncated binary_compressed PCD")
        csize, usize = (int(v) for v in np.frombuffer(raw, "<u4", count=2, offset=pos))
        # validate before anything is allocated from the (untrusted) header fields
        if usize != n * sum(sizes) or pos + 8 + csize > len(raw):
            raise ValueError("corrupt binary_compressed PCD: sizes in the header do not match the file")
        body = _lzf("rsreg_lzf_decode", raw[pos + 8:pos + 8 + csize], usize)
        if len(body) != usize:
            raise ValueError("corrupt binary_compressed PCD body")
        col, off = {}, 0
        for k, name in enumerate(fields):
            col[name] = np.frombuffer(body, _PCD_TYPES[(types[k], sizes[k])], count=n, offset=off)
            off += n * sizes[k]
    else:
        raise ValueError("unsupported PCD DATA mode: " + mode)
    pts = np.zeros(n, POINT_DTYPE)
    for a in "xyz":
        pts[a] = col[a].astype(np.float32)
    pts["w"] = 1.0
    for name in ("rgb", "rgba"):
        if name in col:
            c = col[name]
            # PCL stores packed colour either as a float whose BITS are the colour or as uint32
            pts["rgba"] = c.view(np.uint32) if c.dtype == np.float32 else c.astype(np.uint32)
    dense = bool(np.isfinite(pts["x"]).all() and np.isfinite(pts["y"]).all() and np.isfinite(pts["z"]).all())
    return PointCloud(pts, width=width, height=height, is_dense=dense)


def save_pcd(path, cloud, binary=True, compressed=False):
    """pcl::io::savePCDFileBinary / savePCDFileBinaryCompressed / savePCDFileASCII layout:
    FIELDS x y z rgb, SIZE 4 4 4 4, TYPE F F F F (src/main.cpp:53,87)."""
    n = len(cloud)
    if compressed:
        binary = True
    hdr = (
        "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\n"
        "TYPE F F F F\nCOUNT 1 1 1 1\nWIDTH %d\nHEIGHT %d\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n"
        % (cloud.width, cloud.height, n, "binary_compressed" if compressed else ("binary" if binary else "ascii"))
    )
    p = cloud.points
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        if compressed:
            soa = b"".join(np.ascontiguousarray(p[k]).tobytes() for k in ("x", "y", "z", "rgba"))
            from . import lib as _l
            body = _lzf("rsreg_lzf_encode", soa, int(_l.lib().rsreg_lzf_max_encoded_size(len(soa))))
            if soa and not body:
                raise ValueError("LZF encoding failed")
            f.write(np.array([len(body), len(soa)], "<u4").tobytes())
            f.write(body)
        elif binary:
            rec = np.zeros(n, np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgb", "<u4")]))
            rec["x"], rec["y"], rec["z"], rec["rgb"] = p["x"], p["y"], p["z"], p["rgba"]
            f.write(rec.tobytes())
        else:
            rgbf = p["rgba"].view(np.float32)
            for i in range(n):
                f.write(("%.9g %.9g %.9g %.9g\n" % (p["x"][i], p["y"][i], p["z"][i], rgbf[i])).encode("ascii"))
