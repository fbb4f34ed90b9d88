"""ctypes binding of librsreg.so — the C ABI declared in include/rsreg.h.

There is no CPU fallback anywhere in this package: if the HIP library is missing or no GPU
is usable, the calls raise.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
# RSREG_DIAG=1 (dev tools: per-launch times, wave stamps, dumps): the diagnostic build of the library, librsreg_diag.so, compiled
# with -DRSREG_DIAG -- the shipped librsreg.so has no switch that writes a file or changes a result (csrc/tunables.hpp).
# RSREG_SO (dev, honoured only together with RSREG_DIAG=1): an experiment build of the library.  Without RSREG_DIAG the Python layer
# loads the in-tree librsreg.so and nothing else -- an environment variable alone cannot point the product at another library.
DIAG = os.environ.get("RSREG_DIAG", "") == "1"
SO_PATH = (os.environ.get("RSREG_SO") if DIAG else None) or os.path.join(_HERE, "librsreg_diag.so" if DIAG else "librsreg.so")
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["icp.hip", "ndt.hip", "voxel.hip", "comm.cpp", "voxel_host.cpp", "pcd_host.cpp", "cloud.hip", "edges.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) if os.path.isdir(CSRC) else []
NUM_SUMS = 17
RSREG_ERR_INVALID_ARG, RSREG_ERR_STATE = -1, -9   # include/rsreg.h: rsreg_status
UNIQUE_ID_BYTES = 128

# every symbol include/rsreg.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "rsreg_version", "rsreg_status_string", "rsreg_last_error", "rsreg_device_count",
    "rsreg_ctx_create", "rsreg_ctx_destroy", "rsreg_ctx_synchronize", "rsreg_ctx_prepare", "rsreg_ctx_set_profiling",
    "rsreg_icp_params_default", "rsreg_icp_params_reference", "rsreg_ndt_params_default",
    "rsreg_ndt_params_reference", "rsreg_icp_set_target", "rsreg_icp_set_target_device",
    "rsreg_icp_set_source", "rsreg_icp_set_source_device", "rsreg_icp_align", "rsreg_icp_align_records", "rsreg_icp_begin",
    "rsreg_icp_search", "rsreg_icp_sums", "rsreg_icp_update", "rsreg_icp_end",
    "rsreg_umeyama_from_sums", "rsreg_transform_cloud", "rsreg_approx_voxel_grid", "rsreg_approx_voxel_grid_gpu",
    "rsreg_ndt_set_target", "rsreg_ndt_align", "rsreg_ndt_derivatives", "rsreg_ndt_get_voxels", "rsreg_ndt_set_centroid_mode", "rsreg_ndt_get_centroids",
    "rsreg_comm_unique_id", "rsreg_comm_init", "rsreg_comm_destroy", "rsreg_comm_allreduce_f64",
    "rsreg_cloud_create", "rsreg_cloud_destroy", "rsreg_cloud_upload", "rsreg_cloud_upload_async", "rsreg_cloud_upload_deferred", "rsreg_cloud_download", "rsreg_cloud_download_async", "rsreg_ctx_wait_downloads", "rsreg_cloud_info",
    "rsreg_cloud_device_ptr", "rsreg_cloud_version", "rsreg_cloud_copy", "rsreg_cloud_filter", "rsreg_cloud_filter_async", "rsreg_cloud_transform", "rsreg_cloud_concat",
    "rsreg_icp_set_target_cloud", "rsreg_icp_target_is_cloud", "rsreg_icp_set_source_cloud", "rsreg_icp_align_cloud", "rsreg_ndt_set_target_cloud",
    "rsreg_ndt_align_cloud", "rsreg_ndt_set_target_device", "rsreg_ndt_align_device",
    "rsreg_extract_edge_features", "rsreg_cloud_edge_features", "rsreg_cloud_edge_features_async",
    "rsreg_icp_grid_info", "rsreg_ctx_host_timing", "rsreg_lzf_max_encoded_size", "rsreg_lzf_encode", "rsreg_lzf_decode",
]


class RsregError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__("rsreg status %d (%s)%s" % (status, status_string(status), (": " + detail) if detail else ""))


class IcpParams(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32), ("criteria_mode", C.c_int32), ("pipeline_mode", C.c_int32),
        ("reserved0", C.c_int32), ("max_correspondence_distance", C.c_double),
        ("transformation_epsilon", C.c_double), ("transformation_rotation_epsilon", C.c_double),
        ("euclidean_fitness_epsilon", C.c_double), ("use_reciprocal_correspondences", C.c_int32), ("reserved1", C.c_int32),
        ("trim_overlap_ratio", C.c_double),
    ]


class NdtParams(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32), ("reserved0", C.c_int32), ("transformation_epsilon", C.c_double),
        ("step_size", C.c_double), ("resolution", C.c_double), ("outlier_ratio", C.c_double),
    ]


class IcpResult(C.Structure):
    _fields_ = [
        ("transform", C.c_float * 16), ("converged", C.c_int32), ("state", C.c_int32),
        ("iterations", C.c_int32), ("reserved0", C.c_int32), ("n_correspondences", C.c_uint64),
        ("mse", C.c_double), ("sums_last", C.c_double * NUM_SUMS), ("ms_total", C.c_double),
        ("ms_nn", C.c_double), ("ms_reduce", C.c_double), ("ms_transform", C.c_double),
        ("n_nn_launches", C.c_int32), ("n_scheduled_launches", C.c_int32), ("ms_allreduce", C.c_double),
    ]


class NdtResult(C.Structure):
    _fields_ = [
        ("transform", C.c_float * 16), ("converged", C.c_int32), ("iterations", C.c_int32),
        ("trans_probability", C.c_double), ("score", C.c_double), ("n_voxels", C.c_int32),
        ("n_derivative_passes", C.c_int32), ("ms_total", C.c_double), ("ms_derivatives", C.c_double),
    ]


class GridInfo(C.Structure):
    _fields_ = [
        ("origin", C.c_float * 3), ("cell_size", C.c_float), ("dims", C.c_int32 * 3),
        ("n_target_points", C.c_uint32), ("n_unique_points", C.c_uint32), ("n_cells", C.c_uint32),
        ("max_points_per_cell", C.c_uint32), ("ms_build", C.c_double),
        ("index_kind", C.c_uint32), ("n_source_distinct", C.c_uint32), ("index_bytes", C.c_uint64),
    ]


class HostTiming(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("source_stage_wait", "source_pack", "target_stage_wait", "target_pack", "target_build", "align",
                                           "aligned_copy", "loop_enqueue")]


def hipcc_command(out=SO_PATH):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    extra = (["-DRSREG_DIAG"] if DIAG else []) + os.environ.get("RSREG_CXXFLAGS", "").split()   # (dev: -D switches of experiment builds)
    return [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
            "-Wno-unused-result", *extra, *srcs, "-o", out, "-ldl"]


def needs_build():
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "rsreg.h"),
                                                                os.path.join(ROOT, "include", "rsreg", "lzf.hpp")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


DIAG_SO_PATH = os.path.join(_HERE, "librsreg_diag.so")


def _flags(diag=DIAG):
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", *(["-DRSREG_DIAG"] if diag else []),
            *os.environ.get("RSREG_CXXFLAGS", "").split()]   # (dev: -D switches of experiment builds)


def build(force=False, verbose=False, out=SO_PATH, obj_dir=None):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU): every translation unit to an
    object of its own, side by side, then one link -- the same flags and sources as hipcc_command()'s single call, a third
    of its wall time, and an edit of one file recompiles that file only."""
    if not force and out == SO_PATH and not needs_build():
        return SO_PATH
    if out == SO_PATH and DIAG and os.environ.get("RSREG_SO"):
        raise RuntimeError("RSREG_SO names a library that is missing or older than the sources: build it where it came from")
    from concurrent.futures import ThreadPoolExecutor

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj_dir = obj_dir or os.path.join(CSRC, "_obj" + ("" if out == SO_PATH else "_" + os.path.splitext(os.path.basename(out))[0]))
    os.makedirs(obj_dir, exist_ok=True)
    flags = _flags(DIAG or out == DIAG_SO_PATH)
    stamp = os.path.join(obj_dir, "flags.txt")
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
    hdrs = [os.path.join(CSRC, f) for f in HEADERS] + [os.path.join(ROOT, "include", "rsreg.h"), os.path.join(ROOT, "include", "rsreg", "lzf.hpp")]
    newest_hdr = max(os.path.getmtime(h) for h in hdrs if os.path.exists(h))

    def compile_one(src):
        s, o = os.path.join(CSRC, src), os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        if not force and same_flags and os.path.exists(o) and os.path.getmtime(o) > max(os.path.getmtime(s), newest_hdr):
            return o, 0, ""
        r = subprocess.run([hipcc, *flags, "-c", s, "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return o, r.returncode, r.stdout

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        results = list(ex.map(compile_one, SOURCES))
    for o, rc, log in results:
        if rc != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (o, log[-4000:]))
        if verbose and log:
            print(log)
    open(stamp, "w").write(" ".join(flags))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[o for o, _, _ in results], "-o", out, "-ldl"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout[-4000:])
    return out


def build_diag(force=False):
    """The diagnostic build (-DRSREG_DIAG: dumps, per-wave stamps, per-launch times) next to the shipped library: what the dev
    tools and the two tests that read a dump load (RSREG_DIAG=1 in a process's environment makes lib() load it)."""
    if not force and os.path.exists(DIAG_SO_PATH):
        t = os.path.getmtime(DIAG_SO_PATH)
        deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "rsreg.h")]
        if not any(os.path.getmtime(d) > t for d in deps if os.path.exists(d)):
            return DIAG_SO_PATH
    return build(force=True, out=DIAG_SO_PATH)


_lib = None


def lib():
    """The loaded C ABI.  Raises if the extension has not been built: no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            "librsreg.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
            "this package has no CPU fallback." % SO_PATH)
    L = C.CDLL(SO_PATH)
    vp, sz, i32, dbl = C.c_void_p, C.c_size_t, C.c_int, C.c_double
    L.rsreg_version.restype = i32
    L.rsreg_status_string.restype = C.c_char_p
    L.rsreg_status_string.argtypes = [i32]
    L.rsreg_last_error.restype = C.c_char_p
    L.rsreg_last_error.argtypes = [vp]
    L.rsreg_device_count.argtypes = [C.POINTER(i32)]
    L.rsreg_ctx_create.argtypes = [i32, vp, C.POINTER(vp)]
    L.rsreg_ctx_destroy.argtypes = [vp]
    L.rsreg_ctx_synchronize.argtypes = [vp]
    L.rsreg_ctx_prepare.argtypes = [vp, sz, sz, C.c_uint]
    L.rsreg_ctx_set_profiling.argtypes = [vp, i32]
    for f in ("rsreg_icp_params_default", "rsreg_icp_params_reference"):
        getattr(L, f).argtypes = [C.POINTER(IcpParams)]
        getattr(L, f).restype = None
    for f in ("rsreg_ndt_params_default", "rsreg_ndt_params_reference"):
        getattr(L, f).argtypes = [C.POINTER(NdtParams)]
        getattr(L, f).restype = None
    L.rsreg_icp_set_target.argtypes = [vp, vp, sz, sz, i32, dbl]
    L.rsreg_icp_set_target_device.argtypes = [vp, vp, sz, sz, i32, dbl]
    L.rsreg_icp_set_source.argtypes = [vp, vp, sz, sz, i32]
    L.rsreg_icp_set_source_device.argtypes = [vp, vp, sz, sz, i32]
    L.rsreg_icp_align.argtypes = [vp, vp, C.POINTER(IcpParams), C.POINTER(IcpResult), vp, sz]
    L.rsreg_icp_align_records.argtypes = [vp, vp, C.POINTER(IcpParams), C.POINTER(IcpResult), vp, vp, sz]
    L.rsreg_icp_begin.argtypes = [vp, vp, C.POINTER(IcpParams)]
    L.rsreg_icp_search.argtypes = [vp, vp, vp]
    L.rsreg_icp_sums.argtypes = [vp, vp]
    L.rsreg_icp_update.argtypes = [vp, vp, vp, C.POINTER(i32)]
    L.rsreg_icp_end.argtypes = [vp, C.POINTER(IcpResult), vp, sz]
    L.rsreg_umeyama_from_sums.argtypes = [vp, vp]
    L.rsreg_transform_cloud.argtypes = [vp, vp, vp, sz, sz, i32, vp]
    L.rsreg_approx_voxel_grid.argtypes = [vp, sz, sz, vp, vp, C.POINTER(sz)]
    L.rsreg_approx_voxel_grid_gpu.argtypes = [vp, vp, sz, sz, vp, vp, C.POINTER(sz)]
    L.rsreg_ndt_set_target.argtypes = [vp, vp, sz, sz, i32, dbl]
    L.rsreg_ndt_align.argtypes = [vp, vp, sz, sz, i32, vp, C.POINTER(NdtParams), C.POINTER(NdtResult), vp, sz]
    L.rsreg_ndt_derivatives.argtypes = [vp, vp, sz, sz, i32, vp, C.POINTER(dbl), vp, vp]
    L.rsreg_ndt_get_voxels.argtypes = [vp, C.POINTER(C.c_int32), vp, vp, C.c_int32]
    L.rsreg_ndt_set_centroid_mode.argtypes = [vp, C.c_int]
    L.rsreg_ndt_get_centroids.argtypes = [vp, vp, C.c_int32]
    L.rsreg_comm_unique_id.argtypes = [vp]
    L.rsreg_comm_init.argtypes = [vp, vp, i32, i32]
    L.rsreg_comm_destroy.argtypes = [vp]
    L.rsreg_comm_allreduce_f64.argtypes = [vp, vp, i32]
    L.rsreg_icp_grid_info.argtypes = [vp, C.POINTER(GridInfo)]
    L.rsreg_ctx_host_timing.argtypes = [vp, C.POINTER(HostTiming)]
    u32 = C.c_uint32
    L.rsreg_cloud_create.argtypes = [vp, C.POINTER(vp)]
    L.rsreg_cloud_destroy.argtypes = [vp]
    L.rsreg_cloud_upload.argtypes = [vp, vp, sz, sz, u32, u32, i32]
    L.rsreg_cloud_upload_async.argtypes = [vp, vp, sz, sz, u32, u32, i32]
    L.rsreg_cloud_upload_deferred.argtypes = [vp, vp, sz, sz, u32, u32, i32]
    L.rsreg_cloud_download.argtypes = [vp, vp, sz]
    L.rsreg_cloud_download_async.argtypes = [vp, vp, sz]
    L.rsreg_ctx_wait_downloads.argtypes = [vp]
    L.rsreg_cloud_info.argtypes = [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(u32), C.POINTER(u32), C.POINTER(i32)]
    L.rsreg_cloud_device_ptr.argtypes = [vp]
    L.rsreg_cloud_device_ptr.restype = vp
    L.rsreg_cloud_version.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.rsreg_cloud_copy.argtypes = [vp, vp, vp]
    L.rsreg_cloud_filter.argtypes = [vp, vp, vp, vp]
    L.rsreg_cloud_filter_async.argtypes = [vp, vp, vp, vp]
    L.rsreg_cloud_transform.argtypes = [vp, vp, vp, vp]
    L.rsreg_cloud_concat.argtypes = [vp, vp, vp, vp]
    L.rsreg_icp_set_target_cloud.argtypes = [vp, vp, dbl]
    L.rsreg_icp_target_is_cloud.argtypes = [vp, vp, dbl]
    L.rsreg_icp_set_source_cloud.argtypes = [vp, vp]
    L.rsreg_icp_align_cloud.argtypes = [vp, vp, C.POINTER(IcpParams), C.POINTER(IcpResult), vp]
    L.rsreg_ndt_set_target_cloud.argtypes = [vp, vp, dbl]
    L.rsreg_ndt_align_cloud.argtypes = [vp, vp, vp, C.POINTER(NdtParams), C.POINTER(NdtResult), vp]
    L.rsreg_ndt_set_target_device.argtypes = [vp, vp, sz, sz, i32, dbl]
    L.rsreg_ndt_align_device.argtypes = [vp, vp, sz, sz, i32, vp, C.POINTER(NdtParams), C.POINTER(NdtResult), vp]
    L.rsreg_extract_edge_features.argtypes = [vp, vp, u32, u32, sz, vp, vp, C.POINTER(sz)]
    L.rsreg_cloud_edge_features.argtypes = [vp, vp, vp]
    L.rsreg_cloud_edge_features_async.argtypes = [vp, vp, vp]
    L.rsreg_lzf_max_encoded_size.argtypes = [sz]
    L.rsreg_lzf_max_encoded_size.restype = sz
    for f in ("rsreg_lzf_encode", "rsreg_lzf_decode"):
        getattr(L, f).argtypes = [vp, sz, vp, sz]
        getattr(L, f).restype = sz
    _lib = L
    return L


def status_string(status):
    try:
        return lib().rsreg_status_string(status).decode()
    except Exception:
        return "?"


def check(status, ctx=None):
    if status != 0:
        detail = ""
        if ctx:
            detail = lib().rsreg_last_error(ctx).decode()
        raise RsregError(status, detail)
