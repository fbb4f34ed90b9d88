"""GPU: the asynchronous cloud calls of a context -- rsreg_cloud_upload_deferred / _async, rsreg_cloud_edge_features_async,
rsreg_cloud_filter_async, rsreg_cloud_download_async, concat, transform, copy, dropping handles -- interleaved in random
order over a few hundred seeds, every record compared with what the synchronous twins produce.  They serve the frame
loops of the three schemes (types.hpp:30-43, incremental_icp.hpp:51-66, icp_edge_based_registration.hpp:71-123): frames
staged by the upload worker, features and filters of the frames ahead queued by the side worker, the merged cloud
streaming home through the download worker, buffers recycled through the context's pool."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(rs):
    from rsreg_amd import api as a, lib
    lib.build()
    if a.device_count() < 1:
        pytest.fail("no HIP device: the product has no CPU fallback")
    return a


def _same(a, b):
    assert (len(a), a.width, a.height, a.is_dense) == (len(b), b.width, b.height, b.is_dense)
    for f in ("x", "y", "z", "w", "rgba"):
        np.testing.assert_array_equal(a.points[f].view(np.uint32), b.points[f].view(np.uint32))


def test_random_interleavings_of_the_async_cloud_calls(api, rs):
    frames = [rs.synth.render_frame(k, (96, 72), "bench") for k in range(4)]     # organized: the edge extractor takes them
    leaf = (0.02, 0.02, 0.02)
    ref_ctx = api.Context(0)

    # the synchronous twins, once per frame
    def vox(ctx, cloud):
        v = api.ApproximateVoxelGrid(ctx)
        v.setLeafSize(*leaf)
        v.setInputCloud(cloud)
        return v

    want_edges, want_filt, want_edge_filt = [], [], []
    for f in frames:
        d = api.DeviceCloud(f, ref_ctx)
        e = api.extract_edge_features(d)
        want_edges.append(e.download())
        want_filt.append(vox(ref_ctx, d).filter().download())
        want_edge_filt.append(vox(ref_ctx, e).filter().download())
    T = rs.synth.small_transform(0.7, (0.01, -0.02, 0.005)).astype(np.float32)
    want_moved = [api.transformPointCloud(api.DeviceCloud(f, ref_ctx), T).download() for f in frames]

    n_seeds = 240
    ctx = api.Context(0)
    for seed in range(n_seeds):
        rng = np.random.default_rng(seed)
        if seed % 40 == 39:           # a context closed with whatever the seeds before left queued, a fresh one after it
            ctx.close()
            ctx = api.Context(0)
        live = []                     # (device cloud, the host cloud it must equal)
        pending_down = []             # (host array slice, expected cloud)
        out = np.zeros(sum(len(f) for f in frames) * 3, api.POINT_DTYPE)
        at = 0
        for _ in range(int(rng.integers(6, 14))):
            op = int(rng.integers(0, 9))
            k = int(rng.integers(0, len(frames)))
            if op == 0:
                live.append((api.DeviceCloud(ctx=ctx).upload_deferred(frames[k]), frames[k]))
            elif op == 1:
                live.append((api.DeviceCloud(ctx=ctx).upload_async(frames[k]), frames[k]))
            elif op == 2:             # features of a frame that may still be on its way up
                d = api.DeviceCloud(ctx=ctx).upload_deferred(frames[k])
                live.append((api.extract_edge_features_async(d), want_edges[k]))
            elif op == 3:             # filter chained on a not-yet-run extraction
                d = api.DeviceCloud(ctx=ctx).upload_deferred(frames[k])
                e = api.extract_edge_features_async(d)
                live.append((vox(ctx, e).filter_async(), want_edge_filt[k]))
            elif op == 4:             # filter of a frame still uploading
                d = api.DeviceCloud(ctx=ctx).upload_async(frames[k])
                live.append((vox(ctx, d).filter_async(), want_filt[k]))
            elif op == 5 and live:    # a result streams home while things go on
                c, w = live[int(rng.integers(0, len(live)))]
                n = len(w)
                if at + n <= len(out):
                    assert c.download_async(out, at) == n
                    pending_down.append((at, n, w))
                    at += n
            elif op == 6 and len(live) >= 2:   # concatenation of two results, in place or into a new handle
                (a, wa), (b, wb) = live[int(rng.integers(0, len(live)))], live[int(rng.integers(0, len(live)))]
                if a is not b:
                    live.append((a + b, wa + wb))
            elif op == 7 and live:    # a handle dropped with its job possibly still queued
                live.pop(int(rng.integers(0, len(live))))
            elif op == 8:
                d = api.DeviceCloud(ctx=ctx).upload_deferred(frames[k])
                live.append((api.transformPointCloud(d, T), want_moved[k]))
        order = rng.permutation(len(live))
        for j in order[: max(1, len(order) // 2)]:     # half of what is alive is read back, in another order than it was made
            c, w = live[int(j)]
            _same(c.download(), w)
        ctx.wait_downloads()
        for lo, n, w in pending_down:
            for f in ("x", "y", "z", "w", "rgba"):
                np.testing.assert_array_equal(out[f][lo:lo + n].view(np.uint32), w.points[f].view(np.uint32))
        del live
    ctx.close()
