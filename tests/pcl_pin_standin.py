"""TEST INFRASTRUCTURE: an ORACLE-generated stand-in for tests/golden/pcl_pin.npz -- same keys, same layouts, produced by the CPU
oracle (oracle/*.c) and the scheme logic over it (tests/oracle_backend.py) instead of by PCL.  It exists so that the bodies of
tests/test_pcl_pin.py run somewhere before a machine with PCL produces the real fixture (oracle/pcl_harness); it is written
to a temporary directory only and PINS NOTHING: oracle against oracle proves the plumbing, engine against oracle is the
parity the other GPU tests already assert.  Sizes are smaller than the harness's (50 k frames for the pair too): the schema
does not depend on them."""
import os

import numpy as np

SOURCE = "ORACLE-GENERATED STAND-IN, NOT PCL OUTPUT (tests/pcl_pin_standin.py)"


def _corr_matrix(index, d2):
    m = index >= 0
    return np.stack([np.nonzero(m)[0].astype(np.float64), index[m].astype(np.float64), d2[m].astype(np.float64)], 1)


def make_standin(path, orc, rs, tmp_dir):
    """Writes the stand-in to `path` (an .npz outside tests/golden) and returns its key list."""
    from oracle_backend import OracleBackend
    from rsreg_amd import schemes
    assert os.path.abspath(os.path.dirname(path)) != os.path.abspath(os.path.join(os.path.dirname(__file__), "golden"))
    synth = rs.synth
    pair = [synth.render_frame(k, "50k", "parity") for k in (0, 1)]
    chain = [synth.render_frame(k, "50k", "parity") for k in range(4)]
    tgt, src = np.ascontiguousarray(pair[0].points), np.ascontiguousarray(pair[1].points)
    out = {"standin_source": np.array(SOURCE), "in_pair0": tgt, "in_pair1": src,
           "in_guess": synth.small_transform(0.1, (0.0, 0.0, 0.0)), "in_rads": np.array([-0.0026179939])}
    for k, c in enumerate(chain):
        out["in_chain%d" % k] = np.ascontiguousarray(c.points)
        out["in_chain%d_shape" % k] = np.array([c.width, c.height])
    # (1)-(2) first-iteration correspondences and the reference-parameter transform, PCL's float sums
    o = orc.IcpOracle()
    o.set_target(tgt)
    o.set_source(src)
    p = orc.IcpParams.reference()
    p.accum_mode = 0
    o.begin(None, p)
    oi, od = o.search()
    out["corr_it0"] = _corr_matrix(oi, od)
    r = o.align(None, p)
    out["icp_reference_T"] = np.asarray(r.T, np.float64)
    out["icp_reference_meta"] = np.array([[float(r.converged), float(r.mse)]])
    for iters in (1, 5, 30):
        q = orc.IcpParams.default()
        q.max_iterations, q.max_correspondence_distance, q.accum_mode = iters, 0.05, 0
        out["icp_gate5cm_%dit_T" % iters] = np.asarray(o.align(None, q).T, np.float64)
    # (3) ApproximateVoxelGrid at 1 cm and at PCL's default leaf
    out["voxel_1cm"] = orc.approx_voxel_grid(src, (0.01, 0.01, 0.01))
    out["voxel_default"] = orc.approx_voxel_grid(src, (1.0, 1.0, 1.0))
    # (4) NDT, the reference's constants, PCL's float centroids
    n = orc.NdtOracle()
    n.set_centroid_mode(1)
    n.set_target(tgt, 1.0)
    rn = n.align(src, out["in_guess"], orc.NdtParams.reference())
    out["ndt_reference_T"] = np.asarray(rn.T, np.float64)
    out["ndt_reference_meta"] = np.array([[float(rn.converged), float(rn.iterations), float(rn.trans_probability)]])
    # (5)-(7) the scheme classes over the oracle: merged clouds, edge features, by-product files
    s = schemes.IncrementalICP(backend=OracleBackend())
    out["incremental_icp_merged"] = np.ascontiguousarray(s.registration([c.copy() for c in chain]).points)
    idx = orc.edge_features(np.ascontiguousarray(chain[0].points), chain[0].width, chain[0].height)
    out["edge_features_chain0"] = np.ascontiguousarray(chain[0].points[idx])
    by = os.path.join(str(tmp_dir), "standin_dataset")
    os.makedirs(by, exist_ok=True)
    rads = float(out["in_rads"][0])
    e = schemes.ICPEdgeBasedRegistration(rads=rads, backend=OracleBackend())
    e.write_byproducts, e.byproduct_dir = True, by
    out["icp_edge_merged"] = np.ascontiguousarray(e.registration([c.copy() for c in chain]).points)
    for k in range(4):
        out["icp_edge_byproduct_edge%d" % k] = rs.load_pcd(os.path.join(by, "edge-%d.pcd" % k)).points
    out["icp_edge_byproduct_edge_cloud"] = rs.load_pcd(os.path.join(by, "edge_cloud.pcd")).points
    d = schemes.NDTEdgeBasedRegistration(rads=rads, backend=OracleBackend())
    out["ndt_edge_merged"] = np.ascontiguousarray(d.registration([c.copy() for c in chain]).points)
    # (8)-(9) reciprocal correspondences, the trimmed rejector
    for key, kw in (("reciprocal", dict(use_reciprocal=1)), ("trimmed", dict(trim_overlap_ratio=0.8))):
        q = orc.IcpParams.reference()
        q.accum_mode = 0
        for a, v in kw.items():
            setattr(q, a, v)
        o.begin(None, q)
        ki, kd = o.search()
        out["corr_%s_it0" % key] = _corr_matrix(ki, kd)
        out["icp_%s_T" % key] = np.asarray(o.align(None, q).T, np.float64)
    # (10) a binary_compressed file of the source frame (written by this repository's own LZF coder here)
    f = os.path.join(str(tmp_dir), "standin_pair1_binary_compressed.pcd")
    rs.save_pcd(f, pair[1], binary=True, compressed=True)
    out["pair1_binary_compressed"] = rs.load_pcd(f).points
    out["pair1_binary_compressed_bytes"] = np.frombuffer(open(f, "rb").read(), np.uint8)
    np.savez_compressed(path, **out)
    return sorted(out)
