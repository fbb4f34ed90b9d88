"""Keeps the only route to "parity: green" alive: the bodies of tests/test_pcl_pin.py -- which skip until somebody runs
oracle/pcl_harness where PCL exists -- run here against an ORACLE-generated stand-in of the same schema (tests/pcl_pin_standin.py,
written to a temporary directory).  THIS PINS NOTHING against PCL and says so; what it proves is that the fixture's schema, the
harness's output names, the converter and the four test bodies still agree with today's ABI, oracle and host layers."""
import os
import re

import numpy as np
import pytest

import pcl_pin_standin
import test_pcl_pin as real

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def standin(orc, rs, tmp_path_factory):
    d = tmp_path_factory.mktemp("pcl_pin_standin")
    path = str(d / "pcl_pin_STANDIN_not_pcl.npz")
    keys = pcl_pin_standin.make_standin(path, orc, rs, d)
    pin = np.load(path)
    assert sorted(pin.files) == keys and str(pin["standin_source"]).startswith("ORACLE-GENERATED STAND-IN")
    return pin


def test_standin_has_the_schema_the_harness_writes(standin):
    """Every output the PCL harness writes (oracle/pcl_harness/pcl_pin.cpp) and every input the converter folds in
    (to_golden.py) is a key of the stand-in, and every key a test body reads exists: harness, converter, fixture and tests
    speak of the same names."""
    cpp = open(os.path.join(ROOT, "oracle", "pcl_harness", "pcl_pin.cpp")).read()
    names = set(re.findall(r'out \+ "([A-Za-z0-9_]+)\.(?:txt|pcd)"', cpp))
    names |= {"voxel_1cm", "voxel_default"}                                   # out + (which == 0 ? ... : ...)
    names |= {"icp_gate5cm_%dit_T" % i for i in (1, 5, 30)}                   # "icp_gate5cm_" + to_string(iters) + "it_T.txt"
    names |= {"icp_edge_byproduct_edge%d" % k for k in range(4)}              # "icp_edge_byproduct_edge" + to_string(k) + ".pcd"
    assert {"corr_it0", "icp_reference_T", "ndt_reference_T", "edge_features_chain0", "icp_edge_merged", "ndt_edge_merged", "corr_trimmed_it0",
            "icp_reciprocal_T", "pair1_binary_compressed", "incremental_icp_merged"} <= names
    assert "for (int iters : {1, 5, 30})" in cpp
    conv = open(os.path.join(ROOT, "oracle", "pcl_harness", "to_golden.py")).read()
    assert '"in_" + name.replace("-", "")' in conv and '_shape"' in conv and '"in_guess"' in conv and '"in_rads"' in conv and 'key + "_bytes"' in conv
    names |= {"in_pair0", "in_pair1", "in_guess", "in_rads", "pair1_binary_compressed_bytes"}
    names |= {"in_chain%d" % k for k in range(4)} | {"in_chain%d_shape" % k for k in range(4)}
    have = set(standin.files) - {"standin_source"}
    assert names == have, (sorted(names - have), sorted(have - names))
    # the keys the four bodies read
    body = open(real.__file__).read()
    for key in re.findall(r'pin\["([A-Za-z0-9_]+)"\]', body):
        assert key in have, key
    for fmt in re.findall(r'pin\["([A-Za-z0-9_%]+)" % ', body):
        assert any(re.fullmatch(fmt.replace("%d", r"\d+").replace("%s", r"[a-z]+"), k) for k in have), fmt


def test_oracle_bodies_run_against_the_standin(standin, orc, rs):
    real.check_oracle_matches_pcl(standin, orc)
    real.check_oracle_round2_components_match_pcl(standin, orc, rs)


@pytest.mark.gpu
def test_engine_bodies_run_against_the_standin(standin, rs, tmp_path):
    from rsreg_amd import api, lib
    lib.build()
    if api.device_count() < 1:
        pytest.fail("no HIP device: the product has no CPU fallback")
    real.check_engine_matches_pcl(standin, rs)
    real.check_engine_round2_components_match_pcl(standin, rs, tmp_path)


def test_the_harness_speaks_todays_interfaces():
    """oracle/pcl_harness/{make_inputs,to_golden}.py import what this repository still has, and the harness's C++ names only
    reference headers and PCL classes (nothing of include/rsreg.h: it must not depend on the ABI it is there to pin)."""
    import ast
    for name in ("make_inputs.py", "to_golden.py"):
        src = open(os.path.join(ROOT, "oracle", "pcl_harness", name)).read()
        ast.parse(src)
        for attr in re.findall(r"rsreg_amd\.([a-z_]+)\(", src):
            import rsreg_amd
            assert hasattr(rsreg_amd, attr), (name, attr)
        for attr in re.findall(r"synth\.([a-z_]+)\(", src):
            from rsreg_amd import synth
            assert hasattr(synth, attr), (name, attr)
    cpp = open(os.path.join(ROOT, "oracle", "pcl_harness", "pcl_pin.cpp")).read()
    code = "\n".join(l.split("//")[0] for l in cpp.splitlines())       # (comments may name the engine's counterparts)
    assert "rsreg" not in code
    for header in ("incremental_icp.hpp", "icp_edge_based_registration.hpp", "ndt_edge_based_registration.hpp", "edge_extractor.hpp"):
        assert header in cpp, header
