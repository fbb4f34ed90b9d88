"""CPU oracle — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  It is a CPU restatement of the reference's PCL-backed pair-registration path
(see rsreg_oracle.h: "PARITY UNPINNED").  The product never routes through it.
"""
from .oracle import (  # noqa: F401
    IcpOracle,
    NdtOracle,
    IcpParams,
    NdtParams,
    approx_voxel_grid,
    build,
    edge_features,
    lib,
    mat4_mul,
    transform_cloud,
    umeyama_from_sums,
)
