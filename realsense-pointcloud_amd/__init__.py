"""MI355X-native pairwise point-cloud registration engine (ICP / NDT pair-registration hot
path of hyunminch/realsense-pointcloud).  The directory name carries a hyphen, so import it
with ``importlib.import_module("realsense-pointcloud_amd")`` or through the ``rsreg_amd``
alias module at the repo root."""
from . import cloud, synth  # noqa: F401
from .cloud import POINT_DTYPE, PointCloud, load_pcd, save_pcd  # noqa: F401
