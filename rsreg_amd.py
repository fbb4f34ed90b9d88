"""Importable alias for the package directory ``realsense-pointcloud_amd`` (hyphenated)."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module("realsense-pointcloud_amd")
sys.modules[__name__] = _pkg
