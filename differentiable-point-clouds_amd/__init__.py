"""MI355X-native differentiable point-cloud projector (drop-in for the hot
path of eldar/differentiable-point-clouds: dpc/util/{point_cloud,drc,
gauss_kernel,quaternion,camera}.py).

The directory name is not a Python identifier; import it through the
``dpc_amd`` alias module at the repo root (``import dpc_amd``) or with
``importlib.import_module("differentiable-point-clouds_amd")``.
"""
from . import _capi, _ext, distributed, graphs, model_pc, ops, synthetic, util  # noqa: F401
from ._capi import DpcError, get_library  # noqa: F401
from .util.config import Config, default_config  # noqa: F401
from .util.drc import drc_depth_projection, drc_event_probabilities, drc_projection  # noqa: F401
from .util.gauss_kernel import gauss_kernel_1d, smoothing_kernel  # noqa: F401
from .util.point_cloud import (pc_perspective_transform, pc_point_dropout, pointcloud2voxels3d_fast,  # noqa: F401
                               pointcloud_project_fast, smoothen_voxels3d)

__version__ = "0.2.0"
