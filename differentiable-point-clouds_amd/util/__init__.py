from . import camera, config, drc, gauss_kernel, point_cloud, quaternion  # noqa: F401
