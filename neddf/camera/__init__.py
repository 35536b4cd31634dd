from neddf_amd.camera import BaseCameraCalib, Camera, PinholeCalib  # noqa: F401
