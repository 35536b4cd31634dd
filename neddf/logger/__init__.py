from neddf_amd.logger import ScalarLog  # noqa: F401
