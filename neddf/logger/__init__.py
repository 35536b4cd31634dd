from neddf_amd.logger import BaseLogger, NeRFTBLogger  # noqa: F401
