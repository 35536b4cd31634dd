from neddf_amd.trainer import BaseTrainer, NeRFTrainer  # noqa: F401
