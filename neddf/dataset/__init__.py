from neddf_amd.dataset import BaseDataset, NeRFSyntheticDataset  # noqa: F401
