from neddf_amd.dataset import BaseDataset, LLFFDataset, NeRFSyntheticDataset  # noqa: F401
