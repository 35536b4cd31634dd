from neddf_amd.nn_module import PositionalEncoding, tanhExp  # noqa: F401
