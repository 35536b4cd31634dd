from neddf_amd.ray import Ray, Sampling  # noqa: F401
