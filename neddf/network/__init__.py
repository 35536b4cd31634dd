from neddf_amd.network import BaseNeuralField, NeDDF, NeDDFField, NeRF, NeRFField  # noqa: F401
