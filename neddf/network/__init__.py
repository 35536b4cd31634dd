from neddf_amd.network import BaseNeuralField, NeDDF, NeDDFField, NeRF, NeRFField, NeuS  # noqa: F401
