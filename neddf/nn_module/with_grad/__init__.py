from neddf_amd.nn_module import (LeakyReLUGradFunction, LinearGradFunction, LinearGradLayer,  # noqa: F401
                                 PositionalEncodingGradLayer, ReLUGradFunction, SigmoidGradFunction,
                                 SoftplusGradFunction, TanhExpGradFunction)
