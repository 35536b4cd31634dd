"""neddf_amd -- MI355X-native volumetric renderer for Neural Density-Distance
Fields behind the reference project's plugin surface (`neddf.render`,
`neddf.ray`, `neddf.network`, `neddf.camera`).  The compute path is
libneddf_hip.so (hand-written gfx950 HIP kernels, C ABI in include/neddf_hip.h);
this package is the thin host side.  Import is GPU-free; the library is loaded
on first use and there is no CPU fallback."""
from . import camera, config, dataset, logger, loss, metrics, network, nn_module, ray, render, trainer  # noqa: F401
from ._lib import Context, NeddfError, load  # noqa: F401
from .camera import Camera, PinholeCalib  # noqa: F401
from .network import NeDDF, NeDDFField, NeRF, NeRFField, NeuS  # noqa: F401
from .ray import Ray, Sampling  # noqa: F401
from .render import NeRFRender, RenderTarget  # noqa: F401

__version__ = "0.1.0"
