from neddf_amd.render import BaseNeuralRender, NeRFRender, RenderTarget  # noqa: F401
