"""Alias package: the reference's import paths (`neddf.render.NeRFRender`,
`neddf.network.NeDDF`, ... -- the `_target_` strings of its Hydra configs)
resolve to the MI355X implementation in neddf_amd."""
