"""pgdvs_amd -- MI355X-native implementation of the PGDVS per-target-view rendering
inner loop behind the reference's renderer plugin API.

Module layout mirrors the reference so that Hydra ``_target_`` strings translate by
prefix (``pgdvs.`` -> ``pgdvs_amd.``):

  pgdvs_amd.renderers.pgdvs_renderer.PGDVSRenderer          <- pgdvs/renderers/pgdvs_renderer.py
  pgdvs_amd.renderers.pgdvs_renderer_dyn.PGDVSDynamicRenderer<- pgdvs/renderers/pgdvs_renderer_dyn.py
  pgdvs_amd.renderers.st_geo_renderer.StaticGeoPointRenderer <- pgdvs/renderers/st_geo_renderer.py
  pgdvs_amd.models.gnt.renderer.BaseRenderer                 <- pgdvs/models/gnt/renderer.py
  pgdvs_amd.utils.softsplat.softsplat                        <- pgdvs/utils/softsplat.py
  pgdvs_amd.datasets.static_aggregation.aggregate_static_pcl <- pgdvs/datasets/nvidia_eval_pure_geo.py:183-277

All compute runs in hand-written HIP kernels (ml-pgdvs_amd/csrc) reached through the
C ABI in include/pgdvs_hip.h; there is no CPU fallback.
"""
__version__ = "0.1.0"
