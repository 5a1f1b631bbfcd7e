"""Mirror of ``pgdvs.models.gnt.renderer.BaseRenderer`` (pgdvs/models/gnt/renderer.py:21-177).

Round-1 state: the class exists so that ``static_renderer=gnt`` configurations construct
and so that ``PGDVSRenderer`` can bind ``.projector.compute_projections``
(pgdvs_renderer.py:78); the reference's own ``data["rgb_gnt"]`` hook
(pgdvs_renderer.py:120-122) supplies the static image.  Running the GNT network itself
(ResUNet features + view/ray transformer aggregation, rows A13-A16 of SURVEY.md 8a) is
not built yet and raises.
"""
import torch

from .projector import Projector


class BaseRenderer(torch.nn.Module):
    def __init__(self, model_cfg=None):
        super().__init__()
        self.model_cfg = model_cfg
        self.projector = Projector()

    def forward(self, *args, **kwargs):
        raise NotImplementedError(
            "GNT ray-feature aggregation (A13-A16) is not built yet; pass the static image "
            "through data['rgb_gnt'] (pgdvs_renderer.py:120-122) or use static_renderer=geo")
