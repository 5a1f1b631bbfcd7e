"""Mirror of ``pgdvs.models.gnt.model.GNTModel`` (pgdvs/models/gnt/model.py:12-101): the
coarse GNT aggregation network + the ResUNet feature extractor, with the reference's
checkpoint layout (`net_coarse`, `feature_net`, `net_fine` state dicts)."""
import logging

import torch

from .models.feature_network import ResUNet
from .models.transformer_network import GNT

LOGGER = logging.getLogger(__name__)


class GNTModel(torch.nn.Module):
    def __init__(self, *, netwidth=64, transformer_depth=8, coarse_feat_dim=32, fine_feat_dim=32, single_net=True,
                 posenc_max_freq_log2=9, pos_enc_n_freqs=10, view_enc_n_freqs=10, ckpt_path=None, _target_=None):
        super().__init__()
        mk = lambda ch: GNT(netwidth=netwidth, transformer_depth=transformer_depth, in_feat_ch=ch,
                            posenc_max_freq_log2=posenc_max_freq_log2, pos_enc_n_freqs=pos_enc_n_freqs,
                            view_enc_n_freqs=view_enc_n_freqs, ret_alpha=True)
        self.net_coarse = mk(coarse_feat_dim)
        self.single_net = single_net
        self.net_fine = None if single_net else mk(fine_feat_dim)
        self.feature_net = ResUNet(coarse_out_ch=coarse_feat_dim, fine_out_ch=fine_feat_dim, single_net=single_net)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path)

    def init_from_ckpt(self, path, ignore_keys=()):
        """GNT release checkpoints hold one state dict per sub-network (:64-101)."""
        ckpt = torch.load(path, map_location="cpu")
        state = {}
        for name in ("net_coarse", "feature_net", "net_fine"):
            if name in ckpt:
                for k, v in ckpt[name].items():
                    if not any(k.startswith(ik) for ik in ignore_keys):
                        state[f"{name}.{k}"] = v
        missing, unexpected = self.load_state_dict(state, strict=False)
        LOGGER.info("GNT checkpoint %s: %d missing, %d unexpected keys", path, len(missing), len(unexpected))
