#!/usr/bin/env python3
"""gnt_depth8.npz: the reference's own GNT (pgdvs/models/gnt/models/transformer_network.py:341-539) at the depth the
reference runs it -- netwidth 64, transformer_depth 8 (configs/static_renderer/gnt.yaml:9), 256 samples per ray
(configs/engine/evaluator_pgdvs.yaml:14-16) -- with V = 10 source views (configs/_basic.yaml:45; 16 rays) and V = 24
(BASELINE.json configs[2]; 8 rays), masks that leave rays with no / one / all valid views, view entropy and std on.

Stored: the reference's outputs (rgb + sample weights, extras) and, from forward hooks on its sixteen transformer blocks, a
subsample q[:, ::16, ::4] of the hidden state behind every block (so per-layer drift can be compared with the reference
itself).  NOT stored: inputs and weights -- both come from the seeded numpy generators in gnt_depth8_inputs.py, whose
checksums ride along.  Runs in the build container only (imports /root/reference)."""
import pathlib
import sys

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import make_golden as MG  # noqa: E402
import gnt_depth8_inputs as GI  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent


def main():
    MG._install_stubs()
    from pgdvs.models.gnt.models.transformer_network import GNT

    torch.manual_seed(0)
    net = GNT(netwidth=64, transformer_depth=8, in_feat_ch=32, posenc_max_freq_log2=9, pos_enc_n_freqs=10,
              view_enc_n_freqs=10, ret_alpha=True).eval()
    sd = net.state_dict()
    w = GI.make_weights({k: tuple(v.shape) for k, v in sd.items()})
    net.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    out = {"weights_checksum": GI.checksum(w), "n_params": sum(v.size for v in w.values()),
           "state_dict_keys": np.array(list(w.keys()))}

    hidden = []
    hooks = []
    for i in range(8):
        for mod in (net.view_crosstrans[i], net.view_selftrans[i]):
            hooks.append(mod.register_forward_hook(lambda m, a, o: hidden.append(o[0].detach().clone())))
    T = torch.from_numpy
    for case in GI.CASES:
        x = GI.make_inputs(case)
        hidden.clear()
        with torch.no_grad():
            o, ex = net(T(x["rgb_feat"]), T(x["ray_diff"]), T(x["mask"]), T(x["pts"]), T(x["ray_d"]),
                        ret_view_entropy=True, ret_view_std=True)
        assert len(hidden) == 16
        out[f"{case}_inputs_checksum"] = GI.checksum(x)
        out[f"{case}_out"] = o.numpy()
        for k, v in ex.items():
            out[f"{case}_{k}"] = v.numpy()
        out[f"{case}_hidden"] = np.stack([h[:, ::16, ::4].numpy() for h in hidden])  # [16 blocks, R, 16, 16]
        print(case, "out", tuple(o.shape), "rgb range", float(o[:, :3].min()), float(o[:, :3].max()),
              "max weight", float(o[:, 3:].max()))
    for h in hooks:
        h.remove()
    np.savez_compressed(OUT / "gnt_depth8.npz", **out)
    print(f"  gnt_depth8.npz {(OUT / 'gnt_depth8.npz').stat().st_size / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
