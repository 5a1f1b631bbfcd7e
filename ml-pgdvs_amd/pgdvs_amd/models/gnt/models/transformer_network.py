"""GNT ray-feature aggregation network (rows A14 of SURVEY.md 8a): parameters laid out and
named like ``pgdvs.models.gnt.models.transformer_network.GNT``
(pgdvs/models/gnt/models/transformer_network.py:341-539) so release checkpoints
(`net_coarse.*`) load unchanged; the forward pass is a dense, mask-driven formulation (no
per-group Python loops, no host syncs) that dispatches the view-transformer contraction to
the MFMA kernels in csrc/gnt_view.hip.

Reference behaviours kept: ``v = v_fc(k_fc(feat))`` (:84-85); (ray,sample) rows without any
valid source view have their mask *removed* (:124-129) -- the "uniform attention" branch
(:159-163) is therefore dead code upstream and is not reproduced; LayerNorm eps 1e-6 inside
the transformers and 1e-5 for the final norm; sample weights = row of query sample 0 of the
last ray-attention, averaged over heads (:336)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

TINY_NUMBER = 1e-6  # pgdvs/models/gnt/common.py


def _mlp(cin, chid, cout):
    return nn.Sequential(nn.Linear(cin, chid), nn.ReLU(), nn.Linear(chid, cout))


class FeedForward(nn.Module):
    def __init__(self, dim, hid_dim, dp_rate=0.0):
        super().__init__()
        self.fc1 = nn.Linear(dim, hid_dim)
        self.fc2 = nn.Linear(hid_dim, dim)

    def forward(self, x):
        return self.fc2(F.relu(self.fc1(x)))


class Attention2D(nn.Module):
    """parameters of the subtraction-based view attention (:59-76)"""

    def __init__(self, dim, dp_rate=0.0):
        super().__init__()
        self.q_fc = nn.Linear(dim, dim, bias=False)
        self.k_fc = nn.Linear(dim, dim, bias=False)
        self.v_fc = nn.Linear(dim, dim, bias=False)
        self.pos_fc = _mlp(4, dim // 8, dim)
        self.attn_fc = _mlp(dim, dim // 8, dim)
        self.out_fc = nn.Linear(dim, dim)


class Transformer2D(nn.Module):
    def __init__(self, dim, ff_hid_dim, ff_dp_rate=0.0, attn_dp_rate=0.0):
        super().__init__()
        self.attn_norm = nn.LayerNorm(dim, eps=1e-6)
        self.ff_norm = nn.LayerNorm(dim, eps=1e-6)
        self.ff = FeedForward(dim, ff_hid_dim)
        self.attn = Attention2D(dim)


class Attention(nn.Module):
    def __init__(self, dim, n_heads, dp_rate=0.0):
        super().__init__()
        self.q_fc = nn.Linear(dim, dim, bias=False)
        self.k_fc = nn.Linear(dim, dim, bias=False)
        self.v_fc = nn.Linear(dim, dim, bias=False)
        self.out_fc = nn.Linear(dim, dim)
        self.n_heads = n_heads


class Transformer(nn.Module):
    def __init__(self, dim, ff_hid_dim, ff_dp_rate=0.0, n_heads=4, attn_dp_rate=0.0):
        super().__init__()
        self.attn_norm = nn.LayerNorm(dim, eps=1e-6)
        self.ff_norm = nn.LayerNorm(dim, eps=1e-6)
        self.ff = FeedForward(dim, ff_hid_dim)
        self.attn = Attention(dim, n_heads)


def _posenc(x, n_freqs, max_log2):
    """[x, sin(x f0), cos(x f0), sin(x f1), ...] (gnt/common.py Embedder order), all frequencies in one pass"""
    freqs = 2.0 ** torch.linspace(0.0, max_log2, steps=n_freqs, device=x.device)
    xf = x[..., None, :] * freqs[:, None]  # [..., F, 3]
    return torch.cat((x, torch.cat((torch.sin(xf), torch.cos(xf)), -1).flatten(-2)), -1)


class GNT(nn.Module):
    def __init__(self, *, netwidth, transformer_depth, in_feat_ch=32, posenc_max_freq_log2=9, pos_enc_n_freqs=10,
                 view_enc_n_freqs=10, ret_alpha=True):
        super().__init__()
        self.max_log2, self.pos_freqs, self.view_freqs = posenc_max_freq_log2, pos_enc_n_freqs, view_enc_n_freqs
        self.posenc_dim = 3 + 3 * 2 * pos_enc_n_freqs
        self.viewenc_dim = 3 + 3 * 2 * view_enc_n_freqs
        self.ret_alpha = ret_alpha
        self.hidden_hook = None  # tests: called as hook("view" / "ray", q) behind every transformer block
        self.norm = nn.LayerNorm(netwidth)
        self.rgb_fc = nn.Linear(netwidth, 3)
        self.rgbfeat_fc = _mlp(in_feat_ch + 3, netwidth, netwidth)
        self.view_selftrans = nn.ModuleList([])
        self.view_crosstrans = nn.ModuleList([])
        self.q_fcs = nn.ModuleList([])
        for i in range(transformer_depth):
            self.view_crosstrans.append(Transformer2D(netwidth, 4 * netwidth))
            self.view_selftrans.append(Transformer(netwidth, 4 * netwidth, n_heads=4))
            self.q_fcs.append(_mlp(netwidth + self.posenc_dim + self.viewenc_dim, netwidth, netwidth) if i % 2 == 0
                              else nn.Identity())

    # -- view transformer layer (Transformer2D + Attention2D, :78-169,:197-223) -------------
    def _view_layer(self, layer, q, feat, ray_diff, valid, cnt, want_stats):
        """q[R,S,D], feat[R,S,V,D], ray_diff[R,S,V,4], valid[R,S,V] bool (mask removed for empty
        rows), cnt[R,S] number of valid views."""
        from .... import ops

        if q.is_cuda and not ops.needs_autograd(layer, q, feat):
            if ops.gnt_view_available(q.shape[-1], feat.shape[2]):
                return ops.gnt_view_layer(layer, q, feat, ray_diff, valid, want_stats)
            ops.gnt_fallback("pgdvs_gnt_view_layer", f"width {q.shape[-1]} with {feat.shape[2]} source views (needs 64 and <= 64)")
        a = layer.attn
        x = layer.attn_norm(q)
        qq = a.q_fc(x)
        k = a.k_fc(feat)
        v = a.v_fc(k)
        pos = a.pos_fc(ray_diff)
        att = a.attn_fc(k - qq[:, :, None, :] + pos)
        att = att.masked_fill(~valid[..., None], -float("inf"))
        att = torch.softmax(att, dim=2)
        x = a.out_fc(((v + pos) * att).sum(dim=2)) + q
        x = layer.ff(layer.ff_norm(x)) + x
        stats = None
        if want_stats:
            w = valid[..., None].float()
            n = cnt[..., None].float()
            mean = (k * w).sum(2) / n
            var = (((k - mean[:, :, None]) ** 2) * w).sum(2) / (n - 1).clamp(min=1)
            k_std = torch.where(n > 1, var.sqrt(), torch.zeros_like(var))
            k_std_n = torch.where(n > 1, k_std / ((k.abs() * w).sum(2) / n + TINY_NUMBER), torch.zeros_like(var))
            ent = (-att * torch.log(att + 1e-8)).sum(2).mean(-1)
            stats = (ent, k_std.mean(-1), k_std_n.mean(-1))
        return x, stats

    @staticmethod
    def _ray_layer(layer, q, want_attn):
        from .... import ops

        a = layer.attn
        R, S, D = q.shape
        if q.is_cuda and not ops.needs_autograd(layer, q):
            if ops.gnt_ray_available(D, S, a.n_heads):
                return ops.gnt_ray_layer(layer, q, want_attn)
            ops.gnt_fallback("pgdvs_gnt_ray_layer", f"width {D}, {a.n_heads} heads, {S} samples per ray "
                                                    f"(needs 64, 4 and <= {ops.GNT_RAY_MAX_SAMPLES})")
        hd = D // a.n_heads
        x = layer.attn_norm(q)
        sp = lambda t: t.view(R, S, a.n_heads, hd).permute(0, 2, 1, 3)
        qh, kh, vh = sp(a.q_fc(x)), sp(a.k_fc(x)), sp(a.v_fc(x))
        att = torch.softmax(torch.matmul(qh, kh.transpose(-2, -1)) / math.sqrt(hd), dim=-1)
        out = torch.matmul(att, vh).permute(0, 2, 1, 3).reshape(R, S, D)
        x = a.out_fc(out) + q
        x = layer.ff(layer.ff_norm(x)) + x
        return x, (att.mean(dim=1)[:, 0] if want_attn else None)

    def forward(self, rgb_feat, ray_diff, mask, pts, ray_d, ret_view_entropy=False, ret_view_std=False):
        """Same contract as the reference (:423-539): returns (cat[rgb, weights], extras)."""
        viewdirs = ray_d / torch.norm(ray_d, dim=-1, keepdim=True)
        input_views = _posenc(viewdirs.float(), self.view_freqs, self.max_log2)[:, None].expand(pts.shape[0], pts.shape[1], -1)
        input_pts = _posenc(pts.float(), self.pos_freqs, self.max_log2)
        from .... import ops

        hip = rgb_feat.is_cuda and not ops.needs_autograd(self, rgb_feat)
        fused = hip and ops.gnt_embed_available(self.rgbfeat_fc, rgb_feat.shape[-1])
        if hip and not fused:
            ops.gnt_fallback("pgdvs_gnt_embed", f"{rgb_feat.shape[-1]} input channels / this rgbfeat_fc (needs 33..36 -> 64 -> 64)")
        if fused:
            feat, q, std0 = ops.gnt_embed(self.rgbfeat_fc, rgb_feat, ret_view_std)
        else:
            feat = self.rgbfeat_fc(rgb_feat)
            q = feat.max(dim=2)[0]
        V = feat.shape[2]
        valid = mask[..., 0] != 0
        cnt = valid.sum(-1)
        empty = cnt == 0
        valid = valid | empty[..., None]  # mask removed where no view is valid (:124-129)
        cnt = torch.where(empty, torch.full_like(cnt, V), cnt)
        want_stats = ret_view_entropy or ret_view_std
        ents, stds, stdns = [], [], []
        if ret_view_std and fused:
            stds.append(std0[0])
            stdns.append(std0[1])
        elif ret_view_std:
            s0 = torch.std(feat, dim=2)
            stds.append(s0.mean(-1))
            stdns.append((s0 / (feat.abs().mean(2) + TINY_NUMBER)).mean(-1))
        attn = None
        posfc = None
        if hip and ops.gnt_posfc_available(self.q_fcs, q.shape[-1]):
            posfc = ops.GntPosFc(self.q_fcs, input_pts, input_views[:, 0])
        elif hip:
            ops.gnt_fallback("pgdvs_gnt_posfc", f"width {q.shape[-1]} (needs 64)")
        for i, (vl, qfc, rl) in enumerate(zip(self.view_crosstrans, self.q_fcs, self.view_selftrans)):
            q, stats = self._view_layer(vl, q, feat, ray_diff, valid, cnt, want_stats)
            if self.hidden_hook is not None:
                self.hidden_hook("view", q)
            if i % 2 == 0:
                q = posfc(i, q) if posfc is not None else qfc(torch.cat((q, input_pts, input_views), dim=-1))
            q, attn = self._ray_layer(rl, q, self.ret_alpha)
            if self.hidden_hook is not None:
                self.hidden_hook("ray", q)
            if want_stats:
                ents.append(stats[0])
                stds.append(stats[1])
                stdns.append(stats[2])
        extras = {}
        if ret_view_entropy:
            extras["view_entropy"] = torch.stack(ents, dim=2)
        if ret_view_std:
            extras["view_std"] = torch.stack(stds, dim=2)
            extras["view_std_normalized"] = torch.stack(stdns, dim=2)
        if hip and ops.gnt_head_available(self.norm, self.rgb_fc):
            outputs = ops.gnt_head(self.norm, self.rgb_fc, q)
        else:
            if hip:
                ops.gnt_fallback("pgdvs_gnt_head", f"width {q.shape[-1]} (needs 64, eps 1e-5)")
            outputs = self.rgb_fc(self.norm(q).mean(dim=1))
        if self.ret_alpha:
            return torch.cat([outputs, attn], dim=1), extras
        return outputs, extras
