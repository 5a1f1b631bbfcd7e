"""CPU ORACLE (test infrastructure) for the GNT rows A13-A15 of SURVEY.md 8a: epipolar
projection + bilinear gathering across source views, the view/ray transformer aggregation
and the per-ray reductions.  numpy fp32 restatement of
  pgdvs/models/gnt/projector.py:14-115,117-308   (Projector)
  pgdvs/models/gnt/ray_sampler.py:59-123          (sample_z_vals / sample_along_camera_ray)
  pgdvs/models/gnt/models/transformer_network.py:10-55,59-169,197-223,231-338,341-539
  pgdvs/models/gnt/renderer.py:207-300            (render_rays reductions)
Pinned against tests/golden/gnt_small.npz (outputs of the reference modules themselves).
Only tests/, smoke() and bench.py's cpu_baseline may import this module.
"""
from __future__ import annotations

import numpy as np

from . import oracle as orc

TINY_NUMBER = np.float32(1e-6)  # pgdvs/models/gnt/common.py
f32 = np.float32


# ---------------------------------------------------------------- sampling (A15)
def sample_along_camera_ray(ray_o, ray_d, depth_range, n_samples, inv_uniform=True):
    """deterministic branch of ray_sampler.py:76-123; depth_range[#ray,2]."""
    near, far = depth_range[:, 0].astype(f32), depth_range[:, 1].astype(f32)
    assert np.all(near > 0) and np.all(far > near)
    if inv_uniform:
        start = f32(1.0) / near
        step = (f32(1.0) / far - start) / f32(n_samples - 1)
        inv = np.stack([start + f32(i) * step for i in range(n_samples)], 1)
        z = f32(1.0) / inv
    else:
        step = (far - near) / f32(n_samples - 1)
        z = np.stack([near + f32(i) * step for i in range(n_samples)], 1)
    pts = z[:, :, None] * ray_d[:, None, :] + ray_o[:, None, :]
    return pts.astype(f32), z.astype(f32)


# ---------------------------------------------------------------- projector (A13)
def _grid_sample_ac(img_chw, px, py):
    """F.grid_sample(bilinear, zeros, align_corners=True) at pixel coords (px,py) of THIS map."""
    C, H, W = img_chw.shape
    x0 = np.floor(px)
    y0 = np.floor(py)
    out = np.zeros((C,) + px.shape, f32)
    fin = np.isfinite(px) & np.isfinite(py)
    for dx in (0, 1):
        for dy in (0, 1):
            xi = x0 + dx
            yi = y0 + dy
            wx = (px - x0) if dx else (x0 + 1 - px)
            wy = (py - y0) if dy else (y0 + 1 - py)
            ok = fin & (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
            xi_c = np.clip(np.where(ok, xi, 0), 0, W - 1).astype(np.int64)
            yi_c = np.clip(np.where(ok, yi, 0), 0, H - 1).astype(np.int64)
            w = np.where(ok, wx * wy, 0).astype(f32)
            out += img_chw[:, yi_c, xi_c] * w[None]
    return out


def projector_compute(pts, cam_tgt, src_rgbs, cams_src, featmaps, inv_masks=None):
    """Projector.compute (projector.py:117-308).  pts[R,S,3]; src_rgbs[V,H,W,3];
    cams_src[V,34]; featmaps[V,C,hf,wf]; inv_masks[V,H,W,1] or None.
    -> rgb_feat[R,S,V,3+C], ray_diff[R,S,V,4], mask_inbound[R,S,V,1], mask[R,S,V,1], mask_invalid"""
    R, S, _ = pts.shape
    V, H, W, _ = src_rgbs.shape
    flat = pts.reshape(-1, 3).astype(f32)
    h, w = f32(cams_src[0][0]), f32(cams_src[0][1])
    rgb_feat, ray_diff, m_in, m_inv = [], [], [], []
    q_pos = cam_tgt[18:34].reshape(4, 4)[:3, 3].astype(f32)
    for v in range(V):
        blk = orc.cam_prep(cams_src[v])
        P = blk[21:37].reshape(4, 4)
        p = flat @ P[:, :3].T + P[:, 3]  # [N,4]
        z = np.maximum(p[:, 2:3], f32(1e-8))
        pix = np.clip(p[:, :2] / z, -1e6, 1e6).astype(f32)
        in_front = p[:, 2] > 0
        # normalize with (w-1, h-1) then grid_sample(align_corners=True) on each map (:29-39,:251-268)
        gx = f32(2) * pix[:, 0] / (w - 1) - 1
        gy = f32(2) * pix[:, 1] / (h - 1) - 1
        px_img, py_img = (gx + 1) / 2 * (W - 1), (gy + 1) / 2 * (H - 1)
        hf, wf = featmaps.shape[2:]
        px_f, py_f = (gx + 1) / 2 * (wf - 1), (gy + 1) / 2 * (hf - 1)
        rgb = _grid_sample_ac(np.ascontiguousarray(src_rgbs[v].transpose(2, 0, 1)), px_img, py_img)
        feat = _grid_sample_ac(featmaps[v], px_f, py_f)
        rgb_feat.append(np.concatenate([rgb, feat], 0).T)  # [N,3+C]
        inb = (pix[:, 0] <= w - 1) & (pix[:, 0] >= 0) & (pix[:, 1] <= h - 1) & (pix[:, 1] >= 0)
        m_in.append((inb & in_front).astype(f32))
        if inv_masks is not None:
            mv = _grid_sample_ac(np.ascontiguousarray(inv_masks[v].transpose(2, 0, 1)), px_img, py_img)[0]
            m_inv.append((mv > 1e-3).astype(f32))
        # compute_angle (:75-115)
        t_pos = cams_src[v][18:34].reshape(4, 4)[:3, 3].astype(f32)
        a = q_pos[None] - flat
        b = t_pos[None] - flat
        a = a / (np.linalg.norm(a, axis=-1, keepdims=True) + f32(1e-6))
        b = b / (np.linalg.norm(b, axis=-1, keepdims=True) + f32(1e-6))
        d = a - b
        dn = np.linalg.norm(d, axis=-1, keepdims=True)
        dot = np.sum(a * b, -1, keepdims=True)
        ray_diff.append(np.concatenate([d / np.maximum(dn, f32(1e-6)), dot], -1))
    st = lambda xs, c: np.stack(xs, 1).reshape(R, S, V, c).astype(f32)
    out = {"rgb_feat": st(rgb_feat, rgb_feat[0].shape[-1]), "ray_diff": st(ray_diff, 4),
           "mask_inbound": st([m[:, None] for m in m_in], 1)}
    if inv_masks is not None:
        out["mask_invalid"] = st([m[:, None] for m in m_inv], 1)
        out["mask"] = out["mask_inbound"] * (1 - out["mask_invalid"])
    else:
        out["mask_invalid"] = np.zeros_like(out["mask_inbound"])
        out["mask"] = out["mask_inbound"]
    return out


# ---------------------------------------------------------------- transformer (A14)
def _lin(x, W, b=None):
    y = x @ W.T
    return y + b if b is not None else y


def _ln(x, g, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return ((x - mu) / np.sqrt(var + f32(eps)) * g + b).astype(f32)


def _embed(x, n_freqs=10, max_log2=9):
    freqs = (2.0 ** np.linspace(0.0, max_log2, n_freqs)).astype(f32)
    outs = [x]
    for fr in freqs:
        outs += [np.sin(x * fr), np.cos(x * fr)]
    return np.concatenate(outs, -1).astype(f32)


def _softmax(x, axis):
    m = np.max(x, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0)
    e = np.exp(x - m)
    return (e / e.sum(axis=axis, keepdims=True)).astype(f32)


def _std_unbiased(x, axis):
    n = x.shape[axis]
    mu = x.mean(axis, keepdims=True)
    return np.sqrt(((x - mu) ** 2).sum(axis) / f32(max(n - 1, 1))).astype(f32) if n > 1 else np.full(np.delete(x.shape, axis), np.nan, f32)


def _view_attention(Wt, pre, q, feat, pos4, mask):
    """Attention2D.forward (transformer_network.py:78-169).  q[R,S,D] (already layer-normed),
    feat[R,S,V,D], pos4[R,S,V,4], mask[R,S,V,1] -> x, attn, k_std_mean inputs"""
    R, S, V, D = feat.shape
    qq = _lin(q, Wt[pre + "q_fc.weight"])
    k = _lin(feat, Wt[pre + "k_fc.weight"])
    v = _lin(k, Wt[pre + "v_fc.weight"])  # sic: v from the projected k (:85)
    valid = (mask[..., 0] != 0)
    cnt = valid.sum(-1)  # valid views per (ray,sample)
    # rows with no valid view have their mask removed (:124-129)
    valid = np.where((cnt == 0)[..., None], True, valid)
    cnt = np.where(cnt == 0, V, cnt)
    # masked unbiased std / normalised std over the valid views (:101-137)
    w = valid[..., None].astype(f32)
    n = cnt[..., None].astype(f32)
    mean = (k * w).sum(2) / n
    var = (((k - mean[:, :, None]) ** 2) * w).sum(2) / np.maximum(n - 1, 1)
    k_std = np.where(n > 1, np.sqrt(var), 0).astype(f32)
    k_std_n = np.where(n > 1, k_std / ((np.abs(k) * w).sum(2) / n + TINY_NUMBER), 0).astype(f32)
    pos = _lin(np.maximum(_lin(pos4, Wt[pre + "pos_fc.0.weight"], Wt[pre + "pos_fc.0.bias"]), 0),
               Wt[pre + "pos_fc.2.weight"], Wt[pre + "pos_fc.2.bias"])
    a = k - qq[:, :, None, :] + pos
    a = _lin(np.maximum(_lin(a, Wt[pre + "attn_fc.0.weight"], Wt[pre + "attn_fc.0.bias"]), 0),
             Wt[pre + "attn_fc.2.weight"], Wt[pre + "attn_fc.2.bias"])
    a = np.where(valid[..., None], a, -np.inf)
    attn = _softmax(a, axis=2)
    x = ((v + pos) * attn).sum(2)
    x = _lin(x, Wt[pre + "out_fc.weight"], Wt[pre + "out_fc.bias"])
    return x.astype(f32), attn, k_std, k_std_n


def _ff(Wt, pre, x):
    return _lin(np.maximum(_lin(x, Wt[pre + "fc1.weight"], Wt[pre + "fc1.bias"]), 0), Wt[pre + "fc2.weight"], Wt[pre + "fc2.bias"])


def _ray_attention(Wt, pre, x, n_heads=4):
    """Attention.forward, attn_mode='qk' (:266-297)."""
    R, S, D = x.shape
    hd = D // n_heads
    sp = lambda t: t.reshape(R, S, n_heads, hd).transpose(0, 2, 1, 3)
    q, k, v = sp(_lin(x, Wt[pre + "q_fc.weight"])), sp(_lin(x, Wt[pre + "k_fc.weight"])), sp(_lin(x, Wt[pre + "v_fc.weight"]))
    attn = _softmax((q @ k.transpose(0, 1, 3, 2)) / f32(np.sqrt(hd)), axis=-1)
    out = (attn @ v).transpose(0, 2, 1, 3).reshape(R, S, D)
    return _lin(out, Wt[pre + "out_fc.weight"], Wt[pre + "out_fc.bias"]).astype(f32), attn


def gnt_forward(Wt, rgb_feat, ray_diff, mask, pts, ray_d):
    """GNT.forward with ret_alpha, ret_view_entropy, ret_view_std (transformer_network.py:423-539).
    Wt: state_dict of net_coarse as numpy arrays.  -> out[R,3+S], extras"""
    depth = len({k.split(".")[1] for k in Wt if k.startswith("view_crosstrans.")})
    viewdirs = ray_d / np.linalg.norm(ray_d, axis=-1, keepdims=True)
    input_views = np.broadcast_to(_embed(viewdirs.astype(f32))[:, None], pts.shape[:2] + (63,))
    input_pts = _embed(pts.astype(f32))
    feat = _lin(np.maximum(_lin(rgb_feat, Wt["rgbfeat_fc.0.weight"], Wt["rgbfeat_fc.0.bias"]), 0),
                Wt["rgbfeat_fc.2.weight"], Wt["rgbfeat_fc.2.bias"]).astype(f32)
    q = feat.max(2)
    std0 = _std_unbiased(feat, 2)
    view_std = [std0.mean(-1)]
    view_std_n = [(std0 / (np.abs(feat).mean(2) + TINY_NUMBER)).mean(-1)]
    view_entropy = []
    weights = None
    for i in range(depth):
        pre = f"view_crosstrans.{i}."
        x, attn, k_std, k_std_n = _view_attention(
            Wt, pre + "attn.", _ln(q, Wt[pre + "attn_norm.weight"], Wt[pre + "attn_norm.bias"], 1e-6), feat, ray_diff, mask)
        x = x + q
        q = _ff(Wt, pre + "ff.", _ln(x, Wt[pre + "ff_norm.weight"], Wt[pre + "ff_norm.bias"], 1e-6)) + x
        if i % 2 == 0:
            q = np.concatenate([q, input_pts, input_views], -1)
            q = _lin(np.maximum(_lin(q, Wt[f"q_fcs.{i}.0.weight"], Wt[f"q_fcs.{i}.0.bias"]), 0),
                     Wt[f"q_fcs.{i}.2.weight"], Wt[f"q_fcs.{i}.2.bias"])
        pre = f"view_selftrans.{i}."
        x, rattn = _ray_attention(Wt, pre + "attn.", _ln(q, Wt[pre + "attn_norm.weight"], Wt[pre + "attn_norm.bias"], 1e-6))
        x = x + q
        q = (_ff(Wt, pre + "ff.", _ln(x, Wt[pre + "ff_norm.weight"], Wt[pre + "ff_norm.bias"], 1e-6)) + x).astype(f32)
        weights = rattn.mean(1)[:, 0]  # row of query sample 0 (:336)
        view_entropy.append((-attn * np.log(attn + f32(1e-8))).sum(2).mean(-1))
        view_std.append(k_std.mean(-1))
        view_std_n.append(k_std_n.mean(-1))
    hfin = _ln(q, Wt["norm.weight"], Wt["norm.bias"], 1e-5)
    rgb = _lin(hfin.mean(1), Wt["rgb_fc.weight"], Wt["rgb_fc.bias"])
    extras = {"view_entropy": np.stack(view_entropy, 2).astype(f32), "view_std": np.stack(view_std, 2).astype(f32),
              "view_std_normalized": np.stack(view_std_n, 2).astype(f32)}
    return np.concatenate([rgb, weights], 1).astype(f32), extras


def render_rays(Wt, ray_o, ray_d, depth_range, n_samples, cam_tgt, src_rgbs, cams_src, featmaps, inv_masks=None):
    """render_rays coarse outputs (renderer.py:207-300) for rays of one batch item."""
    V = src_rgbs.shape[0]
    pts, z = sample_along_camera_ray(ray_o, ray_d, np.broadcast_to(depth_range, (ray_o.shape[0], 2)), n_samples, True)
    pr = projector_compute(pts, cam_tgt, src_rgbs, cams_src, featmaps, inv_masks)
    out, ex = gnt_forward(Wt, pr["rgb_feat"], pr["ray_diff"], pr["mask"], pts, ray_d)
    rgb, w = out[:, :3], out[:, 3:]
    ret = {"rgb": rgb, "weights": w, "depth": (w * z).sum(-1),
           "inbound_cnt": (w * pr["mask_inbound"][..., 0].sum(2) / f32(V)).sum(1),
           "dyn_cnt": (w * pr["mask_invalid"][..., 0].sum(2) / f32(V)).sum(1)}
    for k in ("view_entropy", "view_std", "view_std_normalized"):
        ret[k] = (w[..., None] * ex[k]).sum(1)
    return ret
