"""Importance re-sampling along rays: mirror of ``sample_pdf`` / ``sample_fine_pts``
(pgdvs/models/gnt/ray_sampler.py:10-56,183-220).  Small [rays, samples] tensor algebra between
the coarse and the fine network pass (torch; the deterministic coarse sampling itself is fused
into the gather kernel, ``pgdvs_gnt_gather``)."""
import torch


def sample_pdf(bins, weights, N_samples, det=False):
    """bins[R,M+1], weights[R,M] -> samples[R,N_samples] by inverting the piecewise-constant CDF."""
    M = weights.shape[1]
    weights = weights + 1e-5  # (in place upstream; the caller's tensor is a detached clone there)
    pdf = weights / torch.sum(weights, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    cdf = torch.cat([torch.zeros_like(cdf[:, 0:1]), cdf], dim=-1)  # [R,M+1]
    if det:
        u = torch.linspace(0.0, 1.0, N_samples, device=bins.device).unsqueeze(0).repeat(bins.shape[0], 1)
    else:
        u = torch.rand(bins.shape[0], N_samples, device=bins.device)
    # number of CDF knots (of the first M) that are <= u
    above = (u[:, :, None] >= cdf[:, None, :M]).sum(dim=-1)
    below = torch.clamp(above - 1, min=0)
    inds = torch.stack((below, above), dim=2)  # [R,N,2]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(-1, N_samples, -1), -1, inds)
    bins_g = torch.gather(bins.unsqueeze(1).expand(-1, N_samples, -1), -1, inds)
    denom = cdf_g[:, :, 1] - cdf_g[:, :, 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[:, :, 0]) / denom
    return bins_g[:, :, 0] + t * (bins_g[:, :, 1] - bins_g[:, :, 0])


def sample_fine_z(inv_uniform, N_importance, det, weights, z_vals):
    """sorted union of the coarse depths and N_importance depths drawn from the coarse weights"""
    w = weights[:, 1:-1]
    if inv_uniform:
        inv_z = 1.0 / z_vals
        inv_mid = 0.5 * (inv_z[:, 1:] + inv_z[:, :-1])
        z_samples = 1.0 / sample_pdf(torch.flip(inv_mid, dims=[1]), torch.flip(w, dims=[1]), N_importance, det=det)
    else:
        mid = 0.5 * (z_vals[:, 1:] + z_vals[:, :-1])
        z_samples = sample_pdf(mid, w, N_importance, det=det)
    z_all, _ = torch.sort(torch.cat((z_vals, z_samples), dim=-1), dim=-1)
    return z_all
