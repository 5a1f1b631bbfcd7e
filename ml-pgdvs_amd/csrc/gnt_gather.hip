// A13: GNT epipolar projection + bilinear gathering across source views
// (pgdvs/models/gnt/projector.py:41-115,117-308 with the ray sampling of
// pgdvs/models/gnt/ray_sampler.py:59-123 fused in).  One thread per (ray, sample, view):
// sample the point on the target ray, project it into the source view, gather rgb (3) and
// the C-channel feature vector bilinearly (align_corners=True, zero padding; feature maps
// are addressed with the full-resolution normalisation exactly as upstream :29-39,:251-268),
// the in-bounds / in-front / dynamic masks and the 4-d relative direction encoding.
// Feature maps are channels-last [V,hf,wf,C] so that the 4 corner fetches are contiguous
// C*4-byte runs.
#include "common.h"

namespace pgdvs {

struct GatherArgs {
  const float *ray_o, *ray_d, *depth_range;
  const float *z_in;  // [R,S] explicit sample depths (fine pass) or null
  int64_t depth_range_stride;  // 0: one range for all rays, 2: per ray
  int R, S, V, inv_uniform;
  const float *cam_tgt, *cams_src;
  const float *src_rgbs;  // [V,H,W,3]
  int H, W;
  const float *feat;  // [V,hf,wf,C]
  int hf, wf, C;
  const float *inv_masks;  // [V,H,W] or null
  float *pts, *z_vals, *rgb_feat, *ray_diff, *mask_inbound, *mask_invalid, *mask;
};

__device__ __forceinline__ void bilinear_setup(float px, float py, int Wm, int Hm, int idx[4],
                                               float w[4]) {
  float x0f = floorf(px), y0f = floorf(py);
  bool fin = isfinite(px) && isfinite(py) && fabsf(px) < 1e9f && fabsf(py) < 1e9f;
  int x0 = fin ? (int)x0f : -10, y0 = fin ? (int)y0f : -10, x1 = x0 + 1, y1 = y0 + 1;
  w[0] = ((float)x1 - px) * ((float)y1 - py);
  w[1] = (px - (float)x0) * ((float)y1 - py);
  w[2] = ((float)x1 - px) * (py - (float)y0);
  w[3] = (px - (float)x0) * (py - (float)y0);
  bool inx0 = x0 >= 0 && x0 < Wm, inx1 = x1 >= 0 && x1 < Wm;
  bool iny0 = y0 >= 0 && y0 < Hm, iny1 = y1 >= 0 && y1 < Hm;
  idx[0] = (inx0 && iny0) ? y0 * Wm + x0 : -1;
  idx[1] = (inx1 && iny0) ? y0 * Wm + x1 : -1;
  idx[2] = (inx0 && iny1) ? y1 * Wm + x0 : -1;
  idx[3] = (inx1 && iny1) ? y1 * Wm + x1 : -1;
}

__global__ void __launch_bounds__(256) gnt_gather_kernel(GatherArgs a) {
  const int64_t total = (int64_t)a.R * a.S * a.V;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int v = (int)(t % a.V);
  const int64_t rs = t / a.V;
  const int s = (int)(rs % a.S);
  const int r = (int)(rs / a.S);
  // z sample (ray_sampler.py:59-73), deterministic
  const float near = a.depth_range[(int64_t)r * a.depth_range_stride + 0];
  const float far = a.depth_range[(int64_t)r * a.depth_range_stride + 1];
  float z;
  if (a.z_in) {
    z = a.z_in[rs];
  } else if (a.inv_uniform) {
    float start = 1.0f / near;
    float step = (1.0f / far - start) / (float)(a.S - 1);
    z = 1.0f / (start + (float)s * step);
  } else {
    float step = (far - near) / (float)(a.S - 1);
    z = near + (float)s * step;
  }
  float X[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) X[k] = z * a.ray_d[(size_t)r * 3 + k] + a.ray_o[(size_t)r * 3 + k];
  if (v == 0) {
    if (a.z_vals) a.z_vals[rs] = z;
    if (a.pts)
      for (int k = 0; k < 3; ++k) a.pts[rs * 3 + k] = X[k];
  }
  const float *cam = a.cams_src + (size_t)v * PGDVS_CAM_BLOCK;
  const float *P = cam + PGDVS_CAM_P;
  float p[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float acc = P[i * 4 + 0] * X[0];
    acc = acc + P[i * 4 + 1] * X[1];
    acc = acc + P[i * 4 + 2] * X[2];
    acc = acc + P[i * 4 + 3];
    p[i] = acc;
  }
  float zz = p[2] < 1e-8f ? 1e-8f : p[2];
  float u = clampf(p[0] / zz, -1e6f, 1e6f), w_ = clampf(p[1] / zz, -1e6f, 1e6f);
  const bool in_front = p[2] > 0.0f;
  const float hh = a.cams_src[PGDVS_CAM_HW + 0], ww = a.cams_src[PGDVS_CAM_HW + 1];  // of view 0 (:158)
  const bool inb = (u <= ww - 1.0f) && (u >= 0.0f) && (w_ <= hh - 1.0f) && (w_ >= 0.0f);
  float gx = 2.0f * u / (ww - 1.0f) - 1.0f;
  float gy = 2.0f * w_ / (hh - 1.0f) - 1.0f;
  // rgb
  int idx[4];
  float bw[4];
  bilinear_setup(((gx + 1.0f) / 2.0f) * (float)(a.W - 1), ((gy + 1.0f) / 2.0f) * (float)(a.H - 1), a.W, a.H,
                 idx, bw);
  const int D = 3 + a.C;
  float *out = a.rgb_feat + (size_t)t * D;
  const float *img = a.src_rgbs + (size_t)v * a.H * a.W * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (idx[k] >= 0) acc = acc + img[(size_t)idx[k] * 3 + c] * bw[k];
    out[c] = acc;
  }
  float minv = 0.0f;
  if (a.inv_masks) {
    const float *mk = a.inv_masks + (size_t)v * a.H * a.W;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (idx[k] >= 0) acc = acc + mk[idx[k]] * bw[k];
    minv = acc > 1e-3f ? 1.0f : 0.0f;
  }
  // features
  bilinear_setup(((gx + 1.0f) / 2.0f) * (float)(a.wf - 1), ((gy + 1.0f) / 2.0f) * (float)(a.hf - 1), a.wf, a.hf,
                 idx, bw);
  const float *fm = a.feat + (size_t)v * a.hf * a.wf * a.C;
  for (int c = 0; c < a.C; ++c) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (idx[k] >= 0) acc = acc + fm[(size_t)idx[k] * a.C + c] * bw[k];
    out[3 + c] = acc;
  }
  const float mi = (inb && in_front) ? 1.0f : 0.0f;
  if (a.mask_inbound) a.mask_inbound[t] = mi;
  if (a.mask_invalid) a.mask_invalid[t] = minv;
  a.mask[t] = mi * (1.0f - minv);
  // compute_angle (:75-115)
  const float *qpos = a.cam_tgt + PGDVS_CAM_O;
  const float *tpos = cam + PGDVS_CAM_O;
  float va[3], vb[3], na = 0.0f, nb = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    va[k] = qpos[k] - X[k];
    vb[k] = tpos[k] - X[k];
    na = na + va[k] * va[k];
    nb = nb + vb[k] * vb[k];
  }
  na = sqrtf(na) + 1e-6f;
  nb = sqrtf(nb) + 1e-6f;
  float d[3], dn = 0.0f, dot = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    va[k] = va[k] / na;
    vb[k] = vb[k] / nb;
    d[k] = va[k] - vb[k];
    dn = dn + d[k] * d[k];
    dot = dot + va[k] * vb[k];
  }
  dn = sqrtf(dn);
  dn = dn < 1e-6f ? 1e-6f : dn;
  float *rdo = a.ray_diff + (size_t)t * 4;
#pragma unroll
  for (int k = 0; k < 3; ++k) rdo[k] = d[k] / dn;
  rdo[3] = dot;
}

}  // namespace pgdvs

using namespace pgdvs;

PGDVS_API int pgdvs_gnt_gather(const float *ray_o, const float *ray_d, const float *depth_range,
                               int depth_range_per_ray, const float *z_samples, int R, int S,
                               int inv_uniform, const float *cam_tgt, const float *cams_src, int V,
                               const float *src_rgbs, int H, int W, const float *featmaps_cl, int hf,
                               int wf, int C, const float *inv_masks, float *pts, float *z_vals,
                               float *rgb_feat, float *ray_diff, float *mask_inbound,
                               float *mask_invalid, float *mask, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(ray_o && ray_d && depth_range && cam_tgt && cams_src && src_rgbs && featmaps_cl &&
                    rgb_feat && ray_diff && mask,
                "pgdvs_gnt_gather: null pointer");
  PGDVS_REQUIRE(R >= 0 && (S >= 2 || (z_samples && S >= 1)) && V >= 1 && H > 1 && W > 1 && hf > 0 && wf > 0 && C >= 0,
                "pgdvs_gnt_gather: bad shape");
  if (R == 0) return PGDVS_OK;
  GatherArgs a;
  a.ray_o = ray_o;
  a.ray_d = ray_d;
  a.depth_range = depth_range;
  a.depth_range_stride = depth_range_per_ray ? 2 : 0;
  a.z_in = z_samples;
  a.R = R;
  a.S = S;
  a.V = V;
  a.inv_uniform = inv_uniform;
  a.cam_tgt = cam_tgt;
  a.cams_src = cams_src;
  a.src_rgbs = src_rgbs;
  a.H = H;
  a.W = W;
  a.feat = featmaps_cl;
  a.hf = hf;
  a.wf = wf;
  a.C = C;
  a.inv_masks = inv_masks;
  a.pts = pts;
  a.z_vals = z_vals;
  a.rgb_feat = rgb_feat;
  a.ray_diff = ray_diff;
  a.mask_inbound = mask_inbound;
  a.mask_invalid = mask_invalid;
  a.mask = mask;
  const int64_t total = (int64_t)R * S * V;
  PGDVS_LAUNCH("gnt_gather", gnt_gather_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0,
               as_stream(stream), a);
  return check_launch("gnt_gather");
}
