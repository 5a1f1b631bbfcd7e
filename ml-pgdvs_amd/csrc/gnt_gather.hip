// A13: GNT epipolar projection + bilinear gathering across source views
// (pgdvs/models/gnt/projector.py:41-115,117-308 with the ray sampling of
// pgdvs/models/gnt/ray_sampler.py:59-123 fused in).  Per (ray, sample, view): sample the
// point on the target ray, project it into the source view, gather rgb (3) and the C-channel
// feature vector bilinearly (align_corners=True, zero padding; feature maps are addressed
// with the full-resolution normalisation exactly as upstream :29-39,:251-268), the in-bounds /
// in-front / dynamic masks and the 4-d relative direction encoding.
// Feature maps are channels-last [V,hf,wf,C]: a corner is one contiguous C*4-byte run.  Eight
// lanes share an item (gnt_gather8_kernel, C % 4 == 0): lane q fetches channels 4q..4q+3 of
// each corner as one float4 (the 8 lanes read the 128-byte run of C = 32 in one transaction),
// lanes 0-2 / 3 / 4-7 additionally take the rgb channels / the masks / the direction encoding;
// the wavefront's 8 output rows (8 x (3+C) floats, contiguous) go through LDS and leave as
// float4 stores.  A thread-per-item kernel that walks the channels with scalar loads and writes
// its own 140-byte row measured 1.12 ms per 1024 x 256 x 10 chunk and remains for odd C.
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

struct GatherItem {
  float z, X[3];
  float gx, gy;   // normalised source coordinates
  float mi;       // in bounds and in front
  float rd[4];    // relative direction encoding
};

// ray sample + projection + direction encoding of item (r, s, v)
__device__ __forceinline__ GatherItem gather_item(const GatherArgs &a, int r, int s, int64_t rs, int v) {
  GatherItem it;
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
  it.z = z;
#pragma unroll
  for (int k = 0; k < 3; ++k) it.X[k] = z * a.ray_d[(size_t)r * 3 + k] + a.ray_o[(size_t)r * 3 + k];
  const float *cam = a.cams_src + (size_t)v * PGDVS_CAM_BLOCK;
  const float *P = cam + PGDVS_CAM_P;
  float p[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float acc = P[i * 4 + 0] * it.X[0];
    acc = acc + P[i * 4 + 1] * it.X[1];
    acc = acc + P[i * 4 + 2] * it.X[2];
    acc = acc + P[i * 4 + 3];
    p[i] = acc;
  }
  float zz = p[2] < 1e-8f ? 1e-8f : p[2];
  float u = clampf(p[0] / zz, -1e6f, 1e6f), w_ = clampf(p[1] / zz, -1e6f, 1e6f);
  const bool in_front = p[2] > 0.0f;
  const float hh = a.cams_src[PGDVS_CAM_HW + 0], ww = a.cams_src[PGDVS_CAM_HW + 1];  // of view 0 (:158)
  const bool inb = (u <= ww - 1.0f) && (u >= 0.0f) && (w_ <= hh - 1.0f) && (w_ >= 0.0f);
  it.gx = 2.0f * u / (ww - 1.0f) - 1.0f;
  it.gy = 2.0f * w_ / (hh - 1.0f) - 1.0f;
  it.mi = (inb && in_front) ? 1.0f : 0.0f;
  // compute_angle (:75-115)
  const float *qpos = a.cam_tgt + PGDVS_CAM_O;
  const float *tpos = cam + PGDVS_CAM_O;
  float va[3], vb[3], na = 0.0f, nb = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    va[k] = qpos[k] - it.X[k];
    vb[k] = tpos[k] - it.X[k];
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
#pragma unroll
  for (int k = 0; k < 3; ++k) it.rd[k] = d[k] / dn;
  it.rd[3] = dot;
  return it;
}

__global__ void __launch_bounds__(256) gnt_gather_kernel(GatherArgs a) {
  const int64_t total = (int64_t)a.R * a.S * a.V;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int v = (int)(t % a.V);
  const int64_t rs = t / a.V;
  const int s = (int)(rs % a.S);
  const int r = (int)(rs / a.S);
  const GatherItem it = gather_item(a, r, s, rs, v);
  if (v == 0) {
    if (a.z_vals) a.z_vals[rs] = it.z;
    if (a.pts)
      for (int k = 0; k < 3; ++k) a.pts[rs * 3 + k] = it.X[k];
  }
  // rgb
  int idx[4];
  float bw[4];
  bilinear_setup(((it.gx + 1.0f) / 2.0f) * (float)(a.W - 1), ((it.gy + 1.0f) / 2.0f) * (float)(a.H - 1), a.W, a.H,
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
  bilinear_setup(((it.gx + 1.0f) / 2.0f) * (float)(a.wf - 1), ((it.gy + 1.0f) / 2.0f) * (float)(a.hf - 1), a.wf, a.hf,
                 idx, bw);
  const float *fm = a.feat + (size_t)v * a.hf * a.wf * a.C;
  for (int c = 0; c < a.C; ++c) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (idx[k] >= 0) acc = acc + fm[(size_t)idx[k] * a.C + c] * bw[k];
    out[3 + c] = acc;
  }
  if (a.mask_inbound) a.mask_inbound[t] = it.mi;
  if (a.mask_invalid) a.mask_invalid[t] = minv;
  a.mask[t] = it.mi * (1.0f - minv);
  float *rdo = a.ray_diff + (size_t)t * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) rdo[k] = it.rd[k];
}

// eight lanes per item; requires C % 4 == 0 and 3 + C <= kGatherMaxD
constexpr int kGatherMaxD = 68;

__global__ void __launch_bounds__(256) gnt_gather8_kernel(GatherArgs a) {
  __shared__ __attribute__((aligned(16))) float s_rows[4][8 * kGatherMaxD];
  const int64_t total = (int64_t)a.R * a.S * a.V;
  const int q = threadIdx.x & 7, slot = (threadIdx.x & 63) >> 3, wave = threadIdx.x >> 6;
  const int D = 3 + a.C;
  float *rows = s_rows[wave];
  const int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * 8;  // first item of this wavefront
  if (t0 >= total) return;
  const int64_t t_raw = t0 + slot;
  const bool live = t_raw < total;
  const int64_t t = live ? t_raw : total - 1;
  const int v = (int)(t % a.V);
  const int64_t rs = t / a.V;
  const int s = (int)(rs % a.S);
  const int r = (int)(rs / a.S);
  const GatherItem it = gather_item(a, r, s, rs, v);
  if (live && v == 0 && q == 0) {
    if (a.z_vals) a.z_vals[rs] = it.z;
    if (a.pts)
      for (int k = 0; k < 3; ++k) a.pts[rs * 3 + k] = it.X[k];
  }
  int idx[4];
  float bw[4];
  bilinear_setup(((it.gx + 1.0f) / 2.0f) * (float)(a.W - 1), ((it.gy + 1.0f) / 2.0f) * (float)(a.H - 1), a.W, a.H,
                 idx, bw);
  if (q < 3) {  // rgb channel q
    const float *img = a.src_rgbs + (size_t)v * a.H * a.W * 3;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (idx[k] >= 0) acc = acc + img[(size_t)idx[k] * 3 + q] * bw[k];
    rows[slot * D + q] = acc;
  } else if (q == 3) {  // masks
    float minv = 0.0f;
    if (a.inv_masks) {
      const float *mk = a.inv_masks + (size_t)v * a.H * a.W;
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (idx[k] >= 0) acc = acc + mk[idx[k]] * bw[k];
      minv = acc > 1e-3f ? 1.0f : 0.0f;
    }
    if (live) {
      if (a.mask_inbound) a.mask_inbound[t] = it.mi;
      if (a.mask_invalid) a.mask_invalid[t] = minv;
      a.mask[t] = it.mi * (1.0f - minv);
    }
  } else if (live) {  // direction encoding, one component per lane
    a.ray_diff[(size_t)t * 4 + (q - 4)] = it.rd[q - 4];
  }
  // features: channels 4q .. 4q+3 (+32, +64, ...)
  bilinear_setup(((it.gx + 1.0f) / 2.0f) * (float)(a.wf - 1), ((it.gy + 1.0f) / 2.0f) * (float)(a.hf - 1), a.wf, a.hf,
                 idx, bw);
  const float *fm = a.feat + (size_t)v * a.hf * a.wf * a.C;
  for (int c0 = 4 * q; c0 < a.C; c0 += 32) {
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (idx[k] >= 0) {
        const float4 f = *reinterpret_cast<const float4 *>(fm + (size_t)idx[k] * a.C + c0);
        acc[0] = acc[0] + f.x * bw[k];
        acc[1] = acc[1] + f.y * bw[k];
        acc[2] = acc[2] + f.z * bw[k];
        acc[3] = acc[3] + f.w * bw[k];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) rows[slot * D + 3 + c0 + j] = acc[j];
  }
  // the wavefront's rows are contiguous in rgb_feat: [t0 * D, (t0 + 8) * D), a multiple of 16 bytes
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wavefront's LDS writes have landed
  const int64_t nlive = total - t0 < 8 ? total - t0 : 8;
  const int nfl = (int)nlive * D;
  float *dst = a.rgb_feat + (size_t)t0 * D;
  const int lane = threadIdx.x & 63;
  for (int k = 4 * lane; k < nfl; k += 256) {
    if (k + 4 <= nfl) {
      *reinterpret_cast<float4 *>(dst + k) = *reinterpret_cast<const float4 *>(rows + k);
    } else {
      for (int j = k; j < nfl; ++j) dst[j] = rows[j];
    }
  }
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
  if (C % 4 == 0 && 3 + C <= kGatherMaxD && (reinterpret_cast<uintptr_t>(featmaps_cl) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(rgb_feat) & 15) == 0) {
    PGDVS_LAUNCH("gnt_gather", gnt_gather8_kernel, dim3((unsigned)cdiv(total, 32)), dim3(256), 0, as_stream(stream), a);
  } else {
    PGDVS_LAUNCH("gnt_gather", gnt_gather_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(stream), a);
  }
  return check_launch("gnt_gather");
}
