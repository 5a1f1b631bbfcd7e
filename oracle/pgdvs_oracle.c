/*
 * pgdvs_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the PGDVS per-target-view rendering inner loop
 * (SURVEY.md section 8a rows A1..A12, A17).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (ml-pgdvs_amd/) never does.
 *
 * Every function cites the reference file:line (relative to the upstream
 * apple/ml-pgdvs tree) whose behaviour it restates.  Arithmetic is IEEE fp32
 * (fp64 for the static-aggregation row, which is float64 numpy upstream) with
 * a fixed left-to-right operation order and no FMA contraction: build with
 *     gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 * so that the HIP kernels, which use the same operation order, can be compared
 * bit-for-bit on the integer / index paths.
 *
 * Parity pinning:
 *   - rows A1..A8, A11, A12 are pinned against golden vectors produced by
 *     importing the reference itself (tests/golden/make_golden.py); row A17
 *     likewise (tests/golden/make_golden_track.py).
 *   - rows A9 (point rasteriser + norm-weighted compositor) and the kNN used
 *     by A4 restate pytorch3d 0.7.4 (un-vendored third-party dependency,
 *     README.md:38 of the reference); pytorch3d is not installable here, so
 *     for A9 parity is UNPINNED (restated from its published algorithm:
 *     pytorch3d/csrc/rasterize_points/rasterize_points_cpu.cpp
 *     RasterizePointsNaiveCpu, pytorch3d/csrc/compositing/
 *     norm_weighted_sum_cpu.cpp, pytorch3d/renderer/points/renderer.py).
 *     The same holds for the mesh variant of A10 (MeshRasterizer, see
 *     orc_mesh_render): parity UNPINNED.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* Camera block: derived per-camera constants.                          */
/* layout (floats):                                                     */
/*   [ 0: 9]  Kinv   = inverse(K[:3,:3])                                 */
/*   [ 9:18]  M      = c2w[:3,:3] @ Kinv     (ray direction matrix)      */
/*   [18:21]  o      = c2w[:3,3]                                         */
/*   [21:37]  P      = K(4x4) @ inverse(c2w) (projection matrix)         */
/*   [37:53]  w2c    = inverse(c2w)                                      */
/*   [53:62]  R      = c2w[:3,:3]                                        */
/*   [62:64]  h, w                                                       */
/*   [64:80]  K (4x4) as given                                           */
/* ------------------------------------------------------------------ */
#define CAM_KINV 0
#define CAM_M 9
#define CAM_O 18
#define CAM_P 21
#define CAM_W2C 37
#define CAM_R 53
#define CAM_HW 62
#define CAM_K 64
#define CAM_BLOCK 80

/* Gauss-Jordan inverse with partial pivoting in fp64 (n <= 4). */
static int inv_f64(const double *a, double *out, int n) {
  double m[4][8];
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) {
      m[i][j] = a[i * n + j];
      m[i][n + j] = (i == j) ? 1.0 : 0.0;
    }
  }
  for (int c = 0; c < n; ++c) {
    int piv = c;
    double best = fabs(m[c][c]);
    for (int r = c + 1; r < n; ++r) {
      double v = fabs(m[r][c]);
      if (v > best) {
        best = v;
        piv = r;
      }
    }
    if (best == 0.0) return -1;
    if (piv != c) {
      for (int j = 0; j < 2 * n; ++j) {
        double t = m[c][j];
        m[c][j] = m[piv][j];
        m[piv][j] = t;
      }
    }
    double d = m[c][c];
    for (int j = 0; j < 2 * n; ++j) m[c][j] = m[c][j] / d;
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      double f = m[r][c];
      if (f == 0.0) continue;
      for (int j = 0; j < 2 * n; ++j) m[r][j] = m[r][j] - f * m[c][j];
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) out[i * n + j] = m[i][n + j];
  return 0;
}

ORC_API int orc_inv_f64(const double *a, double *out, int n) {
  return inv_f64(a, out, n);
}

/*
 * flat_cam[34] = [h, w, K(4x4 row-major), c2w(4x4 row-major)]
 * (pgdvs/renderers/pgdvs_renderer.py:354-357, pgdvs/datasets/nvidia_eval.py:827-832).
 * Derived matrices follow pgdvs_renderer_base.py:40-45 (M = c2w[:3,:3] @ inv(K[:3,:3]))
 * and gnt/projector.py:49-60 (P = K @ inv(c2w)).  torch.inverse is fp32 LU upstream;
 * here the inverse is formed in fp64 and rounded once to fp32 (|delta| ~ 1 ulp).
 */
ORC_API int orc_cam_prep(const float *flat_cam, float *blk) {
  const float *K = flat_cam + 2;
  const float *c2w = flat_cam + 18;
  double k3[9], k3i[9], c4[16], c4i[16];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) k3[i * 3 + j] = (double)K[i * 4 + j];
  for (int i = 0; i < 16; ++i) c4[i] = (double)c2w[i];
  if (inv_f64(k3, k3i, 3) != 0) return -1;
  if (inv_f64(c4, c4i, 4) != 0) return -2;
  float kinv[9], w2c[16];
  for (int i = 0; i < 9; ++i) kinv[i] = (float)k3i[i];
  for (int i = 0; i < 16; ++i) w2c[i] = (float)c4i[i];
  for (int i = 0; i < 9; ++i) blk[CAM_KINV + i] = kinv[i];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      float s = c2w[i * 4 + 0] * kinv[0 * 3 + j];
      s = s + c2w[i * 4 + 1] * kinv[1 * 3 + j];
      s = s + c2w[i * 4 + 2] * kinv[2 * 3 + j];
      blk[CAM_M + i * 3 + j] = s;
      blk[CAM_R + i * 3 + j] = c2w[i * 4 + j];
    }
    blk[CAM_O + i] = c2w[i * 4 + 3];
  }
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 4; ++j) {
      float s = K[i * 4 + 0] * w2c[0 * 4 + j];
      s = s + K[i * 4 + 1] * w2c[1 * 4 + j];
      s = s + K[i * 4 + 2] * w2c[2 * 4 + j];
      s = s + K[i * 4 + 3] * w2c[3 * 4 + j];
      blk[CAM_P + i * 4 + j] = s;
    }
  }
  for (int i = 0; i < 16; ++i) blk[CAM_W2C + i] = w2c[i];
  blk[CAM_HW + 0] = flat_cam[0];
  blk[CAM_HW + 1] = flat_cam[1];
  for (int i = 0; i < 16; ++i) blk[CAM_K + i] = K[i];
  return 0;
}

/* ------------------------------------------------------------------ */
/* A1: get_batched_rays (pgdvs/renderers/pgdvs_renderer_base.py:17-57)  */
/* integer pixel centres (no +0.5), rays_d un-normalised, stride.       */
/* outputs: rays_o[n,3], rays_d[n,3], uvs[n,2], n = rh*rw               */
/* ------------------------------------------------------------------ */
ORC_API void orc_get_rays(const float *blk, int H, int W, int stride, float *rays_o,
                          float *rays_d, float *uvs) {
  const float *M = blk + CAM_M;
  const float *o = blk + CAM_O;
  int rh = (H + stride - 1) / stride, rw = (W + stride - 1) / stride;
#pragma omp parallel for
  for (int r = 0; r < rh; ++r) {
    for (int c = 0; c < rw; ++c) {
      int i = r * rw + c;
      float u = (float)(c * stride), v = (float)(r * stride);
      for (int k = 0; k < 3; ++k) {
        float d = M[k * 3 + 0] * u;
        d = d + M[k * 3 + 1] * v;
        d = d + M[k * 3 + 2];
        rays_d[i * 3 + k] = d;
        rays_o[i * 3 + k] = o[k];
      }
      uvs[i * 2 + 0] = u;
      uvs[i * 2 + 1] = v;
    }
  }
}

/* ------------------------------------------------------------------ */
/* A2 + A3: unproject + flow-guided temporal warp, dense over [H,W]     */
/* (pgdvs/renderers/pgdvs_renderer_dyn.py:299-388).                     */
/*   mask_eff[p]  : dyn_mask_1 after optional flow-consistency (:304-308)*/
/*   valid[p]     : mask_eff && 0<=uv2<=(W-1,H-1)        (:309-316)      */
/*   pcl[p,3]     : time-lerped world point               (:318-320,385-388)*/
/*   rgbf[p,3]    : colour attached to the point          (:333-337,350-356)*/
/* grid_sample restated from ATen GridSampler (CUDA flavour):            */
/*   unnormalize(align_corners=False) = ((g + 1) * size - 1) / 2         */
/*   nearest = nearbyint (round-half-even), zero padding.                */
/* ------------------------------------------------------------------ */
static inline float fclampf(float x, float lo, float hi) {
  return x < lo ? lo : (x > hi ? hi : x);
}

ORC_API void orc_dyn_warp(int H, int W, const float *dyn_mask1, const float *occ,
                          int use_flow_consistency, const float *flow12,
                          const float *depth1, const float *depth2, const float *rgb1,
                          const float *rgb2, const float *cam1, const float *cam2,
                          float t1, float t2, float tt, uint8_t *mask_eff, uint8_t *valid,
                          float *pcl, float *rgbf) {
  const float *M1 = cam1 + CAM_M, *o1 = cam1 + CAM_O;
  const float *Kinv2 = cam2 + CAM_KINV, *R2 = cam2 + CAM_R, *o2 = cam2 + CAM_O;
  const float fw = (float)W, fh = (float)H;
  const int same_time = (t1 == t2);
  float w1 = 0.f, w2 = 0.f;
  if (!same_time) {
    w1 = (t2 - tt) / (t2 - t1);
    w2 = (tt - t1) / (t2 - t1);
  }
#pragma omp parallel for
  for (int r = 0; r < H; ++r) {
    for (int c = 0; c < W; ++c) {
      int p = r * W + c;
      float u = (float)c, v = (float)r;
      int m = dyn_mask1[p] != 0.0f;
      if (use_flow_consistency) m = m && !(occ[p] > 0.0f);
      mask_eff[p] = (uint8_t)m;
      float ux = u + flow12[p * 2 + 0];
      float uy = v + flow12[p * 2 + 1];
      int ok = m && (ux >= 0.0f) && (ux <= fw - 1.0f) && (uy >= 0.0f) && (uy <= fh - 1.0f);
      valid[p] = (uint8_t)ok;
      float X1[3];
      for (int k = 0; k < 3; ++k) {
        float d = M1[k * 3 + 0] * u;
        d = d + M1[k * 3 + 1] * v;
        d = d + M1[k * 3 + 2];
        X1[k] = o1[k] + d * depth1[p];
      }
      if (!ok) {
        for (int k = 0; k < 3; ++k) {
          pcl[p * 3 + k] = 0.0f;
          rgbf[p * 3 + k] = 0.0f;
        }
        continue;
      }
      if (same_time) {
        for (int k = 0; k < 3; ++k) {
          pcl[p * 3 + k] = X1[k];
          rgbf[p * 3 + k] = rgb1[p * 3 + k];
        }
        continue;
      }
      /* grid = 2*uv/raw_shape - 1 (:341), default align_corners=False */
      float gx = 2.0f * ux / fw - 1.0f;
      float gy = 2.0f * uy / fh - 1.0f;
      float ix = ((gx + 1.0f) * fw - 1.0f) / 2.0f;
      float iy = ((gy + 1.0f) * fh - 1.0f) / 2.0f;
      /* nearest depth (:342-348) */
      float nx = nearbyintf(ix), ny = nearbyintf(iy);
      float dsamp = 0.0f;
      if (nx >= 0.0f && nx <= fw - 1.0f && ny >= 0.0f && ny <= fh - 1.0f)
        dsamp = depth2[(int)ny * W + (int)nx];
      /* bilinear rgb (:350-356), zero padding */
      float x0f = floorf(ix), y0f = floorf(iy);
      int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
      float wnw = ((float)x1 - ix) * ((float)y1 - iy);
      float wne = (ix - (float)x0) * ((float)y1 - iy);
      float wsw = ((float)x1 - ix) * (iy - (float)y0);
      float wse = (ix - (float)x0) * (iy - (float)y0);
      float col[3] = {0.f, 0.f, 0.f};
      for (int k = 0; k < 3; ++k) {
        float acc = 0.0f;
        if (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H) acc = acc + rgb2[(y0 * W + x0) * 3 + k] * wnw;
        if (x1 >= 0 && x1 < W && y0 >= 0 && y0 < H) acc = acc + rgb2[(y0 * W + x1) * 3 + k] * wne;
        if (x0 >= 0 && x0 < W && y1 >= 0 && y1 < H) acc = acc + rgb2[(y1 * W + x0) * 3 + k] * wsw;
        if (x1 >= 0 && x1 < W && y1 >= 0 && y1 < H) acc = acc + rgb2[(y1 * W + x1) * 3 + k] * wse;
        col[k] = acc;
      }
      /* frame-2 ray: c2w_2[:3,:3] @ (inv(K_2[:3,:3]) @ [uv2,1]) (:358-372) */
      float kq[3];
      for (int k = 0; k < 3; ++k) {
        float s = Kinv2[k * 3 + 0] * ux;
        s = s + Kinv2[k * 3 + 1] * uy;
        s = s + Kinv2[k * 3 + 2];
        kq[k] = s;
      }
      for (int k = 0; k < 3; ++k) {
        float d = R2[k * 3 + 0] * kq[0];
        d = d + R2[k * 3 + 1] * kq[1];
        d = d + R2[k * 3 + 2] * kq[2];
        float X2 = o2[k] + d * dsamp;
        pcl[p * 3 + k] = w1 * X1[k] + w2 * X2;
        rgbf[p * 3 + k] = col[k];
      }
    }
  }
}

/* ------------------------------------------------------------------ */
/* A4: brute-force kNN mean squared distance                            */
/* (pgdvs_renderer_dyn.py:405-419; pytorch3d.ops.knn_points semantics: */
/* K nearest by squared L2, ascending).  out[i] = mean of the K         */
/* smallest squared distances after dropping the smallest one (column   */
/* 0 = the point itself).  When N < K+1 the missing columns are zero    */
/* (pytorch3d pads dists with 0).                                       */
/* ------------------------------------------------------------------ */
ORC_API void orc_knn_mean_dist(const float *pts, int64_t N, int K, float *out) {
  int KK = K + 1;
#pragma omp parallel
  {
    float *best = (float *)malloc(sizeof(float) * (size_t)KK);
#pragma omp for schedule(dynamic, 64)
    for (int64_t i = 0; i < N; ++i) {
      int cnt = 0;
      float qx = pts[i * 3], qy = pts[i * 3 + 1], qz = pts[i * 3 + 2];
      for (int64_t j = 0; j < N; ++j) {
        float dx = qx - pts[j * 3], dy = qy - pts[j * 3 + 1], dz = qz - pts[j * 3 + 2];
        float d = dx * dx;
        d = d + dy * dy;
        d = d + dz * dz;
        if (cnt < KK) {
          int k = cnt++;
          while (k > 0 && best[k - 1] > d) {
            best[k] = best[k - 1];
            --k;
          }
          best[k] = d;
        } else if (d < best[KK - 1]) {
          int k = KK - 1;
          while (k > 0 && best[k - 1] > d) {
            best[k] = best[k - 1];
            --k;
          }
          best[k] = d;
        }
      }
      /* torch.mean over K columns (cols 1..K).  torch does not specify a summation
       * order (cascade on CPU, tree on CUDA); this restatement fixes one: a butterfly
       * over 64 slots (s[i] += s[i+32], += s[i+16], ... ) when K+1 <= 64, which is what
       * the HIP kernels use, else a plain ascending sum. */
      float s = 0.0f;
      if (KK <= 64) {
        float sl[64];
        for (int k = 0; k < 64; ++k) sl[k] = 0.0f;
        for (int k = 1; k < KK; ++k) sl[k] = k < cnt ? best[k] : 0.0f;
        for (int off = 32; off > 0; off >>= 1)
          for (int k = 0; k < off; ++k) sl[k] = sl[k] + sl[k + off];
        s = sl[0];
      } else {
        for (int k = 1; k < KK; ++k) s = s + (k < cnt ? best[k] : 0.0f);
      }
      out[i] = s / (float)K;
    }
    free(best);
  }
}

/* ------------------------------------------------------------------ */
/* A5: project world points into the target camera                     */
/* (gnt/projector.py:41-73 as called from pgdvs_renderer_dyn.py:470-475)*/
/*   p = P @ [X,1];  uv = p[:2] / clamp(p[2], min=1e-8); clamp +-1e6    */
/* ------------------------------------------------------------------ */
static inline void project_pt(const float *P, const float *X, float *uv) {
  float p[3];
  for (int i = 0; i < 3; ++i) {
    float s = P[i * 4 + 0] * X[0];
    s = s + P[i * 4 + 1] * X[1];
    s = s + P[i * 4 + 2] * X[2];
    s = s + P[i * 4 + 3];
    p[i] = s;
  }
  float z = p[2] < 1e-8f ? 1e-8f : p[2];
  uv[0] = fclampf(p[0] / z, -1e6f, 1e6f);
  uv[1] = fclampf(p[1] / z, -1e6f, 1e6f);
}

ORC_API void orc_project(const float *cam_tgt, const float *pts, int64_t N, float *uv) {
  const float *P = cam_tgt + CAM_P;
#pragma omp parallel for
  for (int64_t i = 0; i < N; ++i) project_pt(P, pts + i * 3, uv + i * 2);
}

/* dense variant: flow_1_to_tgt[p] = proj(pcl[p]) - uv1[p] for keep[p]!=0, else 0
 * (pgdvs_renderer_dyn.py:477-503) */
ORC_API void orc_project_flow_dense(int H, int W, const float *cam_tgt, const float *pcl,
                                    const uint8_t *keep, float *flow_1_to_tgt,
                                    float *valid_mask) {
  const float *P = cam_tgt + CAM_P;
#pragma omp parallel for
  for (int r = 0; r < H; ++r) {
    for (int c = 0; c < W; ++c) {
      int p = r * W + c;
      if (keep[p]) {
        float uv[2];
        project_pt(P, pcl + p * 3, uv);
        flow_1_to_tgt[p * 2 + 0] = uv[0] - (float)c;
        flow_1_to_tgt[p * 2 + 1] = uv[1] - (float)r;
        valid_mask[p] = 1.0f;
      } else {
        flow_1_to_tgt[p * 2 + 0] = 0.0f;
        flow_1_to_tgt[p * 2 + 1] = 0.0f;
        valid_mask[p] = 0.0f;
      }
    }
  }
}

/* ------------------------------------------------------------------ */
/* A6: softsplat importance metric                                     */
/* (pgdvs_renderer_base.py:68-87 L1 + clip, :91-138 backwarp).          */
/* inputs NCHW planes: rgb1[3,H,W] rgb2[3,H,W] flow[2,H,W]              */
/* out l1[H,W] (= softsplat_metric_src1_to_src2)                        */
/* backwarp grid: linspace(-1,1,W)[x] + flow_x / ((W-1)/2),             */
/* grid_sample(bilinear, zeros, align_corners=True):                    */
/*   unnormalize = ((g + 1) / 2) * (size - 1)                           */
/* linspace restated from ATen: step=(end-start)/(steps-1);             */
/*   i < steps/2 ? start + step*i : end - step*(steps-1-i)              */
/* ------------------------------------------------------------------ */
static inline float linspace_m1_1(int i, int steps) {
  if (steps == 1) return -1.0f;
  float step = (1.0f - (-1.0f)) / (float)(steps - 1);
  if (i < steps / 2) return -1.0f + step * (float)i;
  return 1.0f - step * (float)(steps - 1 - i);
}

ORC_API void orc_backwarp_l1(int H, int W, const float *rgb1, const float *rgb2,
                             const float *flow, float *l1) {
  const float hw_x = ((float)W - 1.0f) / 2.0f, hw_y = ((float)H - 1.0f) / 2.0f;
  const int P = H * W;
#pragma omp parallel for
  for (int r = 0; r < H; ++r) {
    for (int c = 0; c < W; ++c) {
      int p = r * W + c;
      float gx = linspace_m1_1(c, W) + flow[p] / hw_x;
      float gy = linspace_m1_1(r, H) + flow[P + p] / hw_y;
      float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
      float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
      float x0f = floorf(ix), y0f = floorf(iy);
      /* non-finite / huge coordinates: ATen casts to int (UB); treat as out of bounds */
      int fin = isfinite(ix) && isfinite(iy) && fabsf(ix) < 1e9f && fabsf(iy) < 1e9f;
      int x0 = fin ? (int)x0f : -10, y0 = fin ? (int)y0f : -10, x1 = x0 + 1, y1 = y0 + 1;
      float wnw = ((float)x1 - ix) * ((float)y1 - iy);
      float wne = (ix - (float)x0) * ((float)y1 - iy);
      float wsw = ((float)x1 - ix) * (iy - (float)y0);
      float wse = (ix - (float)x0) * (iy - (float)y0);
      float s = 0.0f;
      for (int k = 0; k < 3; ++k) {
        const float *pl = rgb2 + (size_t)k * P;
        float acc = 0.0f;
        if (fin) {
          if (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H) acc = acc + pl[y0 * W + x0] * wnw;
          if (x1 >= 0 && x1 < W && y0 >= 0 && y0 < H) acc = acc + pl[y0 * W + x1] * wne;
          if (x0 >= 0 && x0 < W && y1 >= 0 && y1 < H) acc = acc + pl[y1 * W + x0] * wsw;
          if (x1 >= 0 && x1 < W && y1 >= 0 && y1 < H) acc = acc + pl[y1 * W + x1] * wse;
        }
        s = s + fabsf(rgb1[(size_t)k * P + p] - acc);
      }
      l1[p] = s / 3.0f;
    }
  }
}

/* ------------------------------------------------------------------ */
/* A7: softsplat forward = kernel softsplat_out                        */
/* (pgdvs/utils/softsplat.py:352-402).  in[B,C,H,W] flow[B,2,H,W]       */
/* out[B,C,H,W] must be zeroed by the caller (new_zeros, :343-345).     */
/* Accumulation order here is source-pixel raster order per (n,c) plane */
/* (the CUDA kernel's atomicAdd order is unspecified).                  */
/* ------------------------------------------------------------------ */
ORC_API void orc_softsplat_fwd(const float *in, const float *flow, float *out, int B, int C,
                               int H, int W) {
  const size_t P = (size_t)H * W;
#pragma omp parallel for collapse(2)
  for (int n = 0; n < B; ++n) {
    for (int ch = 0; ch < C; ++ch) {
      const float *src = in + ((size_t)n * C + ch) * P;
      float *dst = out + ((size_t)n * C + ch) * P;
      const float *fx = flow + ((size_t)n * 2 + 0) * P;
      const float *fy = flow + ((size_t)n * 2 + 1) * P;
      for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
          size_t p = (size_t)y * W + x;
          float X = (float)x + fx[p];
          float Y = (float)y + fy[p];
          if (!isfinite(X) || !isfinite(Y)) continue;
          float v = src[p];
          /* (int) floor(): values outside the int range are UB upstream; the
           * bounds test below rejects them either way, so clamp first. */
          float flx = floorf(X), fly = floorf(Y);
          if (flx < -2.0f || flx > (float)W || fly < -2.0f || fly > (float)H) continue;
          int nwx = (int)flx, nwy = (int)fly;
          int nex = nwx + 1, ney = nwy;
          int swx = nwx, swy = nwy + 1;
          int sex = nwx + 1, sey = nwy + 1;
          float wnw = ((float)sex - X) * ((float)sey - Y);
          float wne = (X - (float)swx) * ((float)swy - Y);
          float wsw = ((float)nex - X) * (Y - (float)ney);
          float wse = (X - (float)nwx) * (Y - (float)nwy);
          if (nwx >= 0 && nwx < W && nwy >= 0 && nwy < H) dst[(size_t)nwy * W + nwx] += v * wnw;
          if (nex >= 0 && nex < W && ney >= 0 && ney < H) dst[(size_t)ney * W + nex] += v * wne;
          if (swx >= 0 && swx < W && swy >= 0 && swy < H) dst[(size_t)swy * W + swx] += v * wsw;
          if (sex >= 0 && sex < W && sey >= 0 && sey < H) dst[(size_t)sey * W + sex] += v * wse;
        }
      }
    }
  }
}

/* softsplat backward (softsplat.py:459-617): kernels softsplat_ingrad and
 * softsplat_flowgrad restated.  ingrad[n,c,y,x] = sum_corner outgrad[corner] * w_corner;
 * flowgrad[n,0/1,y,x] = sum_c sum_corner outgrad[c,corner] * in[c] * d(w_corner)/d(fx|fy), in
 * the kernels' accumulation order.  Pixels with a non-finite target get zero gradients (the
 * kernels return early on a zero-initialised buffer).  Either output may be NULL. */
ORC_API void orc_softsplat_bwd(const float *in, const float *flow, const float *outgrad,
                               float *ingrad, float *flowgrad, int B, int C, int H, int W) {
  const size_t P = (size_t)H * W;
#pragma omp parallel for collapse(2)
  for (int n = 0; n < B; ++n) {
    for (int y = 0; y < H; ++y) {
      for (int x = 0; x < W; ++x) {
        size_t p = (size_t)y * W + x;
        float X = (float)x + flow[((size_t)n * 2 + 0) * P + p];
        float Y = (float)y + flow[((size_t)n * 2 + 1) * P + p];
        float gfx = 0.0f, gfy = 0.0f;
        int ok = isfinite(X) && isfinite(Y);
        float flx = floorf(X), fly = floorf(Y);
        if (ok && (flx < -2.0f || flx > (float)W || fly < -2.0f || fly > (float)H)) ok = 0; /* every corner fails the bounds test */
        int nwx = ok ? (int)flx : -8, nwy = ok ? (int)fly : -8;
        int cx[4] = {nwx, nwx + 1, nwx, nwx + 1}, cy[4] = {nwy, nwy, nwy + 1, nwy + 1};
        float sex = (float)(nwx + 1), sey = (float)(nwy + 1), wx = (float)nwx, wy = (float)nwy;
        float w[4] = {(sex - X) * (sey - Y), (X - wx) * (sey - Y), (sex - X) * (Y - wy), (X - wx) * (Y - wy)};
        float dx[4] = {-1.0f * (sey - Y), +1.0f * (sey - Y), -1.0f * (Y - wy), +1.0f * (Y - wy)};
        float dy[4] = {(sex - X) * -1.0f, (X - wx) * -1.0f, (sex - X) * +1.0f, (X - wx) * +1.0f};
        for (int ch = 0; ch < C; ++ch) {
          const float *og = outgrad + ((size_t)n * C + ch) * P;
          float v = in[((size_t)n * C + ch) * P + p];
          float gi = 0.0f;
          for (int k = 0; k < 4; ++k) {
            if (!(ok && cx[k] >= 0 && cx[k] < W && cy[k] >= 0 && cy[k] < H)) continue;
            float g = og[(size_t)cy[k] * W + cx[k]];
            gi = gi + g * w[k];
            gfx = gfx + g * v * dx[k];
            gfy = gfy + g * v * dy[k];
          }
          if (ingrad) ingrad[((size_t)n * C + ch) * P + p] = gi;
        }
        if (flowgrad) {
          flowgrad[((size_t)n * 2 + 0) * P + p] = gfx;
          flowgrad[((size_t)n * 2 + 1) * P + p] = gfy;
        }
      }
    }
  }
}

/* corner indices of the splat (the bit-exact integer path): idx[p,4] = flat
 * destination index y*W+x of NW,NE,SW,SE or -1 if dropped. */
ORC_API void orc_softsplat_corners(const float *flow, int H, int W, int32_t *idx) {
  const size_t P = (size_t)H * W;
#pragma omp parallel for
  for (int y = 0; y < H; ++y) {
    for (int x = 0; x < W; ++x) {
      size_t p = (size_t)y * W + x;
      int32_t *o = idx + p * 4;
      o[0] = o[1] = o[2] = o[3] = -1;
      float X = (float)x + flow[p];
      float Y = (float)y + flow[P + p];
      if (!isfinite(X) || !isfinite(Y)) continue;
      float flx = floorf(X), fly = floorf(Y);
      if (flx < -2.0f || flx > (float)W || fly < -2.0f || fly > (float)H) continue;
      int nwx = (int)flx, nwy = (int)fly;
      int xs[4] = {nwx, nwx + 1, nwx, nwx + 1};
      int ys[4] = {nwy, nwy, nwy + 1, nwy + 1};
      for (int k = 0; k < 4; ++k)
        if (xs[k] >= 0 && xs[k] < W && ys[k] >= 0 && ys[k] < H) o[k] = ys[k] * W + xs[k];
    }
  }
}

/* ------------------------------------------------------------------ */
/* A9: naive point rasteriser + norm-weighted compositor               */
/* pytorch3d 0.7.4 semantics (see file header: parity UNPINNED).        */
/* Call sites: pgdvs/renderers/st_geo_renderer.py:77-120,               */
/*             pgdvs/renderers/pgdvs_renderer_dyn.py:671-724.           */
/* Camera conversion: pgdvs/utils/pytorch3d_utils.py:5-47               */
/*   s = min(W,H)/2; f_ndc = f/s; p0 = -(pp - (W,H)/2)/s;               */
/*   view = w2c applied then x,y negated.                               */
/* Point -> NDC: x_ndc = (fx_ndc*xv + p0x*zv)/zv (4x4 projective        */
/* transform then homogeneous divide), z = zv.                          */
/* Pixel -> NDC (rasterization_utils.cuh PixToNonSquareNdc):            */
/*   xf(xi) with xidx = W-1-xi : -off + (range*xidx + off)/W ...        */
/* A point covers a pixel iff dist2 < radius^2 (strict) and z >= 0.     */
/* Per pixel keep the K smallest by (z, idx) ascending.                 */
/* outputs idx[H,W,K] int64 (-1 pad), zbuf[H,W,K] (-1 pad),             */
/* dist2[H,W,K] (-1 pad).                                               */
/* ------------------------------------------------------------------ */
static inline float pix_to_ndc(int i, int S1, int S2) {
  /* S1 = size along this axis, S2 = the other axis */
  float range = S1 > S2 ? 2.0f * (float)S1 / (float)S2 : 2.0f;
  float offset = range / 2.0f;
  return -offset + (range * (float)i + offset) / (float)S1;
}

/* points (world) -> NDC x,y and view z; ndc[N,3] */
ORC_API void orc_points_to_ndc(const float *cam, int H, int W, const float *pts, int64_t N,
                               int64_t stride, float *ndc) {
  const float *w2c = cam + CAM_W2C;
  const float fx = cam[CAM_K + 0], fy = cam[CAM_K + 5];
  const float cx = cam[CAM_K + 2], cy = cam[CAM_K + 6];
  float s = (float)(W < H ? W : H) / 2.0f;
  float fxn = fx / s, fyn = fy / s;
  float p0x = -(cx - (float)W / 2.0f) / s;
  float p0y = -(cy - (float)H / 2.0f) / s;
#pragma omp parallel for
  for (int64_t i = 0; i < N; ++i) {
    const float *X = pts + i * stride;
    float v[3];
    for (int k = 0; k < 3; ++k) {
      float a = w2c[k * 4 + 0] * X[0];
      a = a + w2c[k * 4 + 1] * X[1];
      a = a + w2c[k * 4 + 2] * X[2];
      a = a + w2c[k * 4 + 3];
      v[k] = a;
    }
    float xv = -v[0], yv = -v[1], zv = v[2];
    ndc[i * 3 + 0] = (fxn * xv + p0x * zv) / zv;
    ndc[i * 3 + 1] = (fyn * yv + p0y * zv) / zv;
    ndc[i * 3 + 2] = zv;
  }
}

static void raster_points_window(const float *ndc, int64_t N, int H, int W, float radius, int K,
                                 int y0, int y1, int x0, int x1, int64_t *idx, float *zbuf,
                                 float *dist2);

ORC_API void orc_raster_points_naive(const float *ndc, int64_t N, int H, int W, float radius,
                                     int K, int64_t *idx, float *zbuf, float *dist2) {
  raster_points_window(ndc, N, H, W, radius, K, 0, H, 0, W, idx, zbuf, dist2);
}

/* The same naive loop restricted to the pixel window [y0,y1) x [x0,x1) of the H x W image:
 * every point is still tested against every pixel of the window, so a 1080p view with millions
 * of points can be checked on a few windows in seconds.  Outputs are [y1-y0, x1-x0, K]. */
ORC_API void orc_raster_points_window(const float *ndc, int64_t N, int H, int W, float radius,
                                      int K, int y0, int y1, int x0, int x1, int64_t *idx,
                                      float *zbuf, float *dist2) {
  raster_points_window(ndc, N, H, W, radius, K, y0, y1, x0, x1, idx, zbuf, dist2);
}

static void raster_points_window(const float *ndc, int64_t N, int H, int W, float radius, int K,
                                 int y0, int y1, int x0, int x1, int64_t *idx, float *zbuf,
                                 float *dist2) {
  const float r2 = radius * radius;
  const int WW = x1 - x0;
#pragma omp parallel
  {
    float *qz = (float *)malloc(sizeof(float) * (size_t)K);
    float *qd = (float *)malloc(sizeof(float) * (size_t)K);
    int64_t *qi = (int64_t *)malloc(sizeof(int64_t) * (size_t)K);
#pragma omp for schedule(dynamic, 1) collapse(2)
    for (int yi = y0; yi < y1; ++yi) {
      for (int xi = x0; xi < x1; ++xi) {
        float yf = pix_to_ndc(H - 1 - yi, H, W);
        float xf = pix_to_ndc(W - 1 - xi, W, H);
        int cnt = 0;
        for (int64_t p = 0; p < N; ++p) {
          float pz = ndc[p * 3 + 2];
          if (pz < 0.0f) continue;
          float dx = ndc[p * 3 + 0] - xf;
          float dy = ndc[p * 3 + 1] - yf;
          float d2 = dx * dx + dy * dy;
          if (!(d2 < r2)) continue;
          /* insert keeping ascending (z, idx); p increases so ties keep order */
          if (cnt < K) {
            int k = cnt++;
            while (k > 0 && qz[k - 1] > pz) {
              qz[k] = qz[k - 1];
              qi[k] = qi[k - 1];
              qd[k] = qd[k - 1];
              --k;
            }
            qz[k] = pz;
            qi[k] = p;
            qd[k] = d2;
          } else if (pz < qz[K - 1]) {
            int k = K - 1;
            while (k > 0 && qz[k - 1] > pz) {
              qz[k] = qz[k - 1];
              qi[k] = qi[k - 1];
              qd[k] = qd[k - 1];
              --k;
            }
            qz[k] = pz;
            qi[k] = p;
            qd[k] = d2;
          }
        }
        size_t o = ((size_t)(yi - y0) * WW + (xi - x0)) * K;
        for (int k = 0; k < K; ++k) {
          if (k < cnt) {
            idx[o + k] = qi[k];
            zbuf[o + k] = qz[k];
            dist2[o + k] = qd[k];
          } else {
            idx[o + k] = -1;
            zbuf[o + k] = -1.0f;
            dist2[o + k] = -1.0f;
          }
        }
      }
    }
    free(qz);
    free(qd);
    free(qi);
  }
}

/* The same per-pixel lists from a POINT-major sweep, for full-size frames (the pixel-major loop above is
 * O(pixels x points): hours at 1080p x 3.5 M points).  Same pix_to_ndc, same disc arithmetic
 * (dx = x - xf; dy = y - yf; d2 = dx*dx + dy*dy; d2 < r2) and the same insertion rule, points visited in
 * index order, so every pixel receives exactly the tests that pass in the naive loop, in the same order:
 * a point is tried on a conservative pixel box (its disc's bounding box in pixel units, widened by two
 * pixels against the rounding of the box itself; the test decides, the box only has to be a superset).
 * Threads own bands of rows.  tests/test_oracle_golden.py proves it equal to the naive loop. */
ORC_API void orc_raster_points_pointmajor(const float *ndc, int64_t N, int H, int W, float radius, int K,
                                          int64_t *idx, float *zbuf, float *dist2) {
  const float r2 = radius * radius;
  const size_t P = (size_t)H * W;
  int32_t *cnt = (int32_t *)calloc(P, sizeof(int32_t));
  float *xf = (float *)malloc(sizeof(float) * (size_t)W);
  float *yf = (float *)malloc(sizeof(float) * (size_t)H);
  for (int xi = 0; xi < W; ++xi) xf[xi] = pix_to_ndc(W - 1 - xi, W, H);
  for (int yi = 0; yi < H; ++yi) yf[yi] = pix_to_ndc(H - 1 - yi, H, W);
  /* pixel index as a real-valued function of the NDC coordinate: ndc = -off + (range*(S-1-i) + off)/S */
  const double rx = W > H ? 2.0 * (double)W / (double)H : 2.0, ry = H > W ? 2.0 * (double)H / (double)W : 2.0;
  const double ox = rx / 2.0, oy = ry / 2.0;
  const double rpx = (double)radius * (double)W / rx + 2.0, rpy = (double)radius * (double)H / ry + 2.0;
#pragma omp parallel
  {
    const int T = omp_get_num_threads(), t = omp_get_thread_num();
    const int ya = (int)((int64_t)H * t / T), yb = (int)((int64_t)H * (t + 1) / T);
    for (int64_t p = 0; p < N && ya < yb; ++p) {
      const float px = ndc[p * 3 + 0], py = ndc[p * 3 + 1], pz = ndc[p * 3 + 2];
      if (pz < 0.0f) continue;
      const double cy = (double)(H - 1) - (((double)py + oy) * (double)H - oy) / ry;
      if (!(cy + rpy >= (double)ya && cy - rpy <= (double)(yb - 1))) continue;
      const double cx = (double)(W - 1) - (((double)px + ox) * (double)W - ox) / rx;
      if (!(cx + rpx >= 0.0 && cx - rpx <= (double)(W - 1))) continue;
      int y0 = (int)floor(cy - rpy), y1 = (int)ceil(cy + rpy);
      int x0 = (int)floor(cx - rpx), x1 = (int)ceil(cx + rpx);
      y0 = y0 < ya ? ya : y0;
      y1 = y1 > yb - 1 ? yb - 1 : y1;
      x0 = x0 < 0 ? 0 : x0;
      x1 = x1 > W - 1 ? W - 1 : x1;
      for (int yi = y0; yi <= y1; ++yi) {
        const float dy = py - yf[yi];
        for (int xi = x0; xi <= x1; ++xi) {
          const float dx = px - xf[xi];
          const float d2 = dx * dx + dy * dy;
          if (!(d2 < r2)) continue;
          const size_t o = ((size_t)yi * W + xi) * K;
          int c = cnt[(size_t)yi * W + xi];
          int k;
          if (c < K) {
            k = c;
            cnt[(size_t)yi * W + xi] = c + 1;
          } else if (pz < zbuf[o + K - 1]) {
            k = K - 1;
          } else {
            continue;
          }
          while (k > 0 && zbuf[o + k - 1] > pz) {
            zbuf[o + k] = zbuf[o + k - 1];
            idx[o + k] = idx[o + k - 1];
            dist2[o + k] = dist2[o + k - 1];
            --k;
          }
          zbuf[o + k] = pz;
          idx[o + k] = p;
          dist2[o + k] = d2;
        }
      }
    }
    for (int yi = ya; yi < yb; ++yi)
      for (int xi = 0; xi < W; ++xi) {
        const size_t o = ((size_t)yi * W + xi) * K;
        for (int k = cnt[(size_t)yi * W + xi]; k < K; ++k) {
          idx[o + k] = -1;
          zbuf[o + k] = -1.0f;
          dist2[o + k] = -1.0f;
        }
      }
  }
  free(cnt);
  free(xf);
  free(yf);
}

/* NormWeightedCompositor: weights = 1 - dist2/(r*r) (points/renderer.py),
 * t = max(sum_k w_k, 1e-4); out[c] = sum_k w_k * feat[idx_k][c] / t
 * (norm_weighted_sum_cpu.cpp).  feat[N, fstride] (first C used); out[H,W,C]. */
ORC_API void orc_norm_weighted_composite(const int64_t *idx, const float *dist2, int H, int W,
                                         int K, float radius, const float *feat,
                                         int64_t fstride, int C, float *out) {
  const float r2 = radius * radius;
#pragma omp parallel for
  for (int64_t p = 0; p < (int64_t)H * W; ++p) {
    float t = 0.0f;
    for (int k = 0; k < K; ++k) {
      if (idx[p * K + k] < 0) continue;
      float w = 1.0f - dist2[p * K + k] / r2;
      t = t + w;
    }
    t = t > 1e-4f ? t : 1e-4f;
    for (int c = 0; c < C; ++c) {
      float acc = 0.0f;
      for (int k = 0; k < K; ++k) {
        int64_t n = idx[p * K + k];
        if (n < 0) continue;
        float w = 1.0f - dist2[p * K + k] / r2;
        float f = feat ? feat[n * fstride + c] : 1.0f;
        acc = acc + w * f / t;
      }
      out[p * C + c] = acc;
    }
  }
}

/* ------------------------------------------------------------------ */
/* A12: static point-cloud aggregation helpers (float64 numpy upstream) */
/* _compute_pcl_proj_mask (pgdvs/datasets/nvidia_eval_pure_geo.py:257-277):*/
/*   verts_cam = w2c @ [X,1]; /= w ; pix = K3 @ verts_cam ; /= z         */
/*   closed bounds 0<=col<=W-1, 0<=row<=H-1 ; astype(int) truncation     */
/*   NO z>0 test, NO epsilon.  pcl is fp32 (rays are FloatTensor), the    */
/*   matrices are float64.                                               */
/* ------------------------------------------------------------------ */
ORC_API void orc_static_proj_mask(const float *pcl, int64_t N, int64_t stride,
                                  const double *K3, const double *w2c, int H, int W,
                                  uint8_t *mask) {
  for (int64_t i = 0; i < N; ++i) {
    const float *X = pcl + i * stride;
    double x = (double)X[0], y = (double)X[1], z = (double)X[2];
    double vc[4];
    for (int k = 0; k < 4; ++k) {
      double s = w2c[k * 4 + 0] * x;
      s = s + w2c[k * 4 + 1] * y;
      s = s + w2c[k * 4 + 2] * z;
      s = s + w2c[k * 4 + 3];
      vc[k] = s;
    }
    double cx = vc[0] / vc[3], cy = vc[1] / vc[3], cz = vc[2] / vc[3];
    double pp[3];
    for (int k = 0; k < 3; ++k) {
      double s = K3[k * 3 + 0] * cx;
      s = s + K3[k * 3 + 1] * cy;
      s = s + K3[k * 3 + 2] * cz;
      pp[k] = s;
    }
    double col = pp[0] / pp[2], row = pp[1] / pp[2];
    if (!(row >= 0.0 && row <= (double)(H - 1))) continue;
    if (!(col >= 0.0 && col <= (double)(W - 1))) continue;
    mask[(int64_t)row * W + (int64_t)col] = 1;
  }
}

/* _compute_pcl (pgdvs/datasets/nvidia_eval.py:840-847) with
 * _get_rays_single_image (pgdvs/datasets/base.py:507-546): K, c2w are cast to
 * fp32 (torch.FloatTensor), rays_d = c2w[:3,:3] @ inv(K) @ [u,v,1] at integer
 * pixel centres, pcl = rays_o + rays_d * depth.  cam = block of orc_cam_prep. */
ORC_API void orc_compute_pcl(const float *cam, int H, int W, const float *depth, float *pcl) {
  const float *M = cam + CAM_M, *o = cam + CAM_O;
#pragma omp parallel for
  for (int r = 0; r < H; ++r) {
    for (int c = 0; c < W; ++c) {
      int p = r * W + c;
      float u = (float)c, v = (float)r;
      for (int k = 0; k < 3; ++k) {
        float d = M[k * 3 + 0] * u;
        d = d + M[k * 3 + 1] * v;
        d = d + M[k * 3 + 2];
        pcl[p * 3 + k] = o[k] + d * depth[p];
      }
    }
  }
}

/* ------------------------------------------------------------------ */
/* A17: tracker-window aggregation                                       */
/* (pgdvs/renderers/pgdvs_renderer_dyn_track.py:98-396).  Tracks and     */
/* visibilities are inputs (the trackers are third-party networks).      */
/* ------------------------------------------------------------------ */

/* mean of the KK smallest squared distances from each query to the base cloud, ALL KK
 * columns (:299-312: no self column to drop); missing columns are zero (pytorch3d pads).
 * Same butterfly summation as orc_knn_mean_dist. */
ORC_API void orc_knn_cross_mean_dist(const float *q, int64_t NQ, const float *pts, int64_t N,
                                     int KK, float *out) {
#pragma omp parallel
  {
    float *best = (float *)malloc(sizeof(float) * (size_t)KK);
#pragma omp for schedule(dynamic, 64)
    for (int64_t i = 0; i < NQ; ++i) {
      int cnt = 0;
      float qx = q[i * 3], qy = q[i * 3 + 1], qz = q[i * 3 + 2];
      for (int64_t j = 0; j < N; ++j) {
        float dx = qx - pts[j * 3], dy = qy - pts[j * 3 + 1], dz = qz - pts[j * 3 + 2];
        float d = dx * dx;
        d = d + dy * dy;
        d = d + dz * dz;
        if (cnt < KK) {
          int k = cnt++;
          while (k > 0 && best[k - 1] > d) {
            best[k] = best[k - 1];
            --k;
          }
          best[k] = d;
        } else if (d < best[KK - 1]) {
          int k = KK - 1;
          while (k > 0 && best[k - 1] > d) {
            best[k] = best[k - 1];
            --k;
          }
          best[k] = d;
        }
      }
      float s = 0.0f;
      if (KK <= 64) {
        float sl[64];
        for (int k = 0; k < 64; ++k) sl[k] = (k < KK && k < cnt) ? best[k] : 0.0f;
        for (int off = 32; off > 0; off >>= 1)
          for (int k = 0; k < off; ++k) sl[k] = sl[k] + sl[k + off];
        s = sl[0];
      } else {
        for (int k = 0; k < KK; ++k) s = s + (k < cnt ? best[k] : 0.0f);
      }
      out[i] = s / (float)KK;
    }
    free(best);
  }
}

/* One track point: validity (:115-127), the two visible frames closest in time to the
 * target (:146-166; ties of |dt| resolved towards the lower frame index -- torch.argsort is
 * not stable, so ties are unpinned), colour (bilinear, align_corners=True) and depth
 * (nearest, align_corners=False) samples at the track position (:197-229), unprojection
 * with c2w[:3,:3] @ inv(K) @ [u,v,1] (:231-253), the mean colour (:271-276) and the linear
 * inter/extrapolation in time (:278-284).
 *   tracks [P,N,2] (col,row); vis [P,N]; kind [N]: 1 = temporally-closest frame, 2 = real
 *   track frame; rgbs [N,H,W,3]; depths [N,H,W]; cams [N,CAM_BLOCK]. */
ORC_API void orc_track_points(int64_t P, int N, int H, int W, const float *tracks,
                              const uint8_t *vis, const uint8_t *kind, const float *times,
                              float time_tgt, const float *rgbs, const float *depths,
                              const float *cams, uint8_t *valid, float *pcl, float *rgb) {
  const float fw = (float)W, fh = (float)H;
#pragma omp parallel for
  for (int64_t p = 0; p < P; ++p) {
    int seen_closest = 0, n_real = 0;
    for (int f = 0; f < N; ++f) {
      if (!vis[p * N + f]) continue;
      if (kind[f] == 1) seen_closest = 1;
      if (kind[f] == 2) ++n_real;
    }
    int ok = !seen_closest && n_real >= 2;
    valid[p] = (uint8_t)ok;
    for (int k = 0; k < 3; ++k) {
      pcl[p * 3 + k] = 0.0f;
      rgb[p * 3 + k] = 0.0f;
    }
    if (!ok) continue;
    int f0 = -1, f1 = -1;
    float d0 = INFINITY, d1 = INFINITY;
    for (int f = 0; f < N; ++f) {
      if (!vis[p * N + f]) continue;
      float d = fabsf(times[f] - time_tgt);
      if (d < d0) {
        f1 = f0;
        d1 = d0;
        f0 = f;
        d0 = d;
      } else if (d < d1) {
        f1 = f;
        d1 = d;
      }
    }
    float X[2][3], col[2][3];
    const int fr[2] = {f0, f1};
    for (int s = 0; s < 2; ++s) {
      const int f = fr[s];
      const float u = tracks[(p * N + f) * 2 + 0], v = tracks[(p * N + f) * 2 + 1];
      const float gx = 2.0f * u / fw - 1.0f, gy = 2.0f * v / fh - 1.0f;
      /* bilinear, align_corners=True: ((g + 1) / 2) * (size - 1) */
      const float ix = ((gx + 1.0f) / 2.0f) * (fw - 1.0f), iy = ((gy + 1.0f) / 2.0f) * (fh - 1.0f);
      const float x0f = floorf(ix), y0f = floorf(iy);
      const int fin = isfinite(ix) && isfinite(iy) && fabsf(ix) < 1e9f && fabsf(iy) < 1e9f;
      const int x0 = fin ? (int)x0f : -10, y0 = fin ? (int)y0f : -10, x1 = x0 + 1, y1 = y0 + 1;
      const float wnw = ((float)x1 - ix) * ((float)y1 - iy), wne = (ix - (float)x0) * ((float)y1 - iy);
      const float wsw = ((float)x1 - ix) * (iy - (float)y0), wse = (ix - (float)x0) * (iy - (float)y0);
      const float *img = rgbs + (size_t)f * H * W * 3;
      for (int k = 0; k < 3; ++k) {
        float acc = 0.0f;
        if (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H) acc = acc + img[(y0 * W + x0) * 3 + k] * wnw;
        if (x1 >= 0 && x1 < W && y0 >= 0 && y0 < H) acc = acc + img[(y0 * W + x1) * 3 + k] * wne;
        if (x0 >= 0 && x0 < W && y1 >= 0 && y1 < H) acc = acc + img[(y1 * W + x0) * 3 + k] * wsw;
        if (x1 >= 0 && x1 < W && y1 >= 0 && y1 < H) acc = acc + img[(y1 * W + x1) * 3 + k] * wse;
        col[s][k] = acc;
      }
      /* nearest, align_corners=False: ((g + 1) * size - 1) / 2, round-half-even */
      const float nx = nearbyintf(((gx + 1.0f) * fw - 1.0f) / 2.0f), ny = nearbyintf(((gy + 1.0f) * fh - 1.0f) / 2.0f);
      float dsamp = 0.0f;
      if (nx >= 0.0f && nx <= fw - 1.0f && ny >= 0.0f && ny <= fh - 1.0f)
        dsamp = depths[(size_t)f * H * W + (int)ny * W + (int)nx];
      const float *M = cams + (size_t)f * CAM_BLOCK + CAM_M, *o = cams + (size_t)f * CAM_BLOCK + CAM_O;
      for (int k = 0; k < 3; ++k) {
        float d = M[k * 3 + 0] * u;
        d = d + M[k * 3 + 1] * v;
        d = d + M[k * 3 + 2];
        X[s][k] = o[k] + d * dsamp;
      }
    }
    const float t0 = times[f0], t1 = times[f1];
    const float ratio = (time_tgt - t0) / ((t1 - t0) + 1e-8f);
    for (int k = 0; k < 3; ++k) {
      pcl[p * 3 + k] = X[0][k] + (X[1][k] - X[0][k]) * ratio;
      rgb[p * 3 + k] = (col[0][k] + col[1][k]) / 2.0f;
    }
  }
}

/* ------------------------------------------------------------------ */
/* A10: dyn_render_type = "mesh" (pgdvs_renderer_dyn.py:542-669)        */
/* Topology (:550-604): each kept source pixel (r,c) spawns the          */
/* triangles {(r,c),(r+1,c),(r+1,c+1)} and {(r,c),(r+1,c+1),(r,c+1)};    */
/* a face survives when its three corners are in bounds and carry a      */
/* vertex index > 0 (sic, :597: the first kept pixel, index 0, is        */
/* treated as "no vertex").  Face order: all first-kind faces, then all  */
/* second-kind ones (:579-581), both in raster order of (r,c).           */
/* Rasteriser: pytorch3d 0.7.4 MeshRasterizer with blur_radius=0,        */
/* faces_per_pixel=1, bin_size=0 (naive path), perspective-correct       */
/* barycentrics (PerspectiveCameras), no z clipping (znear is None),     */
/* no back-face culling; shader = TexturesVertex interpolation +         */
/* hard_rgb_blend on black (pgdvs/utils/pytorch3d_utils.py:50-67).       */
/* pytorch3d is not installable here: parity UNPINNED, restated from     */
/* pytorch3d/csrc/rasterize_meshes/rasterize_meshes.cu                   */
/* (CheckPixelInsideFace) and csrc/utils/geometry_utils.cuh.             */
/*   keep[P] u8, pcl[P,3], rgb[P,3] dense over the source frame;         */
/*   out: img[H,W,3], mask[H,W], face[H,W] (kind*P + pixel, -1 = none).  */
/* ------------------------------------------------------------------ */
#define MESH_EPS 1e-8f

static inline float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

/* returns 1 and fills pz / bary when pixel centre (px,py) is strictly inside the face */
static inline int mesh_pixel_in_face(float px, float py, const float *v0, const float *v1,
                                     const float *v2, float *pz, float *bary) {
  float zmax = fmaxf(fmaxf(v0[2], v1[2]), v2[2]);
  float xmin = fminf(fminf(v0[0], v1[0]), v2[0]), xmax = fmaxf(fmaxf(v0[0], v1[0]), v2[0]);
  float ymin = fminf(fminf(v0[1], v1[1]), v2[1]), ymax = fmaxf(fmaxf(v0[1], v1[1]), v2[1]);
  int outside = (px > xmax) || (px < xmin) || (py > ymax) || (py < ymin);
  float face_area = edge_fn(v0[0], v0[1], v1[0], v1[1], v2[0], v2[1]);
  int zero_area = (face_area <= MESH_EPS) && (face_area >= -MESH_EPS);
  if (zmax < 0.0f || outside || zero_area) return 0;
  float area = edge_fn(v2[0], v2[1], v0[0], v0[1], v1[0], v1[1]) + MESH_EPS;
  float b0 = edge_fn(px, py, v1[0], v1[1], v2[0], v2[1]) / area;
  float b1 = edge_fn(px, py, v2[0], v2[1], v0[0], v0[1]) / area;
  float b2 = edge_fn(px, py, v0[0], v0[1], v1[0], v1[1]) / area;
  float t0 = b0 * v1[2] * v2[2];
  float t1 = v0[2] * b1 * v2[2];
  float t2 = v0[2] * v1[2] * b2;
  float den = fmaxf(t0 + t1 + t2, MESH_EPS);
  bary[0] = t0 / den;
  bary[1] = t1 / den;
  bary[2] = t2 / den;
  float z = bary[0] * v0[2] + bary[1] * v1[2] + bary[2] * v2[2];
  if (z < 0.0f) return 0;
  if (!(bary[0] > 0.0f && bary[1] > 0.0f && bary[2] > 0.0f)) return 0;
  *pz = z;
  return 1;
}

/* candidate pixel index range along one axis for NDC interval [lo,hi] (superset) */
static inline void ndc_to_pix_range(float lo, float hi, int S1, int S2, int *i0, int *i1) {
  float range = S1 > S2 ? 2.0f * (float)S1 / (float)S2 : 2.0f;
  float offset = range / 2.0f;
  /* ndc(i') = -offset + (range*i' + offset)/S1 with i' = S1-1-i */
  float a = ((lo + offset) * (float)S1 - offset) / range;
  float b = ((hi + offset) * (float)S1 - offset) / range;
  if (!(a >= -2.0f)) a = -2.0f;
  if (!(b <= (float)S1 + 1.0f)) b = (float)S1 + 1.0f;
  int ja = (int)floorf(a) - 1, jb = (int)ceilf(b) + 1;
  if (ja < 0) ja = 0;
  if (jb > S1 - 1) jb = S1 - 1;
  *i0 = S1 - 1 - jb;
  *i1 = S1 - 1 - ja;
}

ORC_API void orc_mesh_render(const float *cam, int H, int W, const uint8_t *keep,
                             const float *pcl, const float *rgb, float *img, float *mask,
                             int64_t *face) {
  const int64_t P = (int64_t)H * W;
  float *ndc = (float *)malloc(sizeof(float) * 3 * (size_t)P);
  float *zb = (float *)malloc(sizeof(float) * (size_t)P);
  orc_points_to_ndc(cam, H, W, pcl, P, 3, ndc);
  int64_t first = -1;
  for (int64_t p = 0; p < P; ++p)
    if (keep[p]) {
      first = p;
      break;
    }
  for (int64_t p = 0; p < P; ++p) {
    face[p] = -1;
    zb[p] = 0.0f;
    mask[p] = 0.0f;
    img[p * 3] = img[p * 3 + 1] = img[p * 3 + 2] = 0.0f;
  }
  for (int kind = 0; kind < 2; ++kind) {
    for (int r = 0; r + 1 < H; ++r) {
      for (int c = 0; c + 1 < W; ++c) {
        int64_t q0 = (int64_t)r * W + c;
        int64_t q1 = kind == 0 ? q0 + W : q0 + W + 1;
        int64_t q2 = kind == 0 ? q0 + W + 1 : q0 + 1;
        if (!(keep[q0] && keep[q1] && keep[q2])) continue;
        if (q0 == first || q1 == first || q2 == first) continue; /* vertex index 0 (:597) */
        const float *v0 = ndc + q0 * 3, *v1 = ndc + q1 * 3, *v2 = ndc + q2 * 3;
        float xmin = fminf(fminf(v0[0], v1[0]), v2[0]), xmax = fmaxf(fmaxf(v0[0], v1[0]), v2[0]);
        float ymin = fminf(fminf(v0[1], v1[1]), v2[1]), ymax = fmaxf(fmaxf(v0[1], v1[1]), v2[1]);
        if (!(xmin <= xmax && ymin <= ymax)) continue; /* NaN vertex: never inside */
        int x0, x1, y0, y1;
        ndc_to_pix_range(xmin, xmax, W, H, &x0, &x1);
        ndc_to_pix_range(ymin, ymax, H, W, &y0, &y1);
        int64_t fid = (int64_t)kind * P + q0;
        for (int yi = y0; yi <= y1; ++yi) {
          float yf = pix_to_ndc(H - 1 - yi, H, W);
          for (int xi = x0; xi <= x1; ++xi) {
            float xf = pix_to_ndc(W - 1 - xi, W, H);
            float pz, b[3];
            if (!mesh_pixel_in_face(xf, yf, v0, v1, v2, &pz, b)) continue;
            int64_t t = (int64_t)yi * W + xi;
            if (face[t] >= 0 && !(pz < zb[t] || (pz == zb[t] && fid < face[t]))) continue;
            face[t] = fid;
            zb[t] = pz;
            for (int k = 0; k < 3; ++k) {
              float a = b[0] * rgb[q0 * 3 + k];
              a = a + b[1] * rgb[q1 * 3 + k];
              a = a + b[2] * rgb[q2 * 3 + k];
              img[t * 3 + k] = a;
            }
            mask[t] = 1.0f;
          }
        }
      }
    }
  }
  free(ndc);
  free(zb);
}

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
