// Dynamic-branch kernels: camera prep, rays, unproject + flow warp, projection to
// the target view, softsplat metric.  One thread per pixel, coalesced streaming
// loads; all of these are HBM-bound byte movers (see DESIGN.md for bytes/pixel).
#include <string.h>

#include "common.h"
#include "fused.h"

namespace pgdvs {

// ---------------------------------------------------------------------------
// Gauss-Jordan inverse of an N x N matrix (N <= 4) with partial pivoting, one COLUMN of the augmented
// matrix [A | I] per lane (lanes 0 .. 2N-1 of a wavefront; pivot values and multipliers are broadcast).
// Every element goes through exactly the operations of the serial inv_f64 (common.h, = the oracle's), in
// the same order, so the results agree bit for bit; a single thread walking the 4 x 8 array needed ~7 us for
// the two inverses of a camera, mostly dependent fp64 divisions.  Returns false for a singular matrix.
template <int N>
__device__ __forceinline__ bool inv_f64_wave(double (&m)[N], int lane) {
  // m[r] = element (r, lane) of the augmented matrix
  bool ok = true;
#pragma unroll
  for (int c = 0; c < N; ++c) {
    // column c lives on lane c: pivot row = first row >= c with the largest magnitude
    double colv[N];
#pragma unroll
    for (int r = 0; r < N; ++r) colv[r] = __shfl(m[r], c, 64);
    int piv = c;
    double best = fabs(colv[c]);
#pragma unroll
    for (int r = c + 1; r < N; ++r) {
      const double v = fabs(colv[r]);
      if (v > best) {
        best = v;
        piv = r;
      }
    }
    if (best == 0.0) ok = false;
    // swap rows c and piv (uniform over the wavefront)
#pragma unroll
    for (int r = c + 1; r < N; ++r)
      if (r == piv) {
        const double t = m[c];
        m[c] = m[r];
        m[r] = t;
        const double tc = colv[c];
        colv[c] = colv[r];
        colv[r] = tc;
      }
    const double d = colv[c];
    m[c] = m[c] / d;
#pragma unroll
    for (int r = 0; r < N; ++r) {
      if (r == c) continue;
      const double f = colv[r];
      if (f == 0.0) continue;
      m[r] = m[r] - f * m[c];
    }
  }
  return ok;
}

// one wavefront per camera: flat_cam[34] -> camera block (same values as cam_block_from_flat)
__device__ __forceinline__ void cam_prep_one(const float *__restrict__ fc, float *__restrict__ blk, const int lane) {
  const float *K = fc + 2, *c2w = fc + 18;
  __shared__ float s_kinv[9], s_w2c[16];
  // K[:3,:3]^-1: lanes 0..5 hold the columns of [K3 | I]
  double a3[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) a3[r] = lane < 3 ? (double)K[r * 4 + lane] : (lane - 3 == r ? 1.0 : 0.0);
  const bool ok3 = inv_f64_wave<3>(a3, lane);
  // c2w^-1: lanes 0..7 hold the columns of [c2w | I]
  double a4[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) a4[r] = lane < 4 ? (double)c2w[r * 4 + (lane & 3)] : (lane - 4 == r ? 1.0 : 0.0);
  const bool ok4 = inv_f64_wave<4>(a4, lane);
  const bool bad = !(ok3 && ok4);
  if (lane >= 3 && lane < 6)
#pragma unroll
    for (int r = 0; r < 3; ++r) s_kinv[r * 3 + (lane - 3)] = bad ? __builtin_nanf("") : (float)a3[r];
  if (lane >= 4 && lane < 8)
#pragma unroll
    for (int r = 0; r < 4; ++r) s_w2c[r * 4 + (lane - 4)] = bad ? __builtin_nanf("") : (float)a4[r];
  __syncthreads();
  // derived matrices: fp32 products in a fixed left-to-right order (no FMA contraction in this library)
  if (lane < 9) {
    const int r = lane / 3, c = lane - r * 3;
    blk[PGDVS_CAM_KINV + lane] = s_kinv[lane];
    float v = c2w[r * 4 + 0] * s_kinv[0 * 3 + c];
    v = v + c2w[r * 4 + 1] * s_kinv[1 * 3 + c];
    v = v + c2w[r * 4 + 2] * s_kinv[2 * 3 + c];
    blk[PGDVS_CAM_M + lane] = v;
    blk[PGDVS_CAM_R + lane] = c2w[r * 4 + c];
  }
  if (lane < 3) blk[PGDVS_CAM_O + lane] = c2w[lane * 4 + 3];
  if (lane < 16) {
    const int r = lane >> 2, c = lane & 3;
    float v = K[r * 4 + 0] * s_w2c[0 * 4 + c];
    v = v + K[r * 4 + 1] * s_w2c[1 * 4 + c];
    v = v + K[r * 4 + 2] * s_w2c[2 * 4 + c];
    v = v + K[r * 4 + 3] * s_w2c[3 * 4 + c];
    blk[PGDVS_CAM_P + lane] = v;
    blk[PGDVS_CAM_W2C + lane] = s_w2c[lane];
    blk[PGDVS_CAM_K + lane] = K[lane];
  }
  if (lane < 2) blk[PGDVS_CAM_HW + lane] = fc[lane];
}

__global__ void __launch_bounds__(64) cam_prep_kernel(const float *__restrict__ flat_cams, int n,
                                                      float *__restrict__ blocks) {
  const int i = blockIdx.x;
  if (i >= n) return;
  cam_prep_one(flat_cams + (size_t)i * 34, blocks + (size_t)i * PGDVS_CAM_BLOCK, threadIdx.x);
}

// The per-view native call (view_geo.cpp): the camera blocks of the target and of the two temporal source frames
// (blocks[0] = target, blocks[1..2] = sources) and the packed time stamps (t1, t2, t_tgt) in ONE launch instead of two
// cam_prep launches and a torch.cat.
__global__ void __launch_bounds__(64) view_prep_kernel(const float *__restrict__ flat_tgt, const float *__restrict__ flat_src,
                                                       const float *__restrict__ time_src, const float *__restrict__ time_tgt,
                                                       float *__restrict__ blocks, float *__restrict__ times) {
  const int i = blockIdx.x;
  if (i == 0 && threadIdx.x < 3) times[threadIdx.x] = threadIdx.x < 2 ? time_src[threadIdx.x] : time_tgt[0];
  cam_prep_one(i == 0 ? flat_tgt : flat_src + (size_t)(i - 1) * 34, blocks + (size_t)i * PGDVS_CAM_BLOCK, threadIdx.x);
}

// A1 -- pgdvs_renderer_base.py:17-57
__global__ void get_rays_kernel(const float *__restrict__ cam, int rh, int rw, int stride,
                                float *__restrict__ rays_o, float *__restrict__ rays_d,
                                float *__restrict__ uvs) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rh * rw) return;
  int r = i / rw, c = i - r * rw;
  float u = (float)(c * stride), v = (float)(r * stride);
  const float *M = cam + PGDVS_CAM_M;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float d = M[k * 3 + 0] * u;
    d = d + M[k * 3 + 1] * v;
    d = d + M[k * 3 + 2];
    rays_d[(size_t)i * 3 + k] = d;
    rays_o[(size_t)i * 3 + k] = cam[PGDVS_CAM_O + k];
  }
  uvs[(size_t)i * 2 + 0] = u;
  uvs[(size_t)i * 2 + 1] = v;
}

// A2 + A3 -- pgdvs_renderer_dyn.py:299-388
// (one pixel; returns its validity -- what valid[p] receives)
__device__ __forceinline__ bool
dyn_warp_pixel(const int p, int H, int W, const float *__restrict__ dyn_mask1, const float *__restrict__ occ,
               int use_fc, const float *__restrict__ flow12, const float *__restrict__ depth1,
               const float *__restrict__ depth2, const float *__restrict__ rgb1,
               const float *__restrict__ rgb2, const float *__restrict__ cam1,
               const float *__restrict__ cam2, const float *__restrict__ times,
               uint8_t *__restrict__ mask_eff, uint8_t *__restrict__ valid,
               float *__restrict__ pcl, float *__restrict__ rgbf) {
  int r = p / W, c = p - r * W;
  float u = (float)c, v = (float)r;
  const float fw = (float)W, fh = (float)H;
  bool m = dyn_mask1[p] != 0.0f;
  if (use_fc) m = m && !(occ[p] > 0.0f);
  if (mask_eff != nullptr) mask_eff[p] = (uint8_t)m;  // (null in the per-view call: nothing reads it there)
  if (!m) {
    valid[p] = 0;
    return false;
  }
  float2 fl = reinterpret_cast<const float2 *>(flow12)[p];
  float ux = u + fl.x, uy = v + fl.y;
  bool ok = (ux >= 0.0f) && (ux <= fw - 1.0f) && (uy >= 0.0f) && (uy <= fh - 1.0f);
  valid[p] = (uint8_t)ok;
  if (!ok) return false;
  const float t1 = times[0], t2 = times[1], tt = times[2];
  const float *M1 = cam1 + PGDVS_CAM_M;
  float d1 = depth1[p];
  float X1[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float d = M1[k * 3 + 0] * u;
    d = d + M1[k * 3 + 1] * v;
    d = d + M1[k * 3 + 2];
    X1[k] = cam1[PGDVS_CAM_O + k] + d * d1;
  }
  if (t1 == t2) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      pcl[(size_t)p * 3 + k] = X1[k];
      if (rgbf != nullptr) rgbf[(size_t)p * 3 + k] = rgb1[(size_t)p * 3 + k];
    }
    return true;
  }
  float w1 = (t2 - tt) / (t2 - t1);
  float w2 = (tt - t1) / (t2 - t1);
  // grid = 2*uv/(W,H) - 1, grid_sample(align_corners=False): ((g+1)*size-1)/2
  float gx = 2.0f * ux / fw - 1.0f;
  float gy = 2.0f * uy / fh - 1.0f;
  float ix = ((gx + 1.0f) * fw - 1.0f) / 2.0f;
  float iy = ((gy + 1.0f) * fh - 1.0f) / 2.0f;
  float nx = nearbyintf(ix), ny = nearbyintf(iy);
  float dsamp = 0.0f;
  if (nx >= 0.0f && nx <= fw - 1.0f && ny >= 0.0f && ny <= fh - 1.0f)
    dsamp = depth2[(int)ny * W + (int)nx];
  float x0f = floorf(ix), y0f = floorf(iy);
  int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
  float wnw = ((float)x1 - ix) * ((float)y1 - iy);
  float wne = (ix - (float)x0) * ((float)y1 - iy);
  float wsw = ((float)x1 - ix) * (iy - (float)y0);
  float wse = (ix - (float)x0) * (iy - (float)y0);
  bool inx0 = x0 >= 0 && x0 < W, inx1 = x1 >= 0 && x1 < W;
  bool iny0 = y0 >= 0 && y0 < H, iny1 = y1 >= 0 && y1 < H;
  // (rgbf == nullptr -- the per-view call with the softsplat renderer, round 6: the colours sampled from frame 2 only feed the
  // mesh / point / tracker variants (pgdvs_renderer_dyn.py:129,177-190: the splat path splats frame-1 pixels), so neither the
  // four gathers per valid pixel nor the 12-byte row are spent on them there)
  float col[3] = {0.0f, 0.0f, 0.0f};
  if (rgbf != nullptr) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float acc = 0.0f;
      if (inx0 && iny0) acc = acc + rgb2[((size_t)y0 * W + x0) * 3 + k] * wnw;
      if (inx1 && iny0) acc = acc + rgb2[((size_t)y0 * W + x1) * 3 + k] * wne;
      if (inx0 && iny1) acc = acc + rgb2[((size_t)y1 * W + x0) * 3 + k] * wsw;
      if (inx1 && iny1) acc = acc + rgb2[((size_t)y1 * W + x1) * 3 + k] * wse;
      col[k] = acc;
    }
  }
  const float *Kinv2 = cam2 + PGDVS_CAM_KINV;
  const float *R2 = cam2 + PGDVS_CAM_R;
  float kq[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float s = Kinv2[k * 3 + 0] * ux;
    s = s + Kinv2[k * 3 + 1] * uy;
    s = s + Kinv2[k * 3 + 2];
    kq[k] = s;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float d = R2[k * 3 + 0] * kq[0];
    d = d + R2[k * 3 + 1] * kq[1];
    d = d + R2[k * 3 + 2] * kq[2];
    float X2 = cam2[PGDVS_CAM_O + k] + d * dsamp;
    pcl[(size_t)p * 3 + k] = w1 * X1[k] + w2 * X2;
    if (rgbf != nullptr) rgbf[(size_t)p * 3 + k] = col[k];
  }
  return true;
}

// What the per-view native call (view_geo.cpp) lets this launch do on the side, so that the kernels behind it need no
// launches of their own for it (round 6: a view alone is a chain of launches at ~4.7 us each, whatever they do): two byte maps
// cleared pixel by pixel (the keep map of the outlier filter, the flag map of the splat), two small state blocks cleared by
// the first threads (the kNN grid's bounding box / counters / look-back words, the statistics' histograms), and the number of
// valid pixels per 256-pixel chunk (= per workgroup), from which the compaction that follows takes its offsets: WarpExtras,
// fused.h.
__global__ void __launch_bounds__(256)
dyn_warp_kernel(int H, int W, const float *__restrict__ dyn_mask1, const float *__restrict__ occ,
                int use_fc, const float *__restrict__ flow12, const float *__restrict__ depth1,
                const float *__restrict__ depth2, const float *__restrict__ rgb1,
                const float *__restrict__ rgb2, const float *__restrict__ cam1,
                const float *__restrict__ cam2, const float *__restrict__ times,
                uint8_t *__restrict__ mask_eff, uint8_t *__restrict__ valid,
                float *__restrict__ pcl, float *__restrict__ rgbf, const WarpExtras ex) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = p < H * W;
  bool ok = false;
  if (in)
    ok = dyn_warp_pixel(p, H, W, dyn_mask1, occ, use_fc, flow12, depth1, depth2, rgb1, rgb2, cam1, cam2, times, mask_eff, valid,
                        pcl, rgbf);
  if (in && ex.zero_a) ex.zero_a[p] = 0;
  if (in && ex.zero_b) ex.zero_b[p] = 0;
  const int stride = gridDim.x * blockDim.x;
  if (ex.zero0)
    for (int g = p; g < ex.n16_0; g += stride) ex.zero0[g] = make_uint4(0u, 0u, 0u, 0u);
  if (ex.zero1)
    for (int g = p; g < ex.n16_1; g += stride) ex.zero1[g] = make_uint4(0u, 0u, 0u, 0u);
  if (ex.zero2)
    for (int g = p; g < ex.n16_2; g += stride) ex.zero2[g] = make_uint4(0u, 0u, 0u, 0u);
  if (ex.zero3)
    for (int g = p; g < ex.n16_3; g += stride) ex.zero3[g] = make_uint4(0u, 0u, 0u, 0u);
  if (ex.chunk_cnt) {  // (uniform over the launch: every thread reaches the barrier)
    __shared__ int s_c[4];
    const unsigned long long b = __ballot(ok);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = (int)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) ex.chunk_cnt[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
  }
}


// A5 dense -- pgdvs_renderer_dyn.py:470-503, planar flow output
__global__ void __launch_bounds__(256)
project_flow_dense_kernel(int H, int W, const float *__restrict__ cam_tgt,
                          const float *__restrict__ pcl, const uint8_t *__restrict__ keep,
                          float *__restrict__ flow_1_to_tgt, float *__restrict__ valid_mask) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  int P = H * W;
  if (p >= P) return;
  float fx = 0.0f, fy = 0.0f, vm = 0.0f;
  if (keep[p]) {
    int r = p / W, c = p - r * W;
    float u, v;
    project_point(cam_tgt + PGDVS_CAM_P, pcl[(size_t)p * 3], pcl[(size_t)p * 3 + 1],
                  pcl[(size_t)p * 3 + 2], u, v);
    fx = u - (float)c;
    fy = v - (float)r;
    vm = 1.0f;
  }
  flow_1_to_tgt[p] = fx;
  flow_1_to_tgt[(size_t)P + p] = fy;
  valid_mask[p] = vm;
}

__global__ void project_points_kernel(const float *__restrict__ cam_tgt,
                                      const float *__restrict__ pts, int64_t n,
                                      float *__restrict__ uv) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float u, v;
  project_point(cam_tgt + PGDVS_CAM_P, pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2], u, v);
  uv[i * 2] = u;
  uv[i * 2 + 1] = v;
}

// torch.linspace(-1, 1, steps)[i] as ATen computes it
__device__ __forceinline__ float linspace_m1_1(int i, int steps) {
  if (steps == 1) return -1.0f;
  float step = (1.0f - (-1.0f)) / (float)(steps - 1);
  if (i < steps / 2) return -1.0f + step * (float)i;
  return 1.0f - step * (float)(steps - 1 - i);
}

// A6 -- pgdvs_renderer_base.py:68-78,91-138 on NCHW planes
__global__ void __launch_bounds__(256)
backwarp_l1_kernel(const float *__restrict__ rgb1, const float *__restrict__ rgb2,
                   const float *__restrict__ flow, float *__restrict__ l1, int H, int W) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  int P = H * W;
  if (p >= P) return;
  int b = blockIdx.y;
  rgb1 += (size_t)b * 3 * P;
  rgb2 += (size_t)b * 3 * P;
  flow += (size_t)b * 2 * P;
  int r = p / W, c = p - r * W;
  const float hw_x = ((float)W - 1.0f) / 2.0f, hw_y = ((float)H - 1.0f) / 2.0f;
  float gx = linspace_m1_1(c, W) + flow[p] / hw_x;
  float gy = linspace_m1_1(r, H) + flow[(size_t)P + p] / hw_y;
  float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
  float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
  float x0f = floorf(ix), y0f = floorf(iy);
  bool fin = isfinite(ix) && isfinite(iy) && fabsf(ix) < 1e9f && fabsf(iy) < 1e9f;
  int x0 = fin ? (int)x0f : -10, y0 = fin ? (int)y0f : -10, x1 = x0 + 1, y1 = y0 + 1;
  float wnw = ((float)x1 - ix) * ((float)y1 - iy);
  float wne = (ix - (float)x0) * ((float)y1 - iy);
  float wsw = ((float)x1 - ix) * (iy - (float)y0);
  float wse = (ix - (float)x0) * (iy - (float)y0);
  bool inx0 = x0 >= 0 && x0 < W, inx1 = x1 >= 0 && x1 < W;
  bool iny0 = y0 >= 0 && y0 < H, iny1 = y1 >= 0 && y1 < H;
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float *pl = rgb2 + (size_t)k * P;
    float acc = 0.0f;
    if (inx0 && iny0) acc = acc + pl[y0 * W + x0] * wnw;
    if (inx1 && iny0) acc = acc + pl[y0 * W + x1] * wne;
    if (inx0 && iny1) acc = acc + pl[y1 * W + x0] * wsw;
    if (inx1 && iny1) acc = acc + pl[y1 * W + x1] * wse;
    s = s + fabsf(rgb1[(size_t)k * P + p] - acc);
  }
  l1[(size_t)b * P + p] = s / 3.0f;
}

// A11 -- pgdvs_renderer.py:169-178
__global__ void combine_kernel(const float *__restrict__ st, const float *__restrict__ dy,
                               const float *__restrict__ m, int64_t n, float *__restrict__ comb,
                               float *__restrict__ comb_st, float *__restrict__ comb_dy) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float mm = m[i];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float a = (1.0f - mm) * st[(size_t)k * n + i];
    float b = mm * dy[(size_t)k * n + i];
    if (comb_st) comb_st[(size_t)k * n + i] = a;
    if (comb_dy) comb_dy[(size_t)k * n + i] = b;
    comb[(size_t)k * n + i] = a + b;
  }
}

__global__ void gather_rows_kernel(const float *__restrict__ src, const int32_t *__restrict__ idx,
                                   const int32_t *__restrict__ count, int width,
                                   float *__restrict__ dst) {
  int64_t n = *count;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t s = idx[i];
    for (int k = 0; k < width; ++k) dst[i * width + k] = src[s * width + k];
  }
}

__global__ void scatter_keep_kernel(const int32_t *__restrict__ idx,
                                    const uint8_t *__restrict__ flag,
                                    const int32_t *__restrict__ count,
                                    uint8_t *__restrict__ keep) {
  int64_t n = *count;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (flag[i]) keep[idx[i]] = 1;
  }
}

}  // namespace pgdvs

using namespace pgdvs;

PGDVS_API int pgdvs_cam_prep(const float *flat_cams, int n, float *cam_blocks,
                             pgdvs_stream_t stream) {
  PGDVS_REQUIRE(flat_cams && cam_blocks && n >= 0, "pgdvs_cam_prep: bad arguments");
  if (n == 0) return PGDVS_OK;
  PGDVS_LAUNCH("cam_prep", cam_prep_kernel, dim3((unsigned)n), dim3(64), 0, as_stream(stream),
                     flat_cams, n, cam_blocks);
  return check_launch("cam_prep");
}

namespace pgdvs {
int view_prep(const float *flat_tgt, const float *flat_src, const float *time_src, const float *time_tgt, float *blocks,
              float *times, hipStream_t st) {
  PGDVS_LAUNCH("view_prep", view_prep_kernel, dim3(3), dim3(64), 0, st, flat_tgt, flat_src, time_src, time_tgt, blocks, times);
  return check_launch("view_prep");
}
}  // namespace pgdvs

PGDVS_API int pgdvs_get_rays(const float *cam_block, int H, int W, int stride, float *rays_o,
                             float *rays_d, float *uvs, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(cam_block && rays_o && rays_d && uvs && H > 0 && W > 0 && stride > 0,
                "pgdvs_get_rays: bad arguments");
  int rh = (H + stride - 1) / stride, rw = (W + stride - 1) / stride;
  PGDVS_LAUNCH("get_rays", get_rays_kernel, dim3((unsigned)cdiv((int64_t)rh * rw, 256)), dim3(256), 0,
                     as_stream(stream), cam_block, rh, rw, stride, rays_o, rays_d, uvs);
  return check_launch("get_rays");
}

PGDVS_API int pgdvs_dyn_warp(int H, int W, const float *dyn_mask1, const float *occ,
                             int use_flow_consistency, const float *flow12, const float *depth1,
                             const float *depth2, const float *rgb1, const float *rgb2,
                             const float *cam1, const float *cam2, const float *times,
                             uint8_t *mask_eff, uint8_t *valid, float *pcl, float *rgbf,
                             pgdvs_stream_t stream) {
  PGDVS_REQUIRE(H > 0 && W > 0 && (int64_t)H * W < (1ll << 31), "pgdvs_dyn_warp: bad H/W");
  PGDVS_REQUIRE(dyn_mask1 && flow12 && depth1 && depth2 && rgb1 && rgb2 && cam1 && cam2 &&
                    times && mask_eff && valid && pcl && rgbf,
                "pgdvs_dyn_warp: null pointer");
  PGDVS_REQUIRE(!use_flow_consistency || occ, "pgdvs_dyn_warp: occ mask required");
  WarpExtras none;
  memset(&none, 0, sizeof(none));
  return dyn_warp_fused(H, W, dyn_mask1, occ, use_flow_consistency, flow12, depth1, depth2, rgb1, rgb2, cam1, cam2, times,
                        mask_eff, valid, pcl, rgbf, none, as_stream(stream));
}

namespace pgdvs {
int dyn_warp_fused(int H, int W, const float *dyn_mask1, const float *occ, int use_flow_consistency, const float *flow12,
                   const float *depth1, const float *depth2, const float *rgb1, const float *rgb2, const float *cam1,
                   const float *cam2, const float *times, uint8_t *mask_eff, uint8_t *valid, float *pcl, float *rgbf,
                   const WarpExtras &ex, hipStream_t st) {
  PGDVS_LAUNCH("dyn_warp", dyn_warp_kernel, dim3((unsigned)cdiv((int64_t)H * W, 256)), dim3(256), 0, st, H, W, dyn_mask1, occ,
               use_flow_consistency, flow12, depth1, depth2, rgb1, rgb2, cam1, cam2, times, mask_eff, valid, pcl, rgbf, ex);
  return check_launch("dyn_warp");
}
}  // namespace pgdvs

PGDVS_API int pgdvs_project_flow_dense(int H, int W, const float *cam_tgt, const float *pcl,
                                       const uint8_t *keep, float *flow_1_to_tgt,
                                       float *valid_dyn_mask_1, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(H > 0 && W > 0 && cam_tgt && pcl && keep && flow_1_to_tgt && valid_dyn_mask_1,
                "pgdvs_project_flow_dense: bad arguments");
  PGDVS_LAUNCH("project_flow_dense", project_flow_dense_kernel, dim3((unsigned)cdiv((int64_t)H * W, 256)),
                     dim3(256), 0, as_stream(stream), H, W, cam_tgt, pcl, keep, flow_1_to_tgt,
                     valid_dyn_mask_1);
  return check_launch("project_flow_dense");
}

PGDVS_API int pgdvs_project_points(const float *cam_tgt, const float *pts, int64_t n, float *uv,
                                   pgdvs_stream_t stream) {
  PGDVS_REQUIRE(cam_tgt && n >= 0 && (n == 0 || (pts && uv)), "pgdvs_project_points: bad arguments");
  if (n == 0) return PGDVS_OK;
  PGDVS_LAUNCH("project_points", project_points_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0,
                     as_stream(stream), cam_tgt, pts, n, uv);
  return check_launch("project_points");
}

PGDVS_API int pgdvs_backwarp_l1(const float *rgb1, const float *rgb2, const float *flow, float *l1,
                                int B, int H, int W, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(rgb1 && rgb2 && flow && l1 && B > 0 && H > 0 && W > 0,
                "pgdvs_backwarp_l1: bad arguments");
  PGDVS_LAUNCH("backwarp_l1", backwarp_l1_kernel, dim3((unsigned)cdiv((int64_t)H * W, 256), B), dim3(256),
                     0, as_stream(stream), rgb1, rgb2, flow, l1, H, W);
  return check_launch("backwarp_l1");
}

PGDVS_API int pgdvs_combine(const float *static_rgb, const float *dyn_rgb, const float *dyn_mask,
                            int64_t n, float *combined, float *combined_static,
                            float *combined_dyn, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(static_rgb && dyn_rgb && dyn_mask && combined && n >= 0,
                "pgdvs_combine: bad arguments");
  if (n == 0) return PGDVS_OK;
  PGDVS_LAUNCH("combine", combine_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream),
                     static_rgb, dyn_rgb, dyn_mask, n, combined, combined_static, combined_dyn);
  return check_launch("combine");
}

PGDVS_API int pgdvs_gather_rows(const float *src, const int32_t *idx, const int32_t *count,
                                int64_t capacity, int width, float *dst, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(src && idx && count && dst && width > 0 && capacity >= 0,
                "pgdvs_gather_rows: bad arguments");
  if (capacity == 0) return PGDVS_OK;
  unsigned grid = (unsigned)(cdiv(capacity, 256) < 2048 ? cdiv(capacity, 256) : 2048);
  PGDVS_LAUNCH("gather_rows", gather_rows_kernel, dim3(grid), dim3(256), 0, as_stream(stream), src, idx,
                     count, width, dst);
  return check_launch("gather_rows");
}

PGDVS_API int pgdvs_scatter_keep(const int32_t *idx, const uint8_t *flag, const int32_t *count,
                                 int64_t capacity, uint8_t *keep, int64_t P,
                                 pgdvs_stream_t stream) {
  PGDVS_REQUIRE(idx && flag && count && keep && P >= 0, "pgdvs_scatter_keep: bad arguments");
  if (P == 0) return PGDVS_OK;
  hipError_t e = fill_async(keep, 0, (size_t)P, as_stream(stream));
  if (e != hipSuccess) {
    set_error("scatter_keep memset: %s", hipGetErrorString(e));
    return PGDVS_ERR_LAUNCH;
  }
  if (capacity == 0) return PGDVS_OK;
  unsigned grid = (unsigned)(cdiv(capacity, 256) < 2048 ? cdiv(capacity, 256) : 2048);
  PGDVS_LAUNCH("scatter_keep", scatter_keep_kernel, dim3(grid), dim3(256), 0, as_stream(stream), idx, flag,
                     count, keep);
  return check_launch("scatter_keep");
}
