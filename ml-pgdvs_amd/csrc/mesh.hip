// A10 (mesh variant): dyn_render_type = "mesh" (pgdvs/renderers/pgdvs_renderer_dyn.py:542-669).
// Semantics: the pixel-grid triangulation of the kept source pixels (:550-604, including the
// `vertex index > 0` quirk of :597 that drops every face touching the first kept pixel),
// rendered by pytorch3d 0.7.4 MeshRasterizer (blur_radius 0, faces_per_pixel 1, naive path,
// perspective-correct barycentrics, no clipping / culling) with vertex-colour interpolation
// and hard_rgb_blend on black (pgdvs/utils/pytorch3d_utils.py:50-67).
//
// pytorch3d's naive rasteriser tests every pixel against every face.  Here the triangles are
// never materialised: a thread owns one (source pixel, kind) pair, derives its three corner
// vertices from the dense per-pixel arrays, walks the face's pixel bounding box in the
// target image and resolves visibility with one 64-bit atomicMin per covered pixel on the key
// (z bits << 32 | face order id) -- the total order (z, face id) pytorch3d's queue uses, so
// the winner does not depend on scheduling.  A second pass shades the winners.
#include "raster_cam.h"

namespace pgdvs {

constexpr float kMeshEps = 1e-8f;

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

// CheckPixelInsideFace of rasterize_meshes.cu for blur_radius = 0: true iff the pixel centre is
// strictly inside; pz / bary = perspective-corrected depth and barycentrics
__device__ __forceinline__ bool mesh_pixel_in_face(float px, float py, float3 v0, float3 v1, float3 v2,
                                                   float &pz, float bary[3]) {
  const float zmax = fmaxf(fmaxf(v0.z, v1.z), v2.z);
  const float xmin = fminf(fminf(v0.x, v1.x), v2.x), xmax = fmaxf(fmaxf(v0.x, v1.x), v2.x);
  const float ymin = fminf(fminf(v0.y, v1.y), v2.y), ymax = fmaxf(fmaxf(v0.y, v1.y), v2.y);
  const bool outside = (px > xmax) || (px < xmin) || (py > ymax) || (py < ymin);
  const float face_area = edge_fn(v0.x, v0.y, v1.x, v1.y, v2.x, v2.y);
  const bool zero_area = (face_area <= kMeshEps) && (face_area >= -kMeshEps);
  if (zmax < 0.0f || outside || zero_area) return false;
  const float area = edge_fn(v2.x, v2.y, v0.x, v0.y, v1.x, v1.y) + kMeshEps;
  const float b0 = edge_fn(px, py, v1.x, v1.y, v2.x, v2.y) / area;
  const float b1 = edge_fn(px, py, v2.x, v2.y, v0.x, v0.y) / area;
  const float b2 = edge_fn(px, py, v0.x, v0.y, v1.x, v1.y) / area;
  const float t0 = b0 * v1.z * v2.z;
  const float t1 = v0.z * b1 * v2.z;
  const float t2 = v0.z * v1.z * b2;
  const float den = fmaxf(t0 + t1 + t2, kMeshEps);
  bary[0] = t0 / den;
  bary[1] = t1 / den;
  bary[2] = t2 / den;
  const float z = bary[0] * v0.z + bary[1] * v1.z + bary[2] * v2.z;
  if (z < 0.0f) return false;
  if (!(bary[0] > 0.0f && bary[1] > 0.0f && bary[2] > 0.0f)) return false;
  pz = z;
  return true;
}

// candidate pixel range along one axis for the NDC interval [lo,hi] (a superset)
__device__ __forceinline__ void ndc_to_pix_range(float lo, float hi, int S1, float range, int &i0, int &i1) {
  const float offset = range / 2.0f;
  float a = ((lo + offset) * (float)S1 - offset) / range;
  float b = ((hi + offset) * (float)S1 - offset) / range;
  if (!(a >= -2.0f)) a = -2.0f;
  if (!(b <= (float)S1 + 1.0f)) b = (float)S1 + 1.0f;
  int ja = (int)floorf(a) - 1, jb = (int)ceilf(b) + 1;
  ja = ja < 0 ? 0 : ja;
  jb = jb > S1 - 1 ? S1 - 1 : jb;
  i0 = S1 - 1 - jb;
  i1 = S1 - 1 - ja;
}

// vertices: NDC position of every kept source pixel + the first kept pixel (vertex index 0)
__global__ void __launch_bounds__(256)
mesh_verts_kernel(const float *__restrict__ cam, int H, int W, const uint8_t *__restrict__ keep,
                  const float *__restrict__ pcl, float4 *__restrict__ ndc, int32_t *__restrict__ first) {
  const RasterCam rc = make_raster_cam(cam, H, W);
  const int P = H * W;
  int mine = 0x7fffffff;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (keep[p]) {
      float3 q = point_to_ndc(rc, pcl[(size_t)p * 3], pcl[(size_t)p * 3 + 1], pcl[(size_t)p * 3 + 2]);
      o = make_float4(q.x, q.y, q.z, 1.0f);
      mine = p < mine ? p : mine;
    }
    ndc[p] = o;
  }
  for (int off = 32; off > 0; off >>= 1) {
    int other = __shfl_down(mine, off, 64);
    mine = other < mine ? other : mine;
  }
  if ((threadIdx.x & 63) == 0 && mine != 0x7fffffff) atomicMin(first, mine);
}

__device__ __forceinline__ bool mesh_face_corners(int q0, int kind, int H, int W, const float4 *__restrict__ ndc,
                                                  int first, int &q1, int &q2, float3 &v0, float3 &v1, float3 &v2) {
  const int r = q0 / W, c = q0 - r * W;
  if (r + 1 >= H || c + 1 >= W) return false;
  q1 = kind == 0 ? q0 + W : q0 + W + 1;
  q2 = kind == 0 ? q0 + W + 1 : q0 + 1;
  if (q0 == first || q1 == first || q2 == first) return false;  // vertex index 0 counts as missing (:597)
  const float4 a = ndc[q0], b = ndc[q1], d = ndc[q2];
  if (!(a.w != 0.0f && b.w != 0.0f && d.w != 0.0f)) return false;
  v0 = make_float3(a.x, a.y, a.z);
  v1 = make_float3(b.x, b.y, b.z);
  v2 = make_float3(d.x, d.y, d.z);
  return true;
}

__global__ void __launch_bounds__(256)
mesh_faces_kernel(int H, int W, const float4 *__restrict__ ndc, const int32_t *__restrict__ first_p,
                  unsigned long long *__restrict__ zkey) {
  const int P = H * W;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * P) return;
  const int kind = t >= P ? 1 : 0;
  const int q0 = t - kind * P;
  int q1, q2;
  float3 v0, v1, v2;
  if (!mesh_face_corners(q0, kind, H, W, ndc, *first_p, q1, q2, v0, v1, v2)) return;
  const float range_x = W > H ? 2.0f * (float)W / (float)H : 2.0f;
  const float range_y = H > W ? 2.0f * (float)H / (float)W : 2.0f;
  const float xmin = fminf(fminf(v0.x, v1.x), v2.x), xmax = fmaxf(fmaxf(v0.x, v1.x), v2.x);
  const float ymin = fminf(fminf(v0.y, v1.y), v2.y), ymax = fmaxf(fmaxf(v0.y, v1.y), v2.y);
  if (!(xmin <= xmax && ymin <= ymax)) return;  // NaN vertex: never inside
  int x0, x1, y0, y1;
  ndc_to_pix_range(xmin, xmax, W, range_x, x0, x1);
  ndc_to_pix_range(ymin, ymax, H, range_y, y0, y1);
  for (int yi = y0; yi <= y1; ++yi) {
    const float yf = pix_to_ndc(H - 1 - yi, H, range_y);
    for (int xi = x0; xi <= x1; ++xi) {
      const float xf = pix_to_ndc(W - 1 - xi, W, range_x);
      float pz, b[3];
      if (!mesh_pixel_in_face(xf, yf, v0, v1, v2, pz, b)) continue;
      // pz >= 0: the IEEE bit pattern is monotone in the value
      const unsigned long long key = ((unsigned long long)__float_as_uint(pz) << 32) | (unsigned)t;
      atomicMin(&zkey[(size_t)yi * W + xi], key);
    }
  }
}

__global__ void __launch_bounds__(256)
mesh_shade_kernel(int H, int W, const float4 *__restrict__ ndc, const int32_t *__restrict__ first_p,
                  const unsigned long long *__restrict__ zkey, const float *__restrict__ rgb,
                  float *__restrict__ img_planar, float *__restrict__ mask, int32_t *__restrict__ face_out) {
  const int P = H * W;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P) return;
  const unsigned long long key = zkey[t];
  float out[3] = {0.0f, 0.0f, 0.0f};
  float m = 0.0f;
  int fid = -1;
  if (key != ~0ull) {
    fid = (int)(unsigned)(key & 0xffffffffull);
    const int kind = fid >= P ? 1 : 0;
    const int q0 = fid - kind * P;
    int q1, q2;
    float3 v0, v1, v2;
    mesh_face_corners(q0, kind, H, W, ndc, *first_p, q1, q2, v0, v1, v2);
    const float range_x = W > H ? 2.0f * (float)W / (float)H : 2.0f;
    const float range_y = H > W ? 2.0f * (float)H / (float)W : 2.0f;
    const int yi = t / W, xi = t - yi * W;
    float pz, b[3];
    mesh_pixel_in_face(pix_to_ndc(W - 1 - xi, W, range_x), pix_to_ndc(H - 1 - yi, H, range_y), v0, v1, v2, pz, b);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float a = b[0] * rgb[(size_t)q0 * 3 + k];
      a = a + b[1] * rgb[(size_t)q1 * 3 + k];
      a = a + b[2] * rgb[(size_t)q2 * 3 + k];
      out[k] = a;
    }
    m = 1.0f;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) img_planar[(size_t)k * P + t] = out[k];
  mask[t] = m;
  if (face_out) face_out[t] = fid;
}

}  // namespace pgdvs

using namespace pgdvs;

PGDVS_API int64_t pgdvs_mesh_render_workspace_bytes(int H, int W) {
  const int64_t P = (int64_t)H * W;
  return 256 + align_up(P * 16, 256) + align_up(P * 8, 256);
}

PGDVS_API int pgdvs_mesh_render(const float *cam_tgt, int H, int W, const uint8_t *keep, const float *pcl,
                                const float *rgb, float *img_planar, float *mask, int32_t *face_out,
                                void *workspace, int64_t workspace_bytes, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(cam_tgt && keep && pcl && rgb && img_planar && mask, "pgdvs_mesh_render: null pointer");
  PGDVS_REQUIRE(H > 0 && W > 0 && (int64_t)H * W < (1ll << 30), "pgdvs_mesh_render: bad shape");
  if (!workspace || workspace_bytes < pgdvs_mesh_render_workspace_bytes(H, W)) {
    set_error("pgdvs_mesh_render: workspace too small");
    return PGDVS_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const int64_t P = (int64_t)H * W;
  char *p = reinterpret_cast<char *>(workspace);
  int32_t *first = reinterpret_cast<int32_t *>(p);
  float4 *ndc = reinterpret_cast<float4 *>(p + 256);
  unsigned long long *zkey = reinterpret_cast<unsigned long long *>(p + 256 + align_up(P * 16, 256));
  hipError_t e = hipMemsetAsync(first, 0x7f, 4, st);
  if (e == hipSuccess) e = fill_async(zkey, 0xff, (size_t)P * 8, st);
  if (e != hipSuccess) {
    set_error("mesh_render memset: %s", hipGetErrorString(e));
    return PGDVS_ERR_LAUNCH;
  }
  unsigned gv = (unsigned)(cdiv(P, 256) < 2048 ? cdiv(P, 256) : 2048);
  PGDVS_LAUNCH("mesh_verts", mesh_verts_kernel, dim3(gv), dim3(256), 0, st, cam_tgt, H, W, keep, pcl, ndc, first);
  PGDVS_LAUNCH("mesh_faces", mesh_faces_kernel, dim3((unsigned)cdiv(2 * P, 256)), dim3(256), 0, st, H, W,
               (const float4 *)ndc, (const int32_t *)first, zkey);
  PGDVS_LAUNCH("mesh_shade", mesh_shade_kernel, dim3((unsigned)cdiv(P, 256)), dim3(256), 0, st, H, W,
               (const float4 *)ndc, (const int32_t *)first, (const unsigned long long *)zkey, rgb, img_planar, mask,
               face_out);
  return check_launch("mesh_render");
}
