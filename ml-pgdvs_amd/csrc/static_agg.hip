// A12: static point-cloud aggregation across the S source frames of a video
// (pgdvs/datasets/nvidia_eval_pure_geo.py:183-277, _compute_pcl at
// pgdvs/datasets/nvidia_eval.py:840-847, rays at pgdvs/datasets/base.py:507-546).
//
// Upstream this is single-threaded numpy at dataset construction.  Frame i may only add
// the static pixels that the cloud accumulated from frames < i does not already cover
// (integer-truncated projection occupancy), so frames are processed in order; inside a
// frame everything is data-parallel:
//   mark    : project the accumulated cloud (fp64, as numpy does) and set occ[row,col]
//   flags   : static && !occupied
//   compact : ordered stream compaction -> row-major pixel order, the order numpy's
//             boolean indexing produces (point ids matter: the rasteriser breaks z ties
//             by id)
//   append  : unproject the selected pixels (fp32 rays) and append (xyz,rgb) rows
// No host synchronisation: the running point count lives on the device and every
// kernel reads it there.
#include "common.h"
#include "scan.h"

namespace pgdvs {

struct ProjF64 {
  double K3[9];
  double w2c[16];
};

// _compute_pcl_proj_mask, nvidia_eval_pure_geo.py:257-277 (no z>0 test, no epsilon,
// closed bounds, astype(int) truncation)
__global__ void __launch_bounds__(256)
agg_mark_kernel(const float *__restrict__ cloud, const int64_t *__restrict__ count, ProjF64 pj,
                int H, int W, uint8_t *__restrict__ occ) {
  const int64_t n = *count;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    double x = (double)cloud[i * 6 + 0], y = (double)cloud[i * 6 + 1], z = (double)cloud[i * 6 + 2];
    double vc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double s = pj.w2c[k * 4 + 0] * x;
      s = s + pj.w2c[k * 4 + 1] * y;
      s = s + pj.w2c[k * 4 + 2] * z;
      s = s + pj.w2c[k * 4 + 3];
      vc[k] = s;
    }
    double cx = vc[0] / vc[3], cy = vc[1] / vc[3], cz = vc[2] / vc[3];
    double pp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double s = pj.K3[k * 3 + 0] * cx;
      s = s + pj.K3[k * 3 + 1] * cy;
      s = s + pj.K3[k * 3 + 2] * cz;
      pp[k] = s;
    }
    double col = pp[0] / pp[2], row = pp[1] / pp[2];
    if (!(row >= 0.0 && row <= (double)(H - 1))) continue;
    if (!(col >= 0.0 && col <= (double)(W - 1))) continue;
    occ[(int64_t)row * W + (int64_t)col] = 1;
  }
}

__global__ void __launch_bounds__(256)
agg_flags_kernel(const uint8_t *__restrict__ dyn_mask, const uint8_t *__restrict__ occ, int P,
                 uint8_t *__restrict__ flags) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  bool st = dyn_mask[p] == 0;
  if (occ) st = st && occ[p] == 0;
  flags[p] = (uint8_t)st;
}

// tmp_pcl[tmp_st_mask] / tmp_img[tmp_st_mask] appended to the cloud (:247-251)
__global__ void __launch_bounds__(256)
agg_append_kernel(const int32_t *__restrict__ idx, const int32_t *__restrict__ cnt, CamBlock cam,
                  int W, const float *__restrict__ depth, const float *__restrict__ rgb,
                  float *__restrict__ cloud, const int64_t *__restrict__ count, int64_t capacity) {
  const int n = *cnt;
  const int64_t base = *count;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    if (base + j >= capacity) break;
    int p = idx[j];
    int r = p / W, c = p - r * W;
    float u = (float)c, v = (float)r;
    float d = depth[p];
    float *o = cloud + (base + j) * 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float dir = cam.v[PGDVS_CAM_M + k * 3 + 0] * u;
      dir = dir + cam.v[PGDVS_CAM_M + k * 3 + 1] * v;
      dir = dir + cam.v[PGDVS_CAM_M + k * 3 + 2];
      o[k] = cam.v[PGDVS_CAM_O + k] + dir * d;
      o[3 + k] = rgb[(size_t)p * 3 + k];
    }
  }
}

__global__ void agg_bump_kernel(const int32_t *__restrict__ cnt, int64_t *__restrict__ count,
                                int64_t capacity) {
  int64_t c = *count + (int64_t)*cnt;
  *count = c > capacity ? capacity : c;
}

struct AggWs {
  uint8_t *occ, *flags;
  int32_t *idx, *cnt;
  void *compact_ws;
  int64_t compact_bytes, total_bytes;
};

static AggWs agg_ws_layout(void *base, int H, int W) {
  AggWs w;
  int64_t P = (int64_t)H * W;
  char *p = reinterpret_cast<char *>(base);
  int64_t off = 0;
  w.occ = reinterpret_cast<uint8_t *>(p + off);
  off += align_up(P, 256);
  w.flags = reinterpret_cast<uint8_t *>(p + off);
  off += align_up(P, 256);
  w.idx = reinterpret_cast<int32_t *>(p + off);
  off += align_up(P * 4, 256);
  w.cnt = reinterpret_cast<int32_t *>(p + off);
  off += 256;
  w.compact_ws = p + off;
  w.compact_bytes = compact_workspace_bytes(P);
  off += w.compact_bytes;
  w.total_bytes = off;
  return w;
}

}  // namespace pgdvs

using namespace pgdvs;

PGDVS_API int64_t pgdvs_static_aggregate_workspace_bytes(int H, int W) {
  if (H <= 0 || W <= 0) return -1;
  return agg_ws_layout(nullptr, H, W).total_bytes;
}

PGDVS_API int pgdvs_static_aggregate(const float *rgbs, const float *depths,
                                     const uint8_t *dyn_masks, const double *K3s_host,
                                     const double *c2ws_host, int S, int H, int W, float *out,
                                     int64_t capacity, int64_t *count_out, void *workspace,
                                     int64_t workspace_bytes, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(rgbs && depths && dyn_masks && K3s_host && c2ws_host && out && count_out,
                "pgdvs_static_aggregate: null pointer");
  PGDVS_REQUIRE(S > 0 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 31) && capacity > 0,
                "pgdvs_static_aggregate: bad shape");
  AggWs ws = agg_ws_layout(workspace, H, W);
  if (!workspace || workspace_bytes < ws.total_bytes) {
    set_error("pgdvs_static_aggregate: workspace too small");
    return PGDVS_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const int64_t P = (int64_t)H * W;
  hipError_t e = hipMemsetAsync(count_out, 0, sizeof(int64_t), st);
  if (e != hipSuccess) {
    set_error("static_aggregate memset: %s", hipGetErrorString(e));
    return PGDVS_ERR_LAUNCH;
  }
  for (int i = 0; i < S; ++i) {
    const double *K3 = K3s_host + (size_t)i * 9;
    const double *c2w = c2ws_host + (size_t)i * 16;
    ProjF64 pj;
    for (int k = 0; k < 9; ++k) pj.K3[k] = K3[k];
    if (inv_f64(c2w, pj.w2c, 4) != 0) {
      set_error("pgdvs_static_aggregate: singular c2w for frame %d", i);
      return PGDVS_ERR_INVALID;
    }
    // rays use K and c2w cast to fp32 (torch.FloatTensor, nvidia_eval.py:841-842)
    float flat[34];
    flat[0] = (float)H;
    flat[1] = (float)W;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c)
        flat[2 + r * 4 + c] = (r < 3 && c < 3) ? (float)K3[r * 3 + c] : (r == c ? 1.0f : 0.0f);
    for (int k = 0; k < 16; ++k) flat[18 + k] = (float)c2w[k];
    CamBlock cam;
    if (cam_block_from_flat(flat, cam.v) != 0) {
      set_error("pgdvs_static_aggregate: singular intrinsics for frame %d", i);
      return PGDVS_ERR_INVALID;
    }
    const uint8_t *mask_i = dyn_masks + (size_t)i * P;
    if (i > 0) {
      (void)hipMemsetAsync(ws.occ, 0, (size_t)P, st);
      PGDVS_LAUNCH("agg_mark", agg_mark_kernel, dim3(2048), dim3(256), 0, st, out, count_out, pj, H, W,
                         ws.occ);
    }
    PGDVS_LAUNCH("agg_flags", agg_flags_kernel, dim3((unsigned)cdiv(P, 256)), dim3(256), 0, st, mask_i,
                       i > 0 ? ws.occ : nullptr, (int)P, ws.flags);
    int rc = compact_u8(ws.flags, P, ws.idx, ws.cnt, ws.compact_ws, ws.compact_bytes, st);
    if (rc != PGDVS_OK) return rc;
    PGDVS_LAUNCH("agg_append", agg_append_kernel, dim3(2048), dim3(256), 0, st, ws.idx, ws.cnt, cam, W,
                       depths + (size_t)i * P, rgbs + (size_t)i * P * 3, out, count_out, capacity);
    PGDVS_LAUNCH("agg_bump", agg_bump_kernel, dim3(1), dim3(1), 0, st, ws.cnt, count_out, capacity);
  }
  return check_launch("static_aggregate");
}
