// A12: static point-cloud aggregation across the S source frames of a video
// (pgdvs/datasets/nvidia_eval_pure_geo.py:183-277, _compute_pcl at
// pgdvs/datasets/nvidia_eval.py:840-847, rays at pgdvs/datasets/base.py:507-546).
//
// Upstream this is single-threaded numpy at dataset construction.  Frame i may only add
// the static pixels that the cloud accumulated from frames < i does not already cover
// (integer-truncated projection occupancy), so frames are processed in order; inside a
// frame everything is data-parallel:
//   mark   : project the accumulated cloud (fp64, as numpy does) and stamp occ[row,col] with
//            the frame index (no clearing between frames)
//   count  : per-block number of selected pixels (static && not stamped)
//   scan   : block offsets, append base, new cloud size
//   append : ordered scatter -> row-major pixel order, the order numpy's boolean indexing
//            produces (point ids matter: the rasteriser breaks z ties by id); unproject the
//            selected pixels (fp32 rays) and append (xyz,rgb) rows
// No host synchronisation: the running point count lives on the device and every
// kernel reads it there.
#include "common.h"

namespace pgdvs {

struct ProjF64 {
  double K3[9];
  double w2c[16];
};

// _compute_pcl_proj_mask, nvidia_eval_pure_geo.py:257-277 (no z>0 test, no epsilon,
// closed bounds, astype(int) truncation)
__global__ void __launch_bounds__(256)
agg_mark_kernel(const float *__restrict__ cloud, const int64_t *__restrict__ count, ProjF64 pj,
                int H, int W, int frame, uint16_t *__restrict__ occ) {
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
    occ[(int64_t)row * W + (int64_t)col] = (uint16_t)frame;
  }
}

constexpr int kAggBlock = 1024;
constexpr int kAggItems = 4;
constexpr int kAggTile = kAggBlock * kAggItems;

// selection flag of pixel p in frame `frame`: static and not covered by the accumulated
// cloud.  occ holds the index of the last frame whose mark pass touched the pixel, so it
// never needs clearing between frames.
__device__ __forceinline__ bool agg_selected(const uint8_t *__restrict__ dyn_mask,
                                             const uint16_t *__restrict__ occ, int frame, int p) {
  bool st = dyn_mask[p] == 0;
  if (frame > 0) st = st && occ[p] != (uint16_t)frame;
  return st;
}

// per-block number of selected pixels (tmp_st_mask & ~tmp_proj_mask, :224-245)
__global__ void __launch_bounds__(kAggBlock)
agg_count_kernel(const uint8_t *__restrict__ dyn_mask, const uint16_t *__restrict__ occ, int frame,
                 int P, int32_t *__restrict__ block_counts) {
  __shared__ int wave_sums[kAggBlock / kWave];
  int base = blockIdx.x * kAggTile + threadIdx.x * kAggItems;
  int c = 0;
#pragma unroll
  for (int k = 0; k < kAggItems; ++k)
    if (base + k < P) c += agg_selected(dyn_mask, occ, frame, base + k);
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0) wave_sums[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int i = 0; i < kAggBlock / kWave; ++i) s += wave_sums[i];
    block_counts[blockIdx.x] = s;
  }
}

// exclusive scan of the block counts; publishes the append base (= current cloud size) and
// bumps the cloud size by the number of selected pixels
__global__ void __launch_bounds__(1024)
agg_scan_kernel(int32_t *__restrict__ block_counts, int nb, int64_t *__restrict__ count,
                int64_t *__restrict__ append_base, int64_t capacity) {
  __shared__ int wave_sums[1024 / kWave];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int start = 0; start < nb; start += 1024) {
    int i = start + threadIdx.x;
    int v = i < nb ? block_counts[i] : 0;
    int x = v;
    for (int off = 1; off < 64; off <<= 1) {
      int y = __shfl_up(x, off, 64);
      if ((threadIdx.x & 63) >= off) x += y;
    }
    int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wave_sums[wave] = x;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += wave_sums[w];
    int incl = carry + wave_off + x;
    if (i < nb) block_counts[i] = incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int64_t base = *count;
    *append_base = base;
    int64_t c = base + (int64_t)carry;
    *count = c > capacity ? capacity : c;
  }
}

// ordered scatter: selected pixel -> its row-major rank -> unproject (fp32 rays) and append
// (tmp_pcl[tmp_st_mask], tmp_img[tmp_st_mask] :247-251)
__global__ void __launch_bounds__(kAggBlock)
agg_append_kernel(const uint8_t *__restrict__ dyn_mask, const uint16_t *__restrict__ occ, int frame,
                  int P, const int32_t *__restrict__ block_offsets, CamBlock cam, int W,
                  const float *__restrict__ depth, const float *__restrict__ rgb,
                  float *__restrict__ cloud, const int64_t *__restrict__ append_base,
                  int64_t capacity) {
  __shared__ int wave_sums[kAggBlock / kWave];
  int base = blockIdx.x * kAggTile + threadIdx.x * kAggItems;
  bool f[kAggItems];
  int c = 0;
#pragma unroll
  for (int k = 0; k < kAggItems; ++k) {
    f[k] = (base + k < P) && agg_selected(dyn_mask, occ, frame, base + k);
    c += f[k];
  }
  int x = c;
  for (int off = 1; off < 64; off <<= 1) {
    int y = __shfl_up(x, off, 64);
    if ((threadIdx.x & 63) >= off) x += y;
  }
  int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 63) wave_sums[wave] = x;
  __syncthreads();
  int wave_off = 0;
  for (int w = 0; w < wave; ++w) wave_off += wave_sums[w];
  int64_t pos = *append_base + block_offsets[blockIdx.x] + wave_off + x - c;
#pragma unroll
  for (int k = 0; k < kAggItems; ++k) {
    if (!f[k]) continue;
    if (pos < capacity) {
      int p = base + k;
      int r = p / W, col = p - r * W;
      float u = (float)col, v = (float)r;
      float d = depth[p];
      float *o = cloud + pos * 6;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        float dir = cam.v[PGDVS_CAM_M + a * 3 + 0] * u;
        dir = dir + cam.v[PGDVS_CAM_M + a * 3 + 1] * v;
        dir = dir + cam.v[PGDVS_CAM_M + a * 3 + 2];
        o[a] = cam.v[PGDVS_CAM_O + a] + dir * d;
        o[3 + a] = rgb[(size_t)p * 3 + a];
      }
    }
    ++pos;
  }
}

struct AggWs {
  uint16_t *occ;
  int32_t *block_counts;
  int64_t *append_base;
  int64_t total_bytes;
};

static AggWs agg_ws_layout(void *base, int H, int W) {
  AggWs w;
  int64_t P = (int64_t)H * W;
  char *p = reinterpret_cast<char *>(base);
  int64_t off = 0;
  w.occ = reinterpret_cast<uint16_t *>(p + off);
  off += align_up(P * 2, 256);
  w.block_counts = reinterpret_cast<int32_t *>(p + off);
  off += align_up(cdiv(P, kAggTile) * 4, 256);
  w.append_base = reinterpret_cast<int64_t *>(p + off);
  off += 256;
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
  PGDVS_REQUIRE(S > 0 && S < 65535 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 31) && capacity > 0,
                "pgdvs_static_aggregate: bad shape");
  AggWs ws = agg_ws_layout(workspace, H, W);
  if (!workspace || workspace_bytes < ws.total_bytes) {
    set_error("pgdvs_static_aggregate: workspace too small");
    return PGDVS_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const int64_t P = (int64_t)H * W;
  hipError_t e = hipMemsetAsync(count_out, 0, sizeof(int64_t), st);
  if (e == hipSuccess) e = hipMemsetAsync(ws.occ, 0, (size_t)P * 2, st);
  if (e != hipSuccess) {
    set_error("static_aggregate memset: %s", hipGetErrorString(e));
    return PGDVS_ERR_LAUNCH;
  }
  const int nb = (int)cdiv(P, kAggTile);
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
    if (i > 0)
      PGDVS_LAUNCH("agg_mark", agg_mark_kernel, dim3(2048), dim3(256), 0, st, out, count_out, pj, H, W, i,
                   ws.occ);
    PGDVS_LAUNCH("agg_count", agg_count_kernel, dim3(nb), dim3(kAggBlock), 0, st, mask_i, ws.occ, i, (int)P,
                 ws.block_counts);
    PGDVS_LAUNCH("agg_scan", agg_scan_kernel, dim3(1), dim3(1024), 0, st, ws.block_counts, nb, count_out,
                 ws.append_base, capacity);
    PGDVS_LAUNCH("agg_append", agg_append_kernel, dim3(nb), dim3(kAggBlock), 0, st, mask_i, ws.occ, i,
                 (int)P, ws.block_counts, cam, W, depths + (size_t)i * P, rgbs + (size_t)i * P * 3, out,
                 ws.append_base, capacity);
  }
  return check_launch("static_aggregate");
}
