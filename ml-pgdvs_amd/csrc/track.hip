// A17: tracker-window point aggregation
// (pgdvs/renderers/pgdvs_renderer_dyn_track.py:98-396, prepare_data :716-721).
// Point tracks and visibilities are inputs (the trackers are third-party networks, out of
// scope).  The reference groups the track samples per frame on the host (torch.unique,
// boolean masks, argsort); here one thread owns one track: it decides validity, picks the
// two visible frames nearest in time, samples colour / depth from those two frames, lifts
// both samples to 3-D and inter/extrapolates to the target time.  Everything downstream
// (compaction, kNN filters, concatenation with the base cloud) keeps its element counts on
// the device, so the whole row runs without a host synchronisation.
#include "common.h"

namespace pgdvs {

constexpr int kTrackMaxFrames = 64;

struct TrackArgs {
  const float *tracks;      // [P,N,2] (col,row)
  const uint8_t *vis;       // [P,N]
  const float *times;       // [N] raw time stamps
  const float *time_tgt;    // [1] raw
  const float *rgbs;        // [N,H,W,3]
  const float *depths;      // [N,H,W]
  const float *cams;        // [N,CAM_BLOCK]
  uint8_t *valid;
  float *pcl, *rgb;
  int64_t P;
  int N, H, W;
  uint8_t kind[kTrackMaxFrames];  // 1: temporally-closest frame, 2: real track frame
};

__global__ void __launch_bounds__(256) track_points_kernel(TrackArgs a) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.P) return;
  const int N = a.N;
  // prepare_data :718-721: time stamps are shifted to start from 0 first
  float tmin = a.times[0];
  for (int f = 1; f < N; ++f) tmin = fminf(tmin, a.times[f]);
  const float tt = *a.time_tgt - tmin;
  bool seen_closest = false;
  int n_real = 0;
  int f0 = -1, f1 = -1;
  float d0 = __builtin_inff(), d1 = __builtin_inff();
  for (int f = 0; f < N; ++f) {
    if (!a.vis[p * N + f]) continue;
    seen_closest = seen_closest || a.kind[f] == 1;
    n_real += a.kind[f] == 2;
    float d = fabsf((a.times[f] - tmin) - tt);
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
  const bool ok = !seen_closest && n_real >= 2;
  a.valid[p] = (uint8_t)ok;
  float op[3] = {0.0f, 0.0f, 0.0f}, oc[3] = {0.0f, 0.0f, 0.0f};
  if (ok) {
    const float fw = (float)a.W, fh = (float)a.H;
    float X[2][3], col[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int f = s == 0 ? f0 : f1;
      const float u = a.tracks[(p * N + f) * 2 + 0], v = a.tracks[(p * N + f) * 2 + 1];
      const float gx = 2.0f * u / fw - 1.0f, gy = 2.0f * v / fh - 1.0f;
      const float ix = ((gx + 1.0f) / 2.0f) * (fw - 1.0f), iy = ((gy + 1.0f) / 2.0f) * (fh - 1.0f);
      const float x0f = floorf(ix), y0f = floorf(iy);
      const bool fin = isfinite(ix) && isfinite(iy) && fabsf(ix) < 1e9f && fabsf(iy) < 1e9f;
      const int x0 = fin ? (int)x0f : -10, y0 = fin ? (int)y0f : -10, x1 = x0 + 1, y1 = y0 + 1;
      const float wnw = ((float)x1 - ix) * ((float)y1 - iy), wne = (ix - (float)x0) * ((float)y1 - iy);
      const float wsw = ((float)x1 - ix) * (iy - (float)y0), wse = (ix - (float)x0) * (iy - (float)y0);
      const bool inx0 = x0 >= 0 && x0 < a.W, inx1 = x1 >= 0 && x1 < a.W;
      const bool iny0 = y0 >= 0 && y0 < a.H, iny1 = y1 >= 0 && y1 < a.H;
      const float *img = a.rgbs + (size_t)f * a.H * a.W * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float acc = 0.0f;
        if (inx0 && iny0) acc = acc + img[(y0 * a.W + x0) * 3 + k] * wnw;
        if (inx1 && iny0) acc = acc + img[(y0 * a.W + x1) * 3 + k] * wne;
        if (inx0 && iny1) acc = acc + img[(y1 * a.W + x0) * 3 + k] * wsw;
        if (inx1 && iny1) acc = acc + img[(y1 * a.W + x1) * 3 + k] * wse;
        col[s][k] = acc;
      }
      const float nx = nearbyintf(((gx + 1.0f) * fw - 1.0f) / 2.0f);
      const float ny = nearbyintf(((gy + 1.0f) * fh - 1.0f) / 2.0f);
      float dsamp = 0.0f;
      if (nx >= 0.0f && nx <= fw - 1.0f && ny >= 0.0f && ny <= fh - 1.0f)
        dsamp = a.depths[(size_t)f * a.H * a.W + (int)ny * a.W + (int)nx];
      const float *M = a.cams + (size_t)f * PGDVS_CAM_BLOCK + PGDVS_CAM_M;
      const float *o = a.cams + (size_t)f * PGDVS_CAM_BLOCK + PGDVS_CAM_O;
#pragma unroll
      for (int k = 0; k < 3; ++k) X[s][k] = o[k] + dot3(M + k * 3, u, v, 1.0f) * dsamp;
    }
    const float t0 = a.times[f0] - tmin, t1 = a.times[f1] - tmin;
    const float ratio = (tt - t0) / ((t1 - t0) + 1e-8f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      op[k] = X[0][k] + (X[1][k] - X[0][k]) * ratio;
      oc[k] = (col[0][k] + col[1][k]) / 2.0f;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    a.pcl[p * 3 + k] = op[k];
    a.rgb[p * 3 + k] = oc[k];
  }
}

// flag[i] = avg[i] < thres*mult, unless the gate count is zero (no base cloud): then
// avg[i] < *alt_thres, or 1 when there is no alternative threshold (:296-298,:363-371).
// Flags beyond *count (up to capacity) are cleared.
__global__ void __launch_bounds__(256)
threshold_flags_kernel(const float *__restrict__ avg, const int32_t *__restrict__ count, int64_t capacity,
                       const float *__restrict__ thres, float mult, const float *__restrict__ alt_thres,
                       const int32_t *__restrict__ gate_count, uint8_t *__restrict__ flag) {
  const int64_t n = *count;
  const bool gated = gate_count != nullptr && *gate_count == 0;
  const bool pass_all = gated && alt_thres == nullptr;
  const float t = gated ? (alt_thres ? *alt_thres : 0.0f) : *thres * mult;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < capacity; i += (int64_t)gridDim.x * blockDim.x)
    flag[i] = i < n ? (pass_all ? (uint8_t)1 : (uint8_t)(avg[i] < t)) : (uint8_t)0;
}

// out = [a[0:*ca], b[0:*cb]] (rows of `width` floats); empty when require_a and *ca == 0
// (:390-394: the base cloud is appended only to a non-empty track cloud)
__global__ void __launch_bounds__(256)
concat_rows_kernel(const float *__restrict__ a, const int32_t *__restrict__ count_a,
                   const float *__restrict__ b, const int32_t *__restrict__ count_b, int width,
                   int require_a, float *__restrict__ out, int32_t *__restrict__ count_out) {
  const int64_t ca = *count_a;
  const int64_t cb = (b != nullptr && !(require_a && ca == 0)) ? *count_b : 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) *count_out = (int32_t)(ca + cb);
  const int64_t na = ca * width, total = (ca + cb) * width;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = i < na ? a[i] : b[i - na];
}

}  // namespace pgdvs

using namespace pgdvs;

PGDVS_API int pgdvs_track_points(const float *tracks, const uint8_t *visibles, int64_t P, int N,
                                 const uint8_t *frame_kind_host, const float *times,
                                 const float *time_tgt, const float *rgbs, const float *depths, int H,
                                 int W, const float *cams, uint8_t *valid, float *pcl, float *rgb,
                                 pgdvs_stream_t stream) {
  PGDVS_REQUIRE(tracks && visibles && frame_kind_host && times && time_tgt && rgbs && depths && cams &&
                    valid && pcl && rgb,
                "pgdvs_track_points: null pointer");
  PGDVS_REQUIRE(P >= 0 && N >= 1 && N <= kTrackMaxFrames && H > 0 && W > 0,
                "pgdvs_track_points: bad shape (at most %d frames)", kTrackMaxFrames);
  if (P == 0) return PGDVS_OK;
  TrackArgs a;
  a.tracks = tracks;
  a.vis = visibles;
  a.times = times;
  a.time_tgt = time_tgt;
  a.rgbs = rgbs;
  a.depths = depths;
  a.cams = cams;
  a.valid = valid;
  a.pcl = pcl;
  a.rgb = rgb;
  a.P = P;
  a.N = N;
  a.H = H;
  a.W = W;
  for (int f = 0; f < kTrackMaxFrames; ++f) a.kind[f] = f < N ? frame_kind_host[f] : 0;
  PGDVS_LAUNCH("track_points", track_points_kernel, dim3((unsigned)cdiv(P, 256)), dim3(256), 0,
               as_stream(stream), a);
  return check_launch("track_points");
}

PGDVS_API int pgdvs_threshold_flags(const float *avg, const int32_t *count, int64_t capacity,
                                    const float *thres, float mult, const float *alt_thres,
                                    const int32_t *gate_count, uint8_t *flag_out, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(avg && count && thres && flag_out && capacity >= 0, "pgdvs_threshold_flags: bad arguments");
  if (capacity == 0) return PGDVS_OK;
  unsigned grid = (unsigned)(cdiv(capacity, 256) < 1024 ? cdiv(capacity, 256) : 1024);
  PGDVS_LAUNCH("threshold_flags", threshold_flags_kernel, dim3(grid), dim3(256), 0, as_stream(stream), avg,
               count, capacity, thres, mult, alt_thres, gate_count, flag_out);
  return check_launch("threshold_flags");
}

PGDVS_API int pgdvs_concat_rows(const float *a, const int32_t *count_a, int64_t capacity_a, const float *b,
                                const int32_t *count_b, int64_t capacity_b, int width, int require_a,
                                float *out, int32_t *count_out, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(a && count_a && out && count_out && width >= 1 && capacity_a >= 0 && capacity_b >= 0,
                "pgdvs_concat_rows: bad arguments");
  PGDVS_REQUIRE(b == nullptr || count_b != nullptr, "pgdvs_concat_rows: count_b missing");
  int64_t total = (capacity_a + (b ? capacity_b : 0)) * width;
  unsigned grid = (unsigned)(cdiv(total, 256) < 2048 ? (cdiv(total, 256) > 0 ? cdiv(total, 256) : 1) : 2048);
  PGDVS_LAUNCH("concat_rows", concat_rows_kernel, dim3(grid), dim3(256), 0, as_stream(stream), a, count_a,
               b, count_b, width, require_a, out, count_out);
  return check_launch("concat_rows");
}
