// One native call per target view: the whole geometric path of PGDVSRenderer.forward
// (pgdvs/renderers/pgdvs_renderer.py:84-178 with StaticGeoPointRenderer + the softsplat dynamic branch) enqueued
// from C++, as the reference's evaluator drives it once per view (pgdvs/engines/evaluator_pgdvs.py:36-54).
//
// Nothing new is computed here: the function carves one caller-given workspace into the scratch of the per-op
// entry points (include/pgdvs_hip.h) and calls them in the order the Python classes do.  What it removes is the host
// cost of ~85 ctypes calls, ~40 allocator round trips and the Python between them (0.53-0.74 ms per view in round 3
// against 0.86 ms of GPU time).
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <thread>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"
#include "fused.h"
#include "scan.h"

namespace pgdvs {

int view_prep(const float *flat_tgt, const float *flat_src, const float *time_src, const float *time_tgt, float *blocks,
              float *times, hipStream_t st);  // dyn.hip

int static_aggregate_for_view(const float *rgbs, const float *depths, const uint8_t *dyn_masks, const double *K3s_host,
                              const double *c2ws_host, int S, int H, int W, float *out, float *xyz_out, int64_t capacity,
                              int64_t *count_out, void *workspace, int64_t workspace_bytes, pgdvs_stream_t stream,
                              bool params_cached, void *zero_extra, int64_t zero_extra_bytes);  // static_agg.hip

// the two halves of the splat composite (softsplat.hip)
int dyn_splat_scatter_part(int H, int W, const float *rgb1, const float *rgb2, const float *flow12, const float *flow_1_to_tgt,
                           const float *valid_dyn_mask_1, const float *noise, const unsigned long long *rng, float alpha,
                           void *workspace, hipStream_t st);
int dyn_splat_finish_part(int H, int W, unsigned long long *rng, const float *static_rgb, float *render_dyn_rgb,
                          float *render_dyn_mask, float *combined, float *combined_static, float *combined_dyn, void *workspace,
                          hipStream_t st);

// counters of the sub-workspaces (raster.hip, knn_grid.hip, static_agg.hip)
void raster_counters(const void *workspace, int64_t n_rows, int H, int W, float radius, int64_t *out_dev, hipStream_t st);
void knn_grid_counter_words(const void *workspace, int64_t capacity, int64_t qcapacity, const int32_t **to_ring,
                            const int32_t **to_coarse, const int32_t **to_exhaustive);
const unsigned *agg_stat_words(const void *workspace, int S, int H, int W, int64_t capacity);

__global__ void view_counters_kernel(const int64_t *__restrict__ static_rows, int64_t static_rows_host,
                                     const int32_t *__restrict__ knn_queries, const int32_t *__restrict__ to_ring,
                                     const int32_t *__restrict__ to_coarse, const int32_t *__restrict__ to_exhaustive,
                                     const unsigned *__restrict__ agg_stat, int64_t *__restrict__ out) {
  if (threadIdx.x != 0) return;
  out[PGDVS_VIEW_CNT_STATIC_ROWS] = static_rows ? *static_rows : static_rows_host;
  out[PGDVS_VIEW_CNT_KNN_QUERIES] = knn_queries ? *knn_queries : 0;
  out[PGDVS_VIEW_CNT_KNN_TO_RING] = to_ring ? *to_ring : 0;
  out[PGDVS_VIEW_CNT_KNN_TO_COARSE] = to_coarse ? *to_coarse : 0;
  out[PGDVS_VIEW_CNT_KNN_TO_EXHAUSTIVE] = to_exhaustive ? *to_exhaustive : 0;
  int64_t queued = 0;
  if (agg_stat)
    for (int k = 1; k <= 64; ++k) queued += agg_stat[k];
  out[PGDVS_VIEW_CNT_AGG_FP64_POINTS] = queued;
  out[PGDVS_VIEW_CNT_AGG_REFERENCE_ORDER] = agg_stat ? agg_stat[0] : 0;
  for (int k = PGDVS_VIEW_CNT_AGG_REFERENCE_ORDER + 1; k < PGDVS_VIEW_COUNTERS; ++k) out[k] = 0;
}

namespace {

struct ViewWs {
  float *cams;   // [3][80]: target, source 1, source 2
  float *times;  // [3]
  void *agg;
  int64_t agg_bytes;
  void *raster;
  int64_t raster_bytes;
  uint8_t *valid, *keep;
  float *pcl, *pts, *avg, *thres;
  int32_t *idx, *cnt;
  int32_t *chunk_cnt;  // [ceil(P / 256)] valid pixels per workgroup of dyn_warp_kernel (fused compaction)
  void *compact;
  int64_t compact_bytes;
  void *knn;
  int64_t knn_bytes;
  void *outlier;
  int64_t outlier_bytes;
  void *splat;
  int64_t splat_bytes;
  int64_t raster_rows;  // rows the tile lists are sized for
  int64_t total_bytes;
};

struct Carver {
  char *base;
  int64_t off = 0;
  template <class T>
  T *take(int64_t bytes) {
    T *p = reinterpret_cast<T *>(base + off);
    off += align_up(bytes > 0 ? bytes : 1, 256);
    return p;
  }
};

int view_layout(const pgdvs_view_geo_desc &d, void *base, ViewWs &w) {
  if (d.H <= 0 || d.W <= 0 || (int64_t)d.H * d.W >= (1ll << 31)) {
    set_error("pgdvs_view_geo: bad H/W");
    return PGDVS_ERR_INVALID;
  }
  const int64_t P = (int64_t)d.H * d.W;
  const int64_t rows = d.agg_S > 0 ? d.agg_capacity : d.st_rows;
  if (rows < 0 || rows >= (1ll << 31)) {
    set_error("pgdvs_view_geo: bad row count %lld", (long long)rows);
    return PGDVS_ERR_INVALID;
  }
  w.raster_rows = (d.row_bound > 0 && d.row_bound < rows) ? d.row_bound : rows;
  Carver c{reinterpret_cast<char *>(base)};
  w.cams = c.take<float>(3 * PGDVS_CAM_BLOCK * 4);
  w.times = c.take<float>(16);
  w.agg_bytes = 0;
  w.agg = nullptr;
  if (d.agg_S > 0) {
    w.agg_bytes = pgdvs_static_aggregate_workspace_bytes(d.agg_S, d.H, d.W, d.agg_capacity);
    if (w.agg_bytes < 0) {
      set_error("pgdvs_view_geo: bad aggregation shape");
      return PGDVS_ERR_INVALID;
    }
    w.agg = c.take<char>(w.agg_bytes);
  }
  w.raster_bytes = pgdvs_points_raster_workspace_bytes(w.raster_rows, d.H, d.W, d.radius);
  if (w.raster_bytes < 0) return (int)w.raster_bytes;  // (message set by the query)
  w.raster = c.take<char>(w.raster_bytes);
  w.valid = c.take<uint8_t>(P);
  w.keep = c.take<uint8_t>(P);
  w.pcl = c.take<float>(P * 12);
  w.splat_bytes = pgdvs_dyn_splat_workspace_bytes(d.H, d.W);
  w.splat = c.take<char>(w.splat_bytes);
  w.idx = nullptr;
  w.cnt = nullptr;
  w.chunk_cnt = nullptr;
  w.pts = w.avg = w.thres = nullptr;
  w.compact = w.knn = w.outlier = nullptr;
  w.compact_bytes = w.knn_bytes = w.outlier_bytes = 0;
  if (d.remove_outlier) {
    w.idx = c.take<int32_t>(P * 4);
    w.cnt = c.take<int32_t>(16);
    w.chunk_cnt = c.take<int32_t>(cdiv(P, 256) * 4);
    w.thres = c.take<float>(16);
    w.pts = c.take<float>(P * 12);
    w.avg = c.take<float>(P * 4);
    w.compact_bytes = compact_workspace_bytes(P);
    w.compact = c.take<char>(w.compact_bytes);
    w.knn_bytes = pgdvs_knn_workspace_bytes(P);
    w.knn = c.take<char>(w.knn_bytes);
    w.outlier_bytes = pgdvs_outlier_workspace_bytes(P);
    w.outlier = c.take<char>(w.outlier_bytes);
  }
  w.total_bytes = c.off;
  return PGDVS_OK;
}

// fork / join events of the optional side stream (timing disabled): pools behind a mutex, one per DEVICE (an event belongs to
// the device that was current when it was created; handed to a stream of another device it fails)
std::mutex g_ev_mu;
std::map<int, std::vector<hipEvent_t>> g_ev_pool;
hipEvent_t ev_get() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(g_ev_mu);
    auto &pool = g_ev_pool[dev];
    if (!pool.empty()) {
      hipEvent_t e = pool.back();
      pool.pop_back();
      return e;
    }
  }
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
  return e;
}
void ev_put(hipEvent_t e) {
  if (e == nullptr) return;  // (a failed creation is reported by the record that follows; never pooled)
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_ev_mu);
  g_ev_pool[dev].push_back(e);
}

// The dynamic branch's ~28 launches enqueued by a worker thread of the library while the calling thread enqueues the static
// branch's ~36 (round 6): at 540p x 12 frames the loop is bound by the HOST's ~4 us per launch, and a view alone starts its
// dynamic branch ~0.14 ms earlier.  One worker per process, one job at a time (callers from several host threads queue up
// behind `busy`); the job's error text travels back with its status.  (Round 3 tried the same from a second PYTHON thread and
// lost to the GIL hand-over; this one never touches the interpreter.)
class SideWorker {
 public:
  // runs `f` on the worker with `device` current; returns at once
  void submit(int device, std::function<int()> f) {
    busy_.lock();
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (!started_) {
        started_ = true;
        th_ = std::thread([this] { loop(); });
        th_.detach();  // (lives as long as the process: no join at unload, where the HIP runtime may already be gone)
      }
      job_ = std::move(f);
      device_ = device;
      has_job_ = true;
      done_ = false;
    }
    cv_.notify_all();
  }
  // waits for the job submitted last; its status (the error text is copied into the caller's)
  int wait() {
    int rc;
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [this] { return done_; });
      rc = rc_;
      if (rc != PGDVS_OK) set_error("%s", err_);
    }
    busy_.unlock();
    return rc;
  }

 private:
  void loop() {
    for (;;) {
      std::function<int()> f;
      int dev;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return has_job_; });
        f = std::move(job_);
        dev = device_;
        has_job_ = false;
      }
      int rc = PGDVS_OK;
      if (hipSetDevice(dev) != hipSuccess) {
        set_error("side worker: hipSetDevice(%d) failed", dev);
        rc = PGDVS_ERR_LAUNCH;
      } else {
        rc = f();
      }
      {
        std::lock_guard<std::mutex> lk(mu_);
        rc_ = rc;
        snprintf(err_, sizeof(err_), "%s", last_error_text());
        done_ = true;
      }
      cv_.notify_all();
    }
  }
  std::mutex busy_, mu_;
  std::condition_variable cv_;
  std::thread th_;
  std::function<int()> job_;
  bool started_ = false, has_job_ = false, done_ = false;
  int device_ = 0, rc_ = 0;
  char err_[512] = "";
};
SideWorker &side_worker() {
  static SideWorker *w = new SideWorker();  // (never destroyed: see the detach above)
  return *w;
}

std::mutex g_stat_mu;
int64_t g_stat_calls = 0;
double g_stat_seconds = 0.0;

#define VG_TRY(expr)            \
  do {                          \
    const int _rc = (expr);     \
    if (_rc != PGDVS_OK) return _rc; \
  } while (0)

// A2-A5: the dynamic branch's geometry on stream `s` (cams / times already there).
// Round 6: 28 launches instead of 42 + four memsets -- a view alone is a dependent chain of launches at ~4.7 us each and the
// host pays ~3.8 us per launch (tools/r06_latency_trace.sh), so the small jobs ride on their neighbours (fused.h): dyn_warp
// clears the keep map, the splat's flag map, the kNN state block and the statistics' histograms and counts its valid pixels per
// workgroup; ONE launch compacts, gathers the points and folds their bounding box; the three bin selections of the
// median run at the head of the passes behind them; the filter's last launch writes keep[idx[i]] itself; the projection into
// the target view runs inside the splat's flag pass.  Same bytes as the per-op entry points (tests/test_gpu_round6.py).
int dyn_geometry(const pgdvs_view_geo_desc &d, const ViewWs &w, pgdvs_stream_t s) {
  const int H = d.H, W = d.W;
  const int64_t P = (int64_t)H * W;
  hipStream_t st = as_stream(s);
  const float *cam_t = w.cams, *cam1 = w.cams + PGDVS_CAM_BLOCK, *cam2 = w.cams + 2 * PGDVS_CAM_BLOCK;
  WarpExtras ex;
  memset(&ex, 0, sizeof(ex));
  ex.zero_b = dyn_splat_flag_map(w.splat, H, W);
  unsigned *bbox = nullptr;
  int32_t *occ_count = nullptr;
  int occ_mult = 0;
  if (d.remove_outlier) {
    ex.zero_a = w.keep;
    ex.chunk_cnt = w.chunk_cnt;
    void *blk = nullptr, *tab = nullptr, *coarse = nullptr;
    int64_t bytes = 0, tab_bytes = 0, coarse_bytes = 0;
    knn_grid_state_block(w.knn, P, &blk, &bytes, &bbox, &tab, &tab_bytes, &occ_count, &occ_mult, &coarse, &coarse_bytes);
    ex.zero3 = reinterpret_cast<uint4 *>(coarse);
    ex.n16_3 = (int)(coarse_bytes / 16);
    ex.zero0 = reinterpret_cast<uint4 *>(blk);
    ex.n16_0 = (int)(bytes / 16);
    ex.zero2 = reinterpret_cast<uint4 *>(tab);
    ex.n16_2 = (int)(tab_bytes / 16);
    outlier_hist_block(w.outlier, &blk, &bytes);
    ex.zero1 = reinterpret_cast<uint4 *>(blk);
    ex.n16_1 = (int)(bytes / 16);
  }
  PGDVS_REQUIRE(d.dyn_mask1 && d.flow12 && d.depth1 && d.depth2 && d.rgb1 && d.rgb2, "pgdvs_view_geo_forward: null input pointer");
  VG_TRY(dyn_warp_fused(H, W, d.dyn_mask1, d.flow_occ, d.use_flow_consistency, d.flow12, d.depth1, d.depth2, d.rgb1, d.rgb2, cam1,
                        cam2, w.times, nullptr, w.valid, w.pcl, nullptr, ex, st));  // (no effective-mask map, no frame-2 colours: the splat path reads neither)
  const uint8_t *keep = w.valid;
  if (d.remove_outlier) {
    // pytorch3d's kNN + the statistical filter (pgdvs_renderer_dyn.py:401-457)
    VG_TRY(compact_gather_bbox(w.valid, P, w.chunk_cnt, w.idx, w.cnt, w.pcl, w.pts, bbox, occ_count, occ_mult, st));
    VG_TRY(knn_grid_mean_dist_prepared(w.pts, w.cnt, P, d.outlier_knn, w.avg, w.knn, w.knn_bytes, st));
    VG_TRY(outlier_keep_fused(w.avg, w.cnt, P, d.outlier_std_thres, w.thres, w.idx, w.keep, w.outlier, w.outlier_bytes, st));
    keep = w.keep;
  }
  // A5 + A6-A8, first half: projection, metric, flags and the scatter into the accumulators need the flows, not the static
  // image -- they run here, beside the static branch when there is a side stream; only the finish pass waits for the rasteriser
  const unsigned long long *rng = (d.noise == nullptr && d.rng_state != nullptr) ? reinterpret_cast<const unsigned long long *>(d.rng_state) : nullptr;
  VG_TRY(dyn_splat_scatter_part_fused(H, W, d.rgb1, d.rgb2, d.flow12, cam_t, w.pcl, keep, nullptr, nullptr, d.noise, rng, d.alpha,
                                      w.splat, true, st));  // (the flow and validity planes of A5 are not materialised)
  return PGDVS_OK;
}

int view_forward(const pgdvs_view_geo_desc &d, void *workspace, int64_t workspace_bytes, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(d.flat_cam_tgt && d.flat_cam_src && d.time_src && d.time_tgt && d.rgb1 && d.rgb2 && d.depth1 && d.depth2 &&
                    d.dyn_mask1 && d.flow12,
                "pgdvs_view_geo_forward: null input pointer");
  PGDVS_REQUIRE(d.static_rgb && d.static_mask && d.render_dyn_rgb && d.render_dyn_mask && d.combined && d.combined_static &&
                    d.combined_dyn && d.raster_status,
                "pgdvs_view_geo_forward: null output pointer");
  PGDVS_REQUIRE(!d.use_flow_consistency || d.flow_occ, "pgdvs_view_geo_forward: flow_occ required");
  PGDVS_REQUIRE(d.agg_S > 0 || d.st_rows == 0 || d.st_pcl_rgb, "pgdvs_view_geo_forward: no static cloud");
  PGDVS_REQUIRE(d.agg_S <= 0 || (d.agg_rgbs && d.agg_depths && d.agg_masks && d.agg_K3s_host && d.agg_c2ws_host &&
                                  d.agg_cloud_out && d.agg_xyz_out && d.agg_count_out && d.agg_capacity > 0),
                "pgdvs_view_geo_forward: incomplete aggregation arguments");
  PGDVS_REQUIRE(d.alpha >= 0.0f, "pgdvs_view_geo_forward: alpha must be >= 0");
  PGDVS_REQUIRE(!d.remove_outlier || (d.outlier_knn >= 1 && d.outlier_knn + 1 <= 64),
                "pgdvs_view_geo_forward: dyn_pcl_outlier_knn must be in [1, 63]");
  ViewWs w;
  VG_TRY(view_layout(d, workspace, w));
  if (!workspace || workspace_bytes < w.total_bytes) {
    set_error("pgdvs_view_geo_forward: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)w.total_bytes);
    return PGDVS_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const int H = d.H, W = d.W;
  VG_TRY(view_prep(d.flat_cam_tgt, d.flat_cam_src, d.time_src, d.time_tgt, w.cams, w.times, st));
  // ---- dynamic branch geometry (it depends on nothing the static branch makes): on the caller's stream ahead of the
  // static branch, or -- with a side stream -- forked off here and enqueued BEHIND the static branch's launches (the
  // longer of the two chains: its first kernels should not wait for the host to have enqueued the other ~45)
  hipStream_t side = as_stream(d.side_stream);
  const bool forked = side != nullptr && side != st;
  hipEvent_t ev_join = nullptr;
  if (forked) {
    hipEvent_t ev_fork = ev_get();
    hipError_t e = hipEventRecord(ev_fork, st);
    if (e == hipSuccess) e = hipStreamWaitEvent(side, ev_fork, 0);
    ev_put(ev_fork);
    if (e != hipSuccess) {
      set_error("pgdvs_view_geo_forward: fork onto the side stream: %s", hipGetErrorString(e));
      return PGDVS_ERR_LAUNCH;
    }
  } else {
    VG_TRY(dyn_geometry(d, w, stream));
  }
  // (with a side stream the dynamic branch is enqueued by the worker thread while this one goes on with the static branch;
  // option side_thread = 0: by this thread, behind the static branch)
  const bool on_worker = forked && option_int(options().side_thread) != 0;
  if (on_worker) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const pgdvs_view_geo_desc *dp = &d;
    const ViewWs *wp = &w;
    side_worker().submit(dev, [dp, wp] { return dyn_geometry(*dp, *wp, dp->side_stream); });
  }
  // ---- static branch: A12 (optional) + A9
  const float *cloud = d.st_pcl_rgb, *xyz = d.st_pcl_xyz;
  const int64_t *count_dev = d.st_count_dev;
  int64_t rows = d.st_rows;
  int rc = PGDVS_OK;
  bool counters_cleared = false;
  if (d.agg_S > 0) {
    // (the aggregation's last launch clears the counters the rasteriser starts from: one memset less on this chain)
    void *cblk = nullptr;
    int64_t cbytes = 0;
    raster_counter_block(w.raster, w.raster_rows, H, W, d.radius, &cblk, &cbytes);
    rc = static_aggregate_for_view(d.agg_rgbs, d.agg_depths, d.agg_masks, d.agg_K3s_host, d.agg_c2ws_host, d.agg_S, H, W,
                                   d.agg_cloud_out, d.agg_xyz_out, d.agg_capacity, d.agg_count_out, w.agg, w.agg_bytes, stream,
                                   d.agg_params_cached != 0, cblk, cbytes);
    counters_cleared = true;  // (by agg_rows, or -- ordered chain, single frame -- by a memset of the aggregation)
    cloud = d.agg_cloud_out;
    xyz = d.agg_xyz_out;
    count_dev = d.agg_count_out;
    rows = d.agg_capacity;
  }
  if (rc == PGDVS_OK) {
    const float *pts = xyz ? xyz : cloud;
    const int64_t pts_stride = xyz ? 3 : 6;
    const float *feat = cloud ? cloud + 3 : nullptr;
    rc = points_raster_bounded_cleared(pts, pts_stride, feat, 6, rows, count_dev, w.raster_rows, d.raster_status, w.cams, d.radius,
                                       d.K, H, W, d.static_rgb, 1, d.static_mask, w.raster, w.raster_bytes, stream, counters_cleared);
  }
  if (forked) {
    // (whatever happened above, the side stream is joined: its work must be ordered before the caller's next use of `st`)
    const int rc_dyn = on_worker ? side_worker().wait() : dyn_geometry(d, w, d.side_stream);
    ev_join = ev_get();
    hipError_t e = hipEventRecord(ev_join, side);
    if (e == hipSuccess) e = hipStreamWaitEvent(st, ev_join, 0);
    ev_put(ev_join);
    if (rc == PGDVS_OK) rc = rc_dyn;
    if (e != hipSuccess && rc == PGDVS_OK) {
      set_error("pgdvs_view_geo_forward: join: %s", hipGetErrorString(e));
      rc = PGDVS_ERR_LAUNCH;
    }
  }
  VG_TRY(rc);
  // ---- A8 second half + A11: normalise, threshold, composite over the static image (and advance the noise's draw number)
  unsigned long long *rng = (d.noise == nullptr && d.rng_state != nullptr) ? reinterpret_cast<unsigned long long *>(d.rng_state) : nullptr;
  return dyn_splat_finish_part(H, W, rng, d.static_rgb, d.render_dyn_rgb, d.render_dyn_mask, d.combined, d.combined_static,
                               d.combined_dyn, w.splat, st);
}

}  // namespace
}  // namespace pgdvs

using namespace pgdvs;

PGDVS_API int64_t pgdvs_view_geo_desc_size(void) { return (int64_t)sizeof(pgdvs_view_geo_desc); }

PGDVS_API int64_t pgdvs_view_geo_workspace_bytes(const pgdvs_view_geo_desc *desc) {
  if (!desc) {
    set_error("pgdvs_view_geo_workspace_bytes: null description");
    return PGDVS_ERR_INVALID;
  }
  ViewWs w;
  const int rc = view_layout(*desc, nullptr, w);
  return rc != PGDVS_OK ? (int64_t)rc : w.total_bytes;
}

PGDVS_API int pgdvs_view_geo_forward(const pgdvs_view_geo_desc *desc, void *workspace, int64_t workspace_bytes,
                                     pgdvs_stream_t stream) {
  PGDVS_REQUIRE(desc, "pgdvs_view_geo_forward: null description");
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = view_forward(*desc, workspace, workspace_bytes, stream);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  {
    std::lock_guard<std::mutex> lk(g_stat_mu);
    g_stat_calls += 1;
    g_stat_seconds += dt;
  }
  return rc;
}

PGDVS_API int pgdvs_view_geo_counters(const pgdvs_view_geo_desc *desc, const void *workspace, int64_t workspace_bytes,
                                      int64_t *counters_dev, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(desc && counters_dev, "pgdvs_view_geo_counters: null argument");
  ViewWs w;
  VG_TRY(view_layout(*desc, const_cast<void *>(workspace), w));
  if (!workspace || workspace_bytes < w.total_bytes) {
    set_error("pgdvs_view_geo_counters: not the workspace of this description");
    return PGDVS_ERR_WORKSPACE;
  }
  const pgdvs_view_geo_desc &d = *desc;
  hipStream_t st = as_stream(stream);
  raster_counters(w.raster, w.raster_rows, d.H, d.W, d.radius, counters_dev + PGDVS_VIEW_CNT_RASTER_ENTRIES, st);
  const int32_t *to_ring = nullptr, *to_coarse = nullptr, *to_exh = nullptr;
  if (d.remove_outlier) knn_grid_counter_words(w.knn, (int64_t)d.H * d.W, 0, &to_ring, &to_coarse, &to_exh);
  const unsigned *agg_stat = d.agg_S > 0 ? agg_stat_words(w.agg, d.agg_S, d.H, d.W, d.agg_capacity) : nullptr;
  PGDVS_LAUNCH("view_counters", view_counters_kernel, dim3(1), dim3(64), 0, st,
               d.agg_S > 0 ? (const int64_t *)d.agg_count_out : d.st_count_dev, d.st_rows, (const int32_t *)w.cnt, to_ring,
               to_coarse, to_exh, agg_stat, counters_dev);
  return check_launch("view_geo_counters");
}

PGDVS_API void pgdvs_view_geo_host_stats(int64_t *calls, double *seconds) {
  std::lock_guard<std::mutex> lk(g_stat_mu);
  if (calls) *calls = g_stat_calls;
  if (seconds) *seconds = g_stat_seconds;
  g_stat_calls = 0;
  g_stat_seconds = 0.0;
}
