/*
 * pgdvs_hip.h -- C ABI of libpgdvs_hip.so: the MI355X (gfx950) implementation of
 * the PGDVS per-target-view rendering inner loop.
 *
 * The reference (apple/ml-pgdvs) has no C FFI: its hot path is Python/torch plus
 * three CUDA kernels embedded as strings (pgdvs/utils/softsplat.py) and the
 * un-vendored pytorch3d ops.  Each entry point below names the reference
 * function (file:line, relative to the upstream tree) it replaces; the Python
 * host layer (ml-pgdvs_amd/pgdvs_amd) binds them with ctypes and keeps the
 * reference's renderer plugin API on top.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter is documented "host";
 *   - all buffers are caller-allocated, contiguous, fp32 unless stated otherwise;
 *   - `stream` is a hipStream_t (NULL = default stream); every call only enqueues
 *     work on it and never synchronises or allocates (the diagnostics option knn_stats
 *     does synchronise);
 *   - return value: 0 on success, negative pgdvs_status on error, message via
 *     pgdvs_last_error() (thread-local);
 *   - re-entrant per stream.  Process-wide state is limited to (i) the options below, (ii) the
 *     opt-in profiling records of pgdvs_prof_*, (iii) per-device pools of fork / join events
 *     and the host-side statistics of pgdvs_view_geo_host_stats (mutex-protected).
 *
 * Options: a handful of process-wide switches, read from the ENVIRONMENT ONCE, when the
 * library is loaded, and changed afterwards only through pgdvs_option_set (never by a later
 * setenv: no entry point calls getenv).  An entry point reads the options it needs once, at
 * its start.  Results are identical for every setting (bit for bit for the index paths).
 *   name                  environment at load          meaning
 *   agg_ordered           PGDVS_AGG_ORDERED=1          A12 as the ordered chain of round 2: per frame an ordered
 *                                                      selection + a push that builds the rows (second implementation)
 *   agg_stage             PGDVS_AGG_STAGE=0 -> 0       0: A12's links leave no (depth, colour) rows, agg_rows gathers
 *                                                      them itself (the path of videos too long for the staging block)
 *   gnt_fp32              PGDVS_GNT_FP32=1             every GNT product on the fp32 matrix instruction (default:
 *                                                      exact bf16x3 products where they pay, see pgdvs_gnt_view_layer)
 *   raster_bound_density  PGDVS_RASTER_BOUND_DENSITY   rows per pixel from which the rasteriser computes its depth
 *                                                      bound and runs its long-list launch (default 2.2)
 *   knn_no_tpq            PGDVS_KNN_NO_TPQ=1           diagnostics: the wavefront-per-query search for every query
 *   knn_stats             PGDVS_KNN_STATS=1            diagnostics: ring histogram to stderr (synchronises)
 *   side_thread           PGDVS_SIDE_THREAD=0 -> 0     0: pgdvs_view_geo_forward enqueues the dynamic branch (side stream) from the
 *                                                      calling thread behind the static branch, as in rounds 4-5; default 1: a
 *                                                      worker thread of the library enqueues it while the caller enqueues the static
 *                                                      branch (the host's ~60 launches per view in two halves side by side)
 *
 * Camera block: 80 floats of derived per-camera constants produced by
 * pgdvs_cam_prep from the reference's flat_cam[34] = [h, w, K(4x4), c2w(4x4)]
 * (pgdvs/renderers/pgdvs_renderer.py:354-357).  Layout: see PGDVS_CAM_* below.
 */
#ifndef PGDVS_HIP_H_
#define PGDVS_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *pgdvs_stream_t; /* hipStream_t */

enum pgdvs_status {
  PGDVS_OK = 0,
  PGDVS_ERR_INVALID = -1,   /* bad argument */
  PGDVS_ERR_LAUNCH = -2,    /* HIP launch / runtime error */
  PGDVS_ERR_WORKSPACE = -3, /* workspace too small */
  PGDVS_ERR_UNSUPPORTED = -4
};

#define PGDVS_CAM_KINV 0 /* [9]  inverse(K[:3,:3])                 */
#define PGDVS_CAM_M 9    /* [9]  c2w[:3,:3] @ Kinv                 */
#define PGDVS_CAM_O 18   /* [3]  c2w[:3,3]                         */
#define PGDVS_CAM_P 21   /* [16] K(4x4) @ inverse(c2w)             */
#define PGDVS_CAM_W2C 37 /* [16] inverse(c2w)                      */
#define PGDVS_CAM_R 53   /* [9]  c2w[:3,:3]                        */
#define PGDVS_CAM_HW 62  /* [2]  h, w                              */
#define PGDVS_CAM_K 64   /* [16] K as given                        */
#define PGDVS_CAM_BLOCK 80

const char *pgdvs_last_error(void);

/* Options (see the table at the top).  pgdvs_option_set: 0, or PGDVS_ERR_INVALID for an unknown name; flags take
 * value != 0.  pgdvs_option_get: the current value, NaN for an unknown name. */
int pgdvs_option_set(const char *name, double value);
double pgdvs_option_get(const char *name);
/* library/ABI version and the gfx target the device code was built for */
int pgdvs_abi_version(void);
const char *pgdvs_build_arch(void);

/* Optional per-kernel timing: when enabled every kernel launch is bracketed by HIP events
 * on its launch stream.  pgdvs_prof_report synchronises them, writes "name calls total_ms"
 * lines into buf and clears the records; returns the number of distinct names.  Process-
 * wide switch meant for bench.py; leave it off in production. */
void pgdvs_prof_enable(int on);
int pgdvs_prof_report(char *buf, int buf_len);
/* the fixed cost of one (event, launch, event) bracket that pgdvs_prof_report subtracts from
 * every record (measured once with an empty kernel) */
double pgdvs_prof_overhead_ms(void);

/* ---- cameras ------------------------------------------------------------- */
/* flat_cams[n,34] -> cam_blocks[n,80].  Replaces the torch.inverse / bmm chains of
 * pgdvs/renderers/pgdvs_renderer_base.py:40-45 and pgdvs/models/gnt/projector.py:49-60. */
int pgdvs_cam_prep(const float *flat_cams, int n, float *cam_blocks, pgdvs_stream_t stream);

/* A1: PGDVSBaseRenderer.get_batched_rays, batch_size=1
 * (pgdvs/renderers/pgdvs_renderer_base.py:17-57).  n = ceil(H/stride)*ceil(W/stride);
 * rays_o[n,3] rays_d[n,3] uvs[n,2]. */
int pgdvs_get_rays(const float *cam_block, int H, int W, int stride, float *rays_o,
                   float *rays_d, float *uvs, pgdvs_stream_t stream);

/* ---- dynamic branch -------------------------------------------------------- */
/* A2+A3: unproject frame 1, follow the flow into frame 2, unproject there, lerp in
 * time -- the dense part of PGDVSDynamicRenderer.compute_dyn_pcl
 * (pgdvs/renderers/pgdvs_renderer_dyn.py:299-388).
 *   dyn_mask1[H,W] occ[H,W] flow12[H,W,2] depth1[H,W] depth2[H,W] rgb1[H,W,3] rgb2[H,W,3]
 *   times[3] = (time_1, time_2, time_tgt) on the device
 *   mask_eff[H,W] u8 : dyn mask after the optional flow-consistency test (:304-308)
 *   valid[H,W]    u8 : mask_eff && flow target inside the image (:309-316)
 *   pcl[H,W,3], rgbf[H,W,3] : world point / attached colour, written where valid. */
int pgdvs_dyn_warp(int H, int W, const float *dyn_mask1, const float *occ,
                   int use_flow_consistency, const float *flow12, const float *depth1,
                   const float *depth2, const float *rgb1, const float *rgb2,
                   const float *cam1, const float *cam2, const float *times,
                   uint8_t *mask_eff, uint8_t *valid, float *pcl, float *rgbf,
                   pgdvs_stream_t stream);

/* Ordered stream compaction: idx_out[0..count) = ascending positions p with flags[p]!=0
 * (the boolean-mask indexing / torch.nonzero of pgdvs_renderer_dyn.py:309-320,477).
 * count_out: one int32 on the device.  workspace >= pgdvs_compact_workspace_bytes(n). */
int64_t pgdvs_compact_workspace_bytes(int64_t n);
int pgdvs_compact_u8(const uint8_t *flags, int64_t n, int32_t *idx_out, int32_t *count_out,
                     void *workspace, int64_t workspace_bytes, pgdvs_stream_t stream);

/* gather rows: dst[i,:] = src[idx[i],:] for i < *count (row = `width` floats). */
int pgdvs_gather_rows(const float *src, const int32_t *idx, const int32_t *count,
                      int64_t capacity, int width, float *dst, pgdvs_stream_t stream);

/* A4: statistical outlier filter = pytorch3d.ops.knn_points(X, X, K+1) + mean of the K
 * non-self squared distances (pgdvs_renderer_dyn.py:405-419, st_geo_renderer.py:37-51).
 * pts[capacity,3], *count points used (count on device); avg_out[capacity].
 * algo: 0 = auto, 1 = brute force (O(N^2), what pytorch3d does), 2 = exact uniform-grid
 * search (needs K+1 <= 64; two grid levels, then an exhaustive scan for what is still open).
 * Both return identical values.  workspace >= pgdvs_knn_workspace_bytes(capacity). */
int64_t pgdvs_knn_workspace_bytes(int64_t capacity);
int pgdvs_knn_mean_dist(const float *pts, const int32_t *count, int64_t capacity, int K,
                        float *avg_out, int algo, void *workspace, int64_t workspace_bytes,
                        pgdvs_stream_t stream);

/* threshold = lower-median(avg) + unbiased-std(avg) * std_thres; flag = avg < threshold
 * (pgdvs_renderer_dyn.py:419-427).  thres_out: 1 float; flag_out[capacity] u8.
 * remove_outlier == 0 -> all flags 1 (:453-457), threshold still produced.  flag_out[i] = 0
 * for *count <= i < capacity. */
int64_t pgdvs_outlier_workspace_bytes(int64_t capacity);
int pgdvs_outlier_flags(const float *avg, const int32_t *count, int64_t capacity,
                        float std_thres, int remove_outlier, float *thres_out,
                        uint8_t *flag_out, void *workspace, int64_t workspace_bytes,
                        pgdvs_stream_t stream);

/* keep[P] u8 <- 0 everywhere, 1 at idx[i] where flag[i] (i < *count). */
int pgdvs_scatter_keep(const int32_t *idx, const uint8_t *flag, const int32_t *count,
                       int64_t capacity, uint8_t *keep, int64_t P, pgdvs_stream_t stream);

/* A5: project the surviving points into the target camera and scatter the dense
 * flow (pgdvs/models/gnt/projector.py:41-73 via pgdvs_renderer_dyn.py:470-503).
 * flow_1_to_tgt[2,H,W] (planar x then y), valid_dyn_mask_1[H,W] (0/1 floats). */
int pgdvs_project_flow_dense(int H, int W, const float *cam_tgt, const float *pcl,
                             const uint8_t *keep, float *flow_1_to_tgt,
                             float *valid_dyn_mask_1, pgdvs_stream_t stream);
/* sparse form: uv[n,2] = projection of pts[n,3] (Projector.compute_projections). */
int pgdvs_project_points(const float *cam_tgt, const float *pts, int64_t n, float *uv,
                         pgdvs_stream_t stream);

/* A6: softsplat importance metric, mean_c |rgb1 - backwarp(rgb2, flow)|
 * (pgdvs/renderers/pgdvs_renderer_base.py:68-78,91-138).  NCHW planar:
 * rgb1[B,3,H,W] rgb2[B,3,H,W] flow[B,2,H,W] -> l1[B,1,H,W]. */
int pgdvs_backwarp_l1(const float *rgb1, const float *rgb2, const float *flow, float *l1,
                      int B, int H, int W, pgdvs_stream_t stream);

/* A7: softsplat.softsplat / kernel softsplat_out (pgdvs/utils/softsplat.py:280-333,
 * 352-402).  in[B,C,H,W] flow[B,2,H,W] metric[B,1,H,W] (NULL for sum/avg) -> out[B,C,H,W].
 * mode: 0 sum, 1 avg, 2 linear, 3 soft; eps: 0 addeps (default), 1 zeroeps, 2 clipeps.
 * workspace >= pgdvs_softsplat_workspace_bytes(B,C,H,W,mode). */
int64_t pgdvs_softsplat_workspace_bytes(int B, int C, int H, int W, int mode);
int pgdvs_softsplat_fwd(const float *in, const float *flow, const float *metric, float *out,
                        int B, int C, int H, int W, int mode, int eps, void *workspace,
                        int64_t workspace_bytes, pgdvs_stream_t stream);

/* Backward of the raw splat (mode "sum"; softsplat.py:459-617, kernels softsplat_ingrad /
 * softsplat_flowgrad): ingrad[B,C,H,W] and flowgrad[B,2,H,W] (either nullable) from
 * outgrad[B,C,H,W].  The normalised modes differentiate through their torch pre/post-processing
 * exactly as upstream (softsplat.py:294-333). */
int pgdvs_softsplat_bwd(const float *in, const float *flow, const float *outgrad, float *ingrad,
                        float *flowgrad, int B, int C, int H, int W, pgdvs_stream_t stream);

/* A6+A7+A8+A11 fused for the renderer: noise-fill of static texels, metric, soft
 * splat of rgb and mask with the shared metric, threshold 1e-3, masking and the
 * final static/dynamic composite (pgdvs_renderer_dyn.py:157-202, pgdvs_renderer.py:169-178).
 *   rgb1[H,W,3] rgb2[H,W,3] (channels-last, as in the data dict), flow12[H,W,2],
 *   flow_1_to_tgt[2,H,W], valid_dyn_mask_1[H,W], noise[3,H,W] (un-clamped randn, may be NULL = 0),
 *   static_rgb[3,H,W] (may be NULL -> combined outputs skipped)
 *   outputs (planar): render_dyn_rgb[3,H,W] render_dyn_mask[H,W]
 *                     combined[3,H,W] combined_static[3,H,W] combined_dyn[3,H,W] (nullable)
 *   workspace >= pgdvs_dyn_splat_workspace_bytes(H,W). */
int64_t pgdvs_dyn_splat_workspace_bytes(int H, int W);
int pgdvs_dyn_splat_composite(int H, int W, const float *rgb1, const float *rgb2,
                              const float *flow12, const float *flow_1_to_tgt,
                              const float *valid_dyn_mask_1, const float *noise, float alpha,
                              const float *static_rgb, float *render_dyn_rgb,
                              float *render_dyn_mask, float *combined, float *combined_static,
                              float *combined_dyn, void *workspace, int64_t workspace_bytes,
                              pgdvs_stream_t stream);

/* The same with the noise drawn inside the scatter kernel where it is consumed (upstream: torch.randn_like per forward,
 * pgdvs_renderer_dyn.py:177-182): rng_state = DEVICE uint64[2] {seed, draw number}; the field is a pure function of
 * (seed, draw number, pixel) -- Philox4x32-10 + Box-Muller -- and the call increments the draw number, so a replayed
 * HIP graph draws a fresh field per replay.  pgdvs_splat_noise_field writes the field [3,H,W] the NEXT such call will
 * use (tests: the injected-noise entry point fed with it gives the same images). */
int pgdvs_dyn_splat_composite_rng(int H, int W, const float *rgb1, const float *rgb2, const float *flow12,
                                  const float *flow_1_to_tgt, const float *valid_dyn_mask_1, uint64_t *rng_state,
                                  float alpha, const float *static_rgb, float *render_dyn_rgb, float *render_dyn_mask,
                                  float *combined, float *combined_static, float *combined_dyn, void *workspace,
                                  int64_t workspace_bytes, pgdvs_stream_t stream);
int pgdvs_splat_noise_field(int H, int W, const uint64_t *rng_state, float *noise_out, pgdvs_stream_t stream);

/* ---- static branch --------------------------------------------------------- */
/* A9: pytorch3d PointsRasterizer(bin_size=0) + PointsRenderer + NormWeightedCompositor
 * as used by StaticGeoPointRenderer.forward (pgdvs/renderers/st_geo_renderer.py:77-120)
 * and render_dyn_pcl (pgdvs/renderers/pgdvs_renderer_dyn.py:671-724).
 *   points: xyz at pts[i*pts_stride..+3), features at feat[i*feat_stride..+3)
 *   n_points: host count; n_points_dev (nullable): device int64 count that overrides
 *   it (n_points is then the capacity: a larger or negative device count is clamped to [0, n_points]).
 *   Returns PGDVS_ERR_UNSUPPORTED (and the workspace query a negative size) when n_points times the
 *   tiles a disc of this radius can touch reaches 2^31 list entries.
 *   Workspace: 16 bytes per list entry the rows can produce (n_points x the tiles a disc can touch), 4 bytes per pixel, and --
 *   images of at least 16 tiles -- one segment per tile for the direct binning pass: twice the average list the rows allow,
 *   256 .. 4096 entries (0.53 GB at 1080p).  Sparse clouds (fewer rows than option raster_bound_density x pixels) are
 *   binned straight into the segments, without a counting pass; an entry that finds its segment full raises a device flag
 *   and the exact counting / scan / fill passes, enqueued behind the direct one either way, redo the binning.
 *   outputs (any may be NULL): idx[H,W,K] int64 (-1 pad), zbuf[H,W,K] (-1 pad),
 *   dist2[H,W,K] (-1 pad), rgb ([H,W,3] if rgb_planar == 0, [3,H,W] otherwise),
 *   mask[H,W] ((ones-render) > 0 as 0/1 floats).  K (points_per_pixel) in [1, 8]. */
int64_t pgdvs_points_raster_workspace_bytes(int64_t n_points, int H, int W, float radius);
int pgdvs_points_raster(const float *pts, int64_t pts_stride, const float *feat,
                        int64_t feat_stride, int64_t n_points, const int64_t *n_points_dev,
                        const float *cam_tgt, float radius, int K, int H, int W, int64_t *idx,
                        float *zbuf, float *dist2, float *rgb, int rgb_planar, float *mask,
                        void *workspace, int64_t workspace_bytes, pgdvs_stream_t stream);

/* The same with a workspace sized for `row_bound` rows -- pgdvs_points_raster_workspace_bytes(row_bound, ...) -- instead
 * of the arrays' capacity n_points (callers keep capacity-sized cloud buffers with a device-side count; a hint such as
 * the previous view's count plus a margin saves gigabytes of tile lists per view in flight).  status_dev: DEVICE int32,
 * written by the call: 0 = all rows drawn, 1 = the device count exceeds row_bound (the rows beyond it are not drawn: the
 * images are not valid), 2 = the device count is negative (the producer's error status; nothing drawn). */
int pgdvs_points_raster_bounded(const float *pts, int64_t pts_stride, const float *feat, int64_t feat_stride,
                                int64_t n_points, const int64_t *n_points_dev, int64_t row_bound, int32_t *status_dev,
                                const float *cam_tgt, float radius, int K, int H, int W, int64_t *idx, float *zbuf,
                                float *dist2, float *rgb, int rgb_planar, float *mask, void *workspace,
                                int64_t workspace_bytes, pgdvs_stream_t stream);

/* A12: static point-cloud aggregation across the S frames of a video with the
 * projection-occupancy dedup (pgdvs/datasets/nvidia_eval_pure_geo.py:183-277,
 * pgdvs/datasets/nvidia_eval.py:840-847, pgdvs/datasets/base.py:507-546).
 *   rgbs[S,H,W,3] in [0,1]; depths[S,H,W]; dyn_masks[S,H,W] u8 (non-zero = dynamic)
 *   K3s: HOST double[S,9]; c2ws: HOST double[S,16]   (float64 numpy upstream)
 *   out[capacity,6] (xyz,rgb) in the reference's order; count_out: device int64 = the number
 *   of rows written, or -1 if the kernels' internal ordering protocol reported an error (the rows are
 *   then not valid; pgdvs_points_raster treats a negative device count as 0).  A count EQUAL to `capacity` means the
 *   cloud did not fit: rows were dropped (which ones is unspecified beyond frame 0's prefix) and the cloud must not be
 *   used -- size the buffer so that the count stays below it (S*H*W rows always suffice).
 *   The workspace holds one occupancy byte per (frame, pixel), the later frames' selection bits, 12 bytes per row of
 *   capacity and -- unless it would exceed 4 GB -- a staging block of 16 bytes per (frame, pixel) in which the chain's links
 *   leave (depth, colour) of the pixels they select (address space: a few per cent of it is ever touched, but all of it is
 *   part of the size this query returns: 0.85 GB at 1080p x 24 frames, 1.7 GB at x 48). */
int64_t pgdvs_static_aggregate_workspace_bytes(int S, int H, int W, int64_t capacity);
int pgdvs_static_aggregate(const float *rgbs, const float *depths, const uint8_t *dyn_masks,
                           const double *K3s_host, const double *c2ws_host, int S, int H, int W,
                           float *out, int64_t capacity, int64_t *count_out, void *workspace,
                           int64_t workspace_bytes, pgdvs_stream_t stream);
/* The same, and xyz_out[capacity,3] receives the coordinates alone (12 bytes per point): pass it as `pts` with
 * pts_stride 3 (and out + 3 with stride 6 as `feat`) to pgdvs_points_raster, whose binning passes then read half
 * the bytes.  Same workspace size. */
int pgdvs_static_aggregate_packed(const float *rgbs, const float *depths, const uint8_t *dyn_masks,
                                  const double *K3s_host, const double *c2ws_host, int S, int H, int W,
                                  float *out, float *xyz_out, int64_t capacity, int64_t *count_out,
                                  void *workspace, int64_t workspace_bytes, pgdvs_stream_t stream);

/* ---- GNT static renderer ---------------------------------------------------- */
/* A13: ray sampling + Projector.compute (pgdvs/models/gnt/ray_sampler.py:59-123,
 * pgdvs/models/gnt/projector.py:41-115,117-308) for R rays x S samples x V source views.
 *   ray_o/ray_d[R,3]; depth_range[1,2] or [R,2] (depth_range_per_ray); deterministic sampling,
 *   inverse-depth uniform when inv_uniform != 0; z_samples[R,S] (nullable) = explicit sample
 *   depths instead (the importance-resampled fine pass, ray_sampler.py:183-220)
 *   cam_tgt: camera block of the target; cams_src[V,80]; src_rgbs[V,H,W,3];
 *   featmaps_cl[V,hf,wf,C] (channels-last); inv_masks[V,H,W] (nullable: dynamic masks)
 *   outputs: pts[R,S,3] z_vals[R,S] (nullable), rgb_feat[R,S,V,3+C], ray_diff[R,S,V,4],
 *   mask_inbound / mask_invalid (nullable) / mask [R,S,V] as 0/1 floats. */
int pgdvs_gnt_gather(const float *ray_o, const float *ray_d, const float *depth_range,
                     int depth_range_per_ray, const float *z_samples, int R, int S, int inv_uniform,
                     const float *cam_tgt,
                     const float *cams_src, int V, const float *src_rgbs, int H, int W,
                     const float *featmaps_cl, int hf, int wf, int C, const float *inv_masks,
                     float *pts, float *z_vals, float *rgb_feat, float *ray_diff,
                     float *mask_inbound, float *mask_invalid, float *mask, pgdvs_stream_t stream);

/* A14 (entry of GNT.forward, pgdvs/models/gnt/models/transformer_network.py:455-474):
 * feat = rgbfeat_fc(rgb_feat) (Linear(3+C,64) -> ReLU -> Linear(64,64)) on the fp32 MFMA, fused
 * with the reductions over the source views that follow it upstream.
 *   weights: pgdvs_gnt_embed_weight_floats(Cin) floats = W1 input-major [4*ceil(Cin/4)][64]
 *            (rows >= Cin zero), b1[64], W2 input-major [64][64], b2[64]
 *            (pgdvs_amd.ops.pack_embed); Cin = 3 + C must be in (32, 36]
 *   rgb_feat[N,V,Cin] -> feat[N,V,64]; q0[N,64] = max over the V views (:458);
 *   stats[N,2] (nullable) = mean over features of the unbiased std over views and of
 *   std / (mean |feat| + 1e-6) (:464-472; all views, no mask). */
int64_t pgdvs_gnt_embed_weight_floats(int Cin);
int pgdvs_gnt_embed(const float *weights, const float *rgb_feat, int64_t N, int V, int Cin,
                    float *feat, float *q0, float *stats, pgdvs_stream_t stream);

/* A14, positional re-embedding of the even layers (transformer_network.py:482-486):
 * q <- q_fc(cat(q, posenc(pts), posenc(viewdir))) with q_fc = Linear(64+P+P',64) -> ReLU ->
 * Linear(64,64).  The caller forms the position part T[N,*] = posenc(pts) W1[:,64:64+P]^T and the
 * direction part tv[R,*] = posenc(viewdir) W1[:,64+P:]^T + b1 (row strides in floats, multiples
 * of 4, so that the slices of all even layers can share one GEMM); the kernel computes
 * q_out = W2 relu(W1[:, :64] q + T[g] + tv[g / S]) + b2.
 *   weights: W1[:, :64] input-major [64][64], W2 input-major [64][64], b2[64]. */
int pgdvs_gnt_posfc(const float *weights, const float *q_in, const float *T, int64_t t_stride,
                    const float *tv, int64_t tv_stride, int64_t N, int S, float *q_out,
                    pgdvs_stream_t stream);

/* A14, exit of GNT.forward (transformer_network.py:533-535):
 * rgb_out[R,3] = rgb_fc(mean over the S samples of LayerNorm(q[R,S,64])), eps 1e-5.
 *   weights: gamma[64], beta[64], rgb_fc weight [3][64], rgb_fc bias[3]. */
int pgdvs_gnt_head(const float *weights, const float *q, int R, int S, float *rgb_out,
                   pgdvs_stream_t stream);

/* A14 (view transformer): one fused fp32-MFMA kernel per GNT layer = Transformer2D +
 * Attention2D of pgdvs/models/gnt/models/transformer_network.py:59-169,197-223 (width 64).
 *   weights: pgdvs_gnt_view_weight_floats() floats, packed input-major as laid out in
 *            csrc/gnt_view.hip (VW_* offsets); built by pgdvs_amd.ops.pack_view_layer
 *   q_in[N,64]; feat[N,V,64] (rgbfeat_fc output); ray_diff[N,V,4]; valid[N,V] u8 (rows
 *   without any valid view must be passed as all-valid, :124-129); q_out[N,64]
 *   stats[N,3] (nullable): view entropy (:497-500, evaluated online as log l - sum e a / l, within
 *   2e-7 of the upstream expression), masked std of k, normalised std; means over features.
 *   Arithmetic: fp32 inputs, weights and results; the 64 x 64 products of the attention (k = Wk f and vv = Wv k per source view,
 *   q' and out_fc per tile) and the feed-forward block that closes the layer (also in pgdvs_gnt_ray_layer) run on the bf16
 *   matrix instructions with both operands split EXACTLY into three bf16 pieces (the six partial products above 2^-24 of the
 *   product, fp32 accumulation: the accuracy of an fp32 multiply-add chain, in about half its time); the weight blob carries
 *   the feed-forward weights a second time as pre-split images (pgdvs_amd.ops.ff_bf16x3_images).  The option gnt_fp32
 *   (PGDVS_GNT_FP32=1 at load time, pgdvs_option_set) keeps every product on the fp32 matrix instructions.
 *   Inputs below 2^-110 in magnitude or non-finite are outside the split path's contract: the truncation residues of a
 *   split underflow (the product loses its low pieces) and inf / NaN turn into NaN (inf - inf in the residue), where the
 *   fp32 instruction would carry them through; image features and LayerNorm outputs never get there. */
int64_t pgdvs_gnt_view_weight_floats(void);
int pgdvs_gnt_view_layer(const float *weights, const float *q_in, const float *feat,
                         const float *ray_diff, const uint8_t *valid, int64_t N, int V, float *q_out,
                         float *stats, pgdvs_stream_t stream);

/* A14 (ray transformer): Transformer + Attention(attn_mode="qk", 4 heads) of
 * pgdvs/models/gnt/models/transformer_network.py:231-338 for R rays x S samples (S <= 256),
 * width 64, fused with its feed-forward block.  weights: same packed layout as the view
 * layer (LN1 = attn_norm, WQ/WK/WV, WO = out_fc, LN2/F1/F2 = ff_norm/ff; the view-only
 * regions are unused).  q_out[R,S,64]; sample_weights[R,S] (nullable) = attention row of
 * query sample 0 averaged over heads (:336). */
int pgdvs_gnt_ray_layer(const float *weights, const float *q_in, int R, int S, float *q_out,
                        float *sample_weights, pgdvs_stream_t stream);

/* A10, dyn_render_type = "mesh" (pgdvs_renderer_dyn.py:542-669): triangulates the kept source
 * pixels (two triangles per pixel quad whose corners are all kept; faces touching the first
 * kept pixel are dropped, as upstream :597 does) and renders them into the target camera
 * with pytorch3d MeshRasterizer semantics (blur_radius 0, 1 face per pixel, perspective-
 * correct barycentrics) + vertex colours + hard blend on black.
 *   keep[H*W] u8 (valid_dyn_mask_1), pcl[H*W,3], rgb[H*W,3]: dense over the source frame
 *   img_planar[3,H,W], mask[H,W] (1 where a face covers the pixel centre),
 *   face_out[H*W] int32 (nullable): kind*H*W + source pixel of the winning face, -1 = none. */
int64_t pgdvs_mesh_render_workspace_bytes(int H, int W);
int pgdvs_mesh_render(const float *cam_tgt, int H, int W, const uint8_t *keep, const float *pcl,
                      const float *rgb, float *img_planar, float *mask, int32_t *face_out,
                      void *workspace, int64_t workspace_bytes, pgdvs_stream_t stream);

/* A17: tracker-window point aggregation (pgdvs_renderer_dyn_track.py:98-396); tracks and
 * visibilities are inputs.  Frames are ordered as prepare_data orders them (:599-716):
 * [fwd2tgt tracks..., temporally-closest..., bwd2tgt tracks...].
 *   tracks[P,N,2] (col,row), visibles[P,N] u8, frame_kind_host[N] (HOST array; 1 = temporally
 *   closest frame, 2 = real track frame), times[N] / time_tgt[1] raw time stamps (device;
 *   shifted to start at 0 internally as :718-721), rgbs[N,H,W,3], depths[N,H,W],
 *   cams[N,PGDVS_CAM_BLOCK].
 *   valid[P] u8 : invisible in every closest frame and visible in >= 2 track frames (:115-127)
 *   pcl[P,3]    : the two visible frames nearest in time (:146-166) unprojected with the
 *                 nearest-sampled depth (:220-253) and inter/extrapolated to time_tgt (:278-284)
 *   rgb[P,3]    : mean of the two bilinear (align_corners=True) colour samples (:197-218,:271-276)
 * Rows of invalid tracks are zero.  N <= 64. */
int pgdvs_track_points(const float *tracks, const uint8_t *visibles, int64_t P, int N,
                       const uint8_t *frame_kind_host, const float *times, const float *time_tgt,
                       const float *rgbs, const float *depths, int H, int W, const float *cams,
                       uint8_t *valid, float *pcl, float *rgb, pgdvs_stream_t stream);

/* pytorch3d.ops.knn_points(queries, pts, K=KK) + mean over ALL KK squared distances (the
 * track-to-base filter, :299-312; no self column).  Exact uniform-grid search over `pts`;
 * counts on the device; KK <= 64; missing columns (fewer than KK points) count as 0.
 * avg_out[query_capacity].  workspace >= pgdvs_knn_cross_workspace_bytes(capacity, query_capacity). */
int64_t pgdvs_knn_cross_workspace_bytes(int64_t capacity, int64_t query_capacity);
int pgdvs_knn_cross_mean_dist(const float *queries, const int32_t *query_count, int64_t query_capacity,
                              const float *pts, const int32_t *count, int64_t capacity, int KK,
                              float *avg_out, void *workspace, int64_t workspace_bytes,
                              pgdvs_stream_t stream);

/* flag[i] = avg[i] < (*thres * mult) for i < *count (:314-318, :363-371).  gate_count
 * (nullable, device): when *gate_count == 0 ("no base cloud", :296-298) the test becomes
 * avg[i] < *alt_thres, or passes everything if alt_thres is null.  flag_out[i] = 0 for
 * *count <= i < capacity. */
int pgdvs_threshold_flags(const float *avg, const int32_t *count, int64_t capacity, const float *thres,
                          float mult, const float *alt_thres, const int32_t *gate_count,
                          uint8_t *flag_out, pgdvs_stream_t stream);

/* out = concat(a[0:*count_a], b[0:*count_b]) by rows of `width` floats, *count_out = rows
 * written (torch.cat of :390-394).  require_a != 0: the result is empty when *count_a == 0.
 * b may be null.  out holds capacity_a + capacity_b rows. */
int pgdvs_concat_rows(const float *a, const int32_t *count_a, int64_t capacity_a, const float *b,
                      const int32_t *count_b, int64_t capacity_b, int width, int require_a,
                      float *out, int32_t *count_out, pgdvs_stream_t stream);

/* A11 alone: combined = (1-m)*static + m*dyn (pgdvs_renderer.py:169-178), n elements per
 * channel, planar [3,n] with mask [n]. */
int pgdvs_combine(const float *static_rgb, const float *dyn_rgb, const float *dyn_mask,
                  int64_t n, float *combined, float *combined_static, float *combined_dyn,
                  pgdvs_stream_t stream);

/* ---- 8f-1, the caller's metric: PGDVSEvaluator.eval_step's image statistics for one view in one pass
 * (pgdvs/engines/evaluator_pgdvs.py:52-77: clamp -> NaN to 0 -> (x*255).byte().float()/255 of prediction and ground
 * truth; :190-283 with pgdvs/utils/training.py:281-313: sum((gt - pred)^2 * mask) and sum(mask) in float64 for the
 * masks ones / eval_mask / 1 - eval_mask).
 *   pred_planar[3,H,W] raw render (combined_rgb); gt_hwc[H,W,3] raw ground truth; mask_hwc[H,W,3] eval_mask
 *   pred_q / gt_q [3,H,W] (nullable): the quantised images
 *   sums: DEVICE double[8] = sum d2, sum d2*m, sum d2*(1-m), 3*H*W, sum m, sum (1-m), then two words that ride along so that
 *   the evaluator's step reads ONE block back: (double)*count_dev (the static cloud's device count; -1 when NULL) and
 *   (double)*status_dev (pgdvs_points_raster_bounded's status word; 0 when NULL)
 *   PSNR_k = 10 log10(1 / (sums[k] / (sums[3+k] + 1e-8))), 0 when the sum of squares is 0 (upstream's quirk). */
int64_t pgdvs_eval_psnr_workspace_bytes(void);
int pgdvs_eval_psnr_sums(const float *pred_planar, const float *gt_hwc, const float *mask_hwc, int H, int W,
                         float *pred_q, float *gt_q, const int64_t *count_dev, const int32_t *status_dev, double *sums,
                         void *workspace, int64_t workspace_bytes, pgdvs_stream_t stream);

/* ---- one native call per target view -------------------------------------------------
 * PGDVSRenderer.forward with static_renderer = StaticGeoPointRenderer, dyn_render_type = "softsplat",
 * batch item of size 1, render_stride 1, no tracker (pgdvs/renderers/pgdvs_renderer.py:84-178 ->
 * st_geo_renderer.py:77-120 + pgdvs_renderer_dyn.py:63-257,275-540), as the evaluator drives it once per
 * target view (pgdvs/engines/evaluator_pgdvs.py:36-54) -- and, when agg_S > 0, the static cloud aggregated
 * first from the resident video (A12, nvidia_eval_pure_geo.py:183-277).  Enqueues the whole chain (A12, A9,
 * A2-A5 with the kNN outlier filter, A6-A8, A11) from C++: one ctypes call instead of ~85, same kernels, same
 * results as the per-op entry points above.
 * Every pointer is a DEVICE pointer except agg_K3s_host / agg_c2ws_host. */
typedef struct pgdvs_view_geo_desc {
  int32_t H, W;                 /* source and target resolution (render_stride 1)                      */
  /* cameras and times: rows of the data dict (A0)                                                    */
  const float *flat_cam_tgt;    /* [34]                                                                */
  const float *flat_cam_src;    /* [2,34]  the two temporally closest source frames                    */
  const float *time_src;        /* [2]     time_src_temporal                                           */
  const float *time_tgt;        /* [1]                                                                 */
  /* dynamic branch inputs                                                                            */
  const float *rgb1, *rgb2;     /* [H,W,3] each (rgb_src_temporal[0], [1])                             */
  const float *depth1, *depth2; /* [H,W]                                                               */
  const float *dyn_mask1;       /* [H,W]   0/1 floats (dyn_mask_src_temporal[0])                       */
  const float *flow12;          /* [H,W,2] flow_fwd                                                    */
  const float *flow_occ;        /* [H,W]   flow_fwd_occ_mask (needed when use_flow_consistency)        */
  int32_t use_flow_consistency; /* render_cfg.dyn_render_use_flow_consistency                          */
  int32_t remove_outlier;       /* render_cfg.dyn_pcl_remove_outlier                                   */
  int32_t outlier_knn;          /* render_cfg.dyn_pcl_outlier_knn                                      */
  float outlier_std_thres;      /* render_cfg.dyn_pcl_outlier_std_thres                                */
  float alpha;                  /* softsplat_metric_abs_alpha                                          */
  const float *noise;           /* [3,H,W] injected normal field, or NULL                              */
  uint64_t *rng_state;          /* {seed, draw number} (pgdvs_dyn_splat_composite_rng), or NULL; both NULL: zeros */
  /* static cloud, given ...                                                                          */
  const float *st_pcl_rgb;      /* [st_rows,6] (xyz,rgb); ignored when agg_S > 0                       */
  const float *st_pcl_xyz;      /* [st_rows,3] packed coordinates or NULL                              */
  int64_t st_rows;              /* rows of the two buffers                                             */
  const int64_t *st_count_dev;  /* device count (rows actually present) or NULL = st_rows              */
  /* ... or aggregated inside the call (A12; arguments of pgdvs_static_aggregate_packed)              */
  int32_t agg_S;                /* 0: use st_pcl_rgb                                                   */
  const float *agg_rgbs;        /* [S,H,W,3]                                                           */
  const float *agg_depths;      /* [S,H,W]                                                             */
  const uint8_t *agg_masks;     /* [S,H,W]                                                             */
  const double *agg_K3s_host;   /* HOST [S,9]                                                          */
  const double *agg_c2ws_host;  /* HOST [S,16]                                                         */
  float *agg_cloud_out;         /* [agg_capacity,6]                                                    */
  float *agg_xyz_out;           /* [agg_capacity,3]                                                    */
  int64_t agg_capacity;
  int64_t *agg_count_out;       /* device int64 (or -1, see pgdvs_static_aggregate)                    */
  /* rasteriser                                                                                       */
  int64_t row_bound;            /* rows the tile lists are sized for (<= the buffers' rows); <= 0: all rows */
  float radius;                 /* render_cfg.st_render_pcl_pt_radius                                  */
  int32_t K;                    /* render_cfg.st_render_pcl_pts_per_pixel                              */
  /* outputs, planar                                                                                  */
  float *static_rgb;            /* [3,H,W] geo_static_rgb                                              */
  float *static_mask;           /* [H,W]   geo_static_mask                                             */
  int32_t *raster_status;       /* device int32 (pgdvs_points_raster_bounded)                          */
  float *render_dyn_rgb;        /* [3,H,W]                                                             */
  float *render_dyn_mask;       /* [H,W]                                                               */
  float *combined;              /* [3,H,W] combined_rgb                                                */
  float *combined_static;       /* [3,H,W]                                                             */
  float *combined_dyn;          /* [3,H,W]                                                             */
  /* optional: the dynamic branch's geometry (A2-A5) on a second stream, joined before the splat       */
  pgdvs_stream_t side_stream;   /* NULL: everything on `stream`                                        */
  /* agg_S > 0 only: non-zero = this workspace ran the previous pgdvs_view_geo_forward with the SAME agg_S, H, W,
   * agg_capacity, agg_K3s_host and agg_c2ws_host and nothing else has written to it since: the per-frame camera constants
   * are still in it and are not uploaded again (five launches less per view).  0 is always correct. */
  int32_t agg_params_cached;
} pgdvs_view_geo_desc;

/* sizeof(pgdvs_view_geo_desc) as this library was compiled: a binding checks its own struct against it */
int64_t pgdvs_view_geo_desc_size(void);
/* bytes of workspace pgdvs_view_geo_forward needs for this description (negative status on a bad one) */
int64_t pgdvs_view_geo_workspace_bytes(const pgdvs_view_geo_desc *desc);
int pgdvs_view_geo_forward(const pgdvs_view_geo_desc *desc, void *workspace, int64_t workspace_bytes,
                           pgdvs_stream_t stream);
/* What the fast paths of the LAST pgdvs_view_geo_forward on this (description, workspace) left to their slower exits, read
 * from device words the kernels write anyway: counters_dev = DEVICE int64[PGDVS_VIEW_COUNTERS], written on `stream` (enqueue
 * it behind the forward call, before the workspace is used again). */
#define PGDVS_VIEW_CNT_STATIC_ROWS 0         /* rows of the static cloud (device count)                                   */
#define PGDVS_VIEW_CNT_RASTER_ENTRIES 1      /* tile-list entries of the rasteriser                                       */
#define PGDVS_VIEW_CNT_RASTER_LONGEST 2      /* longest tile list                                                         */
#define PGDVS_VIEW_CNT_RASTER_TILES_LONG 3   /* tiles whose list exceeds the LDS-sorted path's 2048 entries (general path) */
#define PGDVS_VIEW_CNT_RASTER_TILES_TIES 4   /* tiles the sorted path handed to the general path for > 64 equal depths     */
#define PGDVS_VIEW_CNT_KNN_QUERIES 5         /* points of the dynamic cloud the outlier filter searched                   */
#define PGDVS_VIEW_CNT_KNN_TO_RING 6         /* queries the thread-per-query pass left to the ring search                 */
#define PGDVS_VIEW_CNT_KNN_TO_COARSE 7       /* queries the ring search left to the coarse grid                           */
#define PGDVS_VIEW_CNT_KNN_TO_EXHAUSTIVE 8   /* queries scanned exhaustively                                              */
#define PGDVS_VIEW_CNT_AGG_FP64_POINTS 9     /* aggregation: points with projections the fp32 screening left to the fp64 queue */
#define PGDVS_VIEW_CNT_AGG_REFERENCE_ORDER 10 /* aggregation: projections evaluated in the reference's operation order      */
#define PGDVS_VIEW_COUNTERS 12
int pgdvs_view_geo_counters(const pgdvs_view_geo_desc *desc, const void *workspace, int64_t workspace_bytes,
                            int64_t *counters_dev, pgdvs_stream_t stream);
/* host enqueue statistics of pgdvs_view_geo_forward since the last call of this function: calls made and the
 * wall-clock seconds spent inside them (bench.py's host_enqueue figure); resets both. */
void pgdvs_view_geo_host_stats(int64_t *calls, double *seconds);

#ifdef __cplusplus
}
#endif
#endif /* PGDVS_HIP_H_ */
