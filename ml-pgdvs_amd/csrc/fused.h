// Internal interface of the per-view call's FUSED launches (round 6).  One view alone is a dependent chain of launches that cost
// ~4.7 us each whatever they do (tools/r06_latency_trace.sh), and the host pays ~3.8 us per launch: the native call
// (view_geo.cpp) therefore hands small jobs -- clears, counts, a bounding box, a median's bin selection, a projection -- to the
// neighbouring launch that already walks the same data.  The per-op entry points of include/pgdvs_hip.h keep their own
// launches and produce the same bytes; the tests run both.
#pragma once
#include "common.h"

namespace pgdvs {

// dyn.hip: what dyn_warp_kernel does on the side (see the kernel)
struct WarpExtras {
  uint8_t *zero_a, *zero_b;  // [P] byte maps cleared pixel by pixel, or null
  uint4 *zero0, *zero1, *zero2, *zero3;  // 16-byte aligned blocks of n16_0 .. n16_3 granules cleared by the launch's threads, or null
  int n16_0, n16_1, n16_2, n16_3;
  int32_t *chunk_cnt;        // [ceil(P / 256)] valid pixels per 256-pixel chunk (= per workgroup), or null
};
int dyn_warp_fused(int H, int W, const float *dyn_mask1, const float *occ, int use_flow_consistency, const float *flow12,
                   const float *depth1, const float *depth2, const float *rgb1, const float *rgb2, const float *cam1,
                   const float *cam2, const float *times, uint8_t *mask_eff, uint8_t *valid, float *pcl, float *rgbf,
                   const WarpExtras &ex, hipStream_t st);

// scan.hip: ordered compaction of flags[n] whose per-chunk counts exist already (chunk_cnt[ceil(n / 256)]): indices, count,
// the gathered rows rows_out[i] = rows_in[idx[i]] (3 floats) and their bounding box (knn_grid's encoding; bbox zeroed by
// the caller) in ONE launch
// (zero_per_row: zero_mult ints cleared per compacted row, + one behind the last row -- the kNN grid's quarter-cell counters --, or null)
int compact_gather_bbox(const uint8_t *flags, int64_t n, const int32_t *chunk_cnt, int32_t *idx_out, int32_t *count_out,
                        const float *rows_in, float *rows_out, unsigned *bbox, int32_t *zero_per_row, int zero_mult, hipStream_t st);

// knn_grid.hip: the state block the search's memset clears (bounding box first), and the search with that block already
// cleared and the bounding box already there
// (tab / tab_bytes: the sparse cell index's bit table, cleared in full by the caller; occ_count: the quarter-cell counters, of
// which the caller clears occ_mult per point + 1 -- what grid_tab_zero_kernel clears for the per-op entry point)
// (coarse / coarse_bytes: the second-level grid's cell counters, cleared in full by the caller)
void knn_grid_state_block(void *workspace, int64_t capacity, void **block, int64_t *bytes, unsigned **bbox, void **tab,
                          int64_t *tab_bytes, int32_t **occ_count, int *occ_mult, void **coarse, int64_t *coarse_bytes);
int knn_grid_mean_dist_prepared(const float *pts, const int32_t *count, int64_t capacity, int K, float *avg_out, void *workspace,
                                int64_t workspace_bytes, hipStream_t st);

// knn.hip: the statistical filter with every bin selection inside the pass that follows it, the flags scattered straight
// into keep[idx[i]] (keep cleared by the caller); the histograms' block (outlier_hist_block) cleared by the caller
void outlier_hist_block(void *workspace, void **block, int64_t *bytes);
int outlier_keep_fused(const float *avg, const int32_t *count, int64_t capacity, float std_thres, float *thres_out,
                       const int32_t *idx, uint8_t *keep, void *workspace, int64_t workspace_bytes, hipStream_t st);

// softsplat.hip: the first half of the splat composite with the projection of the kept points into the target view
// (pgdvs_project_flow_dense) inside the flag pass; flags cleared by the caller
int dyn_splat_scatter_part_fused(int H, int W, const float *rgb1, const float *rgb2, const float *flow12, const float *cam_tgt,
                                 const float *pcl, const uint8_t *keep, float *flow_1_to_tgt, float *valid_mask,
                                 const float *noise, const unsigned long long *rng, float alpha, void *workspace,
                                 bool flags_cleared, hipStream_t st);
uint8_t *dyn_splat_flag_map(void *workspace, int H, int W);


// raster.hip: the counters a rasterisation starts from (cleared per call) and the bounded rasterisation with them cleared by
// the caller (static_agg.hip's agg_rows launch clears them for the per-view call)
void raster_counter_block(void *workspace, int64_t n_rows, int H, int W, float radius, void **block, int64_t *bytes);
int points_raster_bounded_cleared(const float *pts, int64_t pts_stride, const float *feat, int64_t feat_stride, int64_t n_points,
                                  const int64_t *n_points_dev, int64_t row_bound, int32_t *status_dev, const float *cam_tgt,
                                  float radius, int K, int H, int W, float *rgb, int rgb_planar, float *mask, void *workspace,
                                  int64_t workspace_bytes, pgdvs_stream_t stream, bool counters_cleared);

}  // namespace pgdvs
