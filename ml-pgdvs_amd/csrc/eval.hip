// SURVEY 8f-1, the caller's side of the path: the evaluator's per-view image metric
// (pgdvs/engines/evaluator_pgdvs.py:52-77 quantisation, :190-283 the three masked PSNRs through
// pgdvs/utils/training.py:281-313 calculate_psnr) as ONE pass over the rendered image: clamp -> NaN to 0 -> 8-bit
// quantisation of prediction and ground truth, squared differences and mask sums in float64.  Upstream this is ~45
// elementwise torch / numpy passes and a dozen host synchronisations per view -- several times the cost of rendering
// the view on this GPU; here the step enqueues two launches and reads eight doubles back (the six sums and the
// geometry path's two status words).
#include "common.h"

namespace pgdvs {

constexpr int kEvalBlocks = 256;
constexpr int kEvalThreads = 256;
constexpr int kEvalSums = 6;  // sum d2, sum d2 m, sum d2 (1 - m), count, sum m, sum (1 - m)

// (x.clamp(0, 1) -> nan_to_num(nan=0) -> (x * 255).byte().float() / 255.0), fp32 like torch
__device__ __forceinline__ float quantise_u8(float x) {
  x = x != x ? 0.0f : (x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x));
  const float q = (float)(uint8_t)(x * 255.0f);
  return q / 255.0f;
}

__global__ void __launch_bounds__(kEvalThreads)
eval_partials_kernel(const float *__restrict__ pred, const float *__restrict__ gt, const float *__restrict__ mask, int P,
                     float *__restrict__ pred_q, float *__restrict__ gt_q, double *__restrict__ partials) {
  double acc[kEvalSums] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int p = blockIdx.x * kEvalThreads + threadIdx.x; p < P; p += kEvalBlocks * kEvalThreads) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float a = quantise_u8(gt[(size_t)p * 3 + c]);     // img1 = ground truth [H,W,3]
      const float b = quantise_u8(pred[(size_t)c * P + p]);   // img2 = prediction, planar [3,H,W]
      const float m = mask[(size_t)p * 3 + c];
      const float ms = 1.0f - m;  // (the static mask is formed in fp32 upstream, :243-246)
      if (pred_q) pred_q[(size_t)c * P + p] = b;
      if (gt_q) gt_q[(size_t)c * P + p] = a;
      const double d = (double)a - (double)b;
      const double d2 = d * d;
      acc[0] += d2;
      acc[1] += d2 * (double)m;
      acc[2] += d2 * (double)ms;
      acc[3] += 1.0;
      acc[4] += (double)m;
      acc[5] += (double)ms;
    }
  }
  __shared__ double s[kEvalThreads / 64][kEvalSums];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kEvalSums; ++k) {
    double v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) s[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kEvalSums) {
    double v = 0.0;
    for (int w = 0; w < kEvalThreads / 64; ++w) v += s[w][threadIdx.x];
    partials[(size_t)blockIdx.x * kEvalSums + threadIdx.x] = v;
  }
}

// fixed-order final sum: the result does not depend on scheduling
// (+ the renderer's two device-side status words, so that the step reads everything back in ONE transfer: sums[6] = the
// static cloud's row count or -1 without one, sums[7] = the rasteriser's status word or 0)
__global__ void eval_final_kernel(const double *__restrict__ partials, double *__restrict__ sums, const int64_t *__restrict__ count_dev,
                                  const int32_t *__restrict__ status_dev) {
  if (threadIdx.x < kEvalSums) {
    double v = 0.0;
    for (int b = 0; b < kEvalBlocks; ++b) v += partials[(size_t)b * kEvalSums + threadIdx.x];
    sums[threadIdx.x] = v;
  }
  if (threadIdx.x == kEvalSums) sums[kEvalSums] = count_dev ? (double)*count_dev : -1.0;
  if (threadIdx.x == kEvalSums + 1) sums[kEvalSums + 1] = status_dev ? (double)*status_dev : 0.0;
}

}  // namespace pgdvs

using namespace pgdvs;

PGDVS_API int64_t pgdvs_eval_psnr_workspace_bytes(void) { return (int64_t)kEvalBlocks * kEvalSums * 8; }

PGDVS_API int pgdvs_eval_psnr_sums(const float *pred_planar, const float *gt_hwc, const float *mask_hwc, int H, int W,
                                   float *pred_q, float *gt_q, const int64_t *count_dev, const int32_t *status_dev, double *sums,
                                   void *workspace, int64_t workspace_bytes, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(pred_planar && gt_hwc && mask_hwc && sums && H > 0 && W > 0 && (int64_t)H * W < (1ll << 30),
                "pgdvs_eval_psnr_sums: bad arguments");
  if (!workspace || workspace_bytes < pgdvs_eval_psnr_workspace_bytes()) {
    set_error("pgdvs_eval_psnr_sums: workspace too small");
    return PGDVS_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  double *partials = reinterpret_cast<double *>(workspace);
  PGDVS_LAUNCH("eval_partials", eval_partials_kernel, dim3(kEvalBlocks), dim3(kEvalThreads), 0, st, pred_planar, gt_hwc, mask_hwc,
               H * W, pred_q, gt_q, partials);
  PGDVS_LAUNCH("eval_final", eval_final_kernel, dim3(1), dim3(64), 0, st, (const double *)partials, sums, count_dev, status_dev);
  return check_launch("eval_psnr_sums");
}
