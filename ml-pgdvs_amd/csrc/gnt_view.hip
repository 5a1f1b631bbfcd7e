// A14: the per-layer kernels of GNT.forward (pgdvs/models/gnt/models/transformer_network.py):
//   gnt_view_layer   Transformer2D + Attention2D (:59-169,197-223), attention part
//   gnt_ray_attn     Transformer + Attention, attn_mode="qk" (:231-338), attention part
//   gnt_ff           the feed-forward block that follows both (:44-55,:218-221)
// all on the fp32 matrix cores (exact fp32 products, fp32 accumulate: TF32 is off in the
// reference, pgdvs/run.py:21-24, and outputs must agree to 1e-4).
//
// View layer, per (ray,sample) group:
//   x = LN(q); q' = Wq x
//   for every source view v:  k = Wk f_v ; vv = Wv k ; pos = P2 relu(P1 d_v + b) + b ;
//                             a = A2 relu(A1 (k - q' + pos) + b) + b          (64 -> 8 -> 64)
//   attn = softmax_v(a) (masked);  x = Wo sum_v (vv + pos) * attn + bo + q
//
// Everything is computed TRANSPOSED: features run along the MFMA M dimension, groups along N;
// weights are the A operand (input-major rows), activations the B operand, one group per lane
// column; an accumulator's register layout is the next product's B-operand layout, so chained
// layers never leave registers.  Tile shapes and helpers: gnt_mfma.h (16-row tiles) for the two
// attention kernels; the feed-forward block keeps 32-row tiles on v_mfma_f32_32x32x2_f32 (it
// carries little per-lane state, and a 32-row tile reads each LDS weight half as often).
// What bounds these kernels (vector-ALU work never overlaps the fp32 MFMA, global loads must
// not sit in front of accumulator chains): DESIGN.md section 4.
#include "common.h"
#include "gnt_mfma.h"

namespace pgdvs {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// ---- packed per-layer weights (floats), input-major ("t" = transposed) -------------------
constexpr int VW_LN1_G = 0;                       // [64]
constexpr int VW_LN1_B = VW_LN1_G + 64;           // [64]
constexpr int VW_WQ = VW_LN1_B + 64;              // [64 in][64 out]
constexpr int VW_WK = VW_WQ + 4096;
constexpr int VW_WV = VW_WK + 4096;
constexpr int VW_P1 = VW_WV + 4096;               // [4 in][32 out, 8 used]
constexpr int VW_P1B = VW_P1 + 128;               // [32]
constexpr int VW_P2 = VW_P1B + 32;                // [8 in][64 out]
constexpr int VW_P2B = VW_P2 + 512;               // [64]
constexpr int VW_A1 = VW_P2B + 64;                // [64 in][32 out, 8 used]
constexpr int VW_A1B = VW_A1 + 2048;              // [32]
constexpr int VW_A2 = VW_A1B + 32;                // [8 in][64 out]
constexpr int VW_A2B = VW_A2 + 512;               // [64]
constexpr int VW_WO = VW_A2B + 64;                // [64][64]
constexpr int VW_WOB = VW_WO + 4096;              // [64]
constexpr int VW_LN2_G = VW_WOB + 64;
constexpr int VW_LN2_B = VW_LN2_G + 64;
constexpr int VW_F1 = VW_LN2_B + 64;              // [64 in][256 out]
constexpr int VW_F1B = VW_F1 + 16384;             // [256]
constexpr int VW_F2 = VW_F1B + 256;               // [256 in][64 out]
constexpr int VW_F2B = VW_F2 + 16384;             // [64]
// the feed-forward weights once more, as the bf16x3 images of gnt_ff_bf16x3_kernel (ops.ff_bf16x3_images): two halves of 24576 floats
constexpr int VW_FFIMG = VW_F2B + 64;
constexpr int VW_TOTAL = VW_FFIMG + 2 * 24576;

// 32-row tiles (feed-forward block): lane (i, h) keeps, for its row i, the 32 features
// feature(t,h) = featc(t) + 4*h.  The lane-dependent 4*h always goes into a per-lane BASE
// pointer and featc(t) stays a compile-time constant, so every access is base + immediate
// (written as one sum, the compiler merges 4*h with OR and materialises one address register
// per element).
__host__ __device__ constexpr int featc(int t) { return (t & 3) + 8 * ((t & 15) >> 2) + 32 * (t >> 4); }

__device__ __forceinline__ floatx16 mfma(float a, float b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// LayerNorm over the 64 features of a group (32 here, 32 in the partner lane l^32)
__device__ __forceinline__ void layer_norm64(const float (&x)[32], const float *__restrict__ g,
                                             const float *__restrict__ b, float eps, float (&y)[32], int h) {
  float s = 0.0f;
#pragma unroll
  for (int t = 0; t < 32; ++t) s += x[t];
  s += __shfl_xor(s, 32, 64);
  float mean = s * (1.0f / 64.0f);
  float v = 0.0f;
#pragma unroll
  for (int t = 0; t < 32; ++t) {
    float d = x[t] - mean;
    v += d * d;
  }
  v += __shfl_xor(v, 32, 64);
  float rstd = 1.0f / sqrtf(v * (1.0f / 64.0f) + eps);
  const float *gb = g + 4 * h, *bb = b + 4 * h;
#pragma unroll
  for (int t = 0; t < 32; ++t) y[t] = (x[t] - mean) * rstd * gb[featc(t)] + bb[featc(t)];
}

__device__ __forceinline__ void store_row32(float *__restrict__ row, const float (&x)[32], int h) {
  float *rb = row + 4 * h;
#pragma unroll
  for (int c = 0; c < 8; ++c)
    *reinterpret_cast<float4 *>(rb + featc(c * 4)) = make_float4(x[c * 4], x[c * 4 + 1], x[c * 4 + 2], x[c * 4 + 3]);
}

__device__ __forceinline__ void load_row32(const float *__restrict__ row, float (&x)[32], int h) {
  // features feature(t,h): groups of 4 contiguous floats -> 8 x dwordx4
  const float *rb = row + 4 * h;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float4 v = *reinterpret_cast<const float4 *>(rb + featc(c * 4));
    x[c * 4 + 0] = v.x;
    x[c * 4 + 1] = v.y;
    x[c * 4 + 2] = v.z;
    x[c * 4 + 3] = v.w;
  }
}

// ---------------------------------------------------------------------------------------
// View transformer kernel.  Tiles of 16 (ray,sample) groups per wavefront on the 16x16x4 MFMA:
// lane = j + 16*hq (group j of the tile, quarter hq), and a lane keeps, for its group, the 16
// features F(t,hq) = 16*(t>>2) + 4*hq + (t&3), t = 0..15 -- the accumulator layout of
// v_mfma_f32_16x16x4_f32 (row = 4*hq + r of output tile mt <-> t = 4*mt + r), which again is
// the B-operand layout of the next product (K-step (c,r) <-> t = 4*c + r, k index = hq).
// Half the per-lane state of a 32-group tile: every running quantity of the softmax and of the
// side statistics stays in the 256 architectural VGPRs (the vector ALU cannot read AGPRs), and
// two wavefronts fit per SIMD, so one wavefront's exp / accumulate work runs in the shadow of
// the other's matrix instructions.  The 64 -> 8 -> 64 MLPs also waste less padding (M = 16).
// ---------------------------------------------------------------------------------------
// offsets inside the unpadded tail of the LDS image (same order as the packed weights)
constexpr int kSmP1B = VW_P1B - VW_P1, kSmP2 = VW_P2 - VW_P1, kSmP2B = VW_P2B - VW_P1;
constexpr int kSmA1B = VW_A1 - VW_P1;  // A1 itself lives in the padded region
constexpr int kSmA2 = kSmA1B + (VW_A2 - VW_A1B), kSmA2B = kSmA1B + (VW_A2B - VW_A1B);
constexpr int kSmallFloats = kSmA1B + (VW_WO - VW_A1B);
// four lane-major 64 x 64 images (Wk and Wv as bf16x3 images of 6144 floats when the k / v products run on the bf16 pipe), A1's,
// the small pieces
constexpr int kViewLdsFloats = 4 * 6144 + 1024 + kSmallFloats + 192 + 1024 + 8192;

constexpr float kLog2e = 1.4426950408889634f;
// The softmax over views keeps a per-feature reference logit m and rescales the running sums
// only when a new logit exceeds it by more than this gap (always on the first valid view,
// where m = -inf): exp(a - m) <= e^16 cannot overflow, and softmax is shift-invariant, so a
// stale reference changes nothing but the rounding.  The common case is then one v_exp_f32
// and two fused multiply-adds per (feature, view) instead of two library expf calls.
constexpr float kRescaleGap = 16.0f;

__device__ __forceinline__ void layer_norm64q(const float (&x)[16], const float *__restrict__ g,
                                              const float *__restrict__ b, float eps, float (&y)[16], int hq) {
  float s = 0.0f;
#pragma unroll
  for (int t = 0; t < 16; ++t) s += x[t];
  const float mean = quad_sum(s) * (1.0f / 64.0f);
  float v = 0.0f;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float d = x[t] - mean;
    v += d * d;
  }
  const float rstd = 1.0f / sqrtf(quad_sum(v) * (1.0f / 64.0f) + eps);
  float gg[16], bb[16];
  load_row16(g, gg, hq);
  load_row16(b, bb, hq);
#pragma unroll
  for (int t = 0; t < 16; ++t) y[t] = (x[t] - mean) * rstd * gg[t] + bb[t];
}

// 64 -> 16 (8 used): one output tile (lane-major image [s>>2][lane][s&3], see gnt_mfma.h), started from the bias
// `c0` (accumulator operand of the first MFMA).  One accumulator: a dependent chain of these MFMAs issues back to
// back at the pipe's own pace (profiles/r01_mfma_calibration.txt), two would need an addition at the end.
template <class Next>
__device__ __forceinline__ floatx4 chain64n(const float *__restrict__ lb, const float (&x)[16], float (&w)[8],
                                            floatx4 c0, Next &&next) {
  float w2[8];
  ldq8v(w2, lb, 1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < 8; ++u) c0 = mfma16(w[u], x[u], c0);
  __builtin_amdgcn_sched_barrier(0);
  next(w);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < 8; ++u) c0 = mfma16(w2[u], x[8 + u], c0);
  __builtin_amdgcn_sched_barrier(0);
  return c0;
}

// The 8 hidden units of a 64 -> 8 -> 64 MLP leave the first layer in lanes hq = 0,1 (unit
// 4*hq + r in register r; hq = 2,3 hold the M padding).  Moving the lower half-wave's odd
// registers into the upper half-wave packs them into two full K = 4 steps:
// step u, lane (j,hq) <-> hidden unit 4*(hq&1) + (hq>>1) + 2*u.
__device__ __forceinline__ void pack_hidden(const float (&hid)[4], float (&hk)[2]) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(hid[2 * u]), __float_as_int(hid[2 * u + 1]), false, false);
    hk[u] = __int_as_float(r[0]);
  }
}

// SPLIT: the 64 x 64 products (q' = Wq LN(q) and the out_fc product per tile, k = Wk f and vv = Wv k per view) as bf16x3
// products on v_mfma_f32_16x16x32_bf16 (gnt_mfma.h: fp32-faithful, 768 + ~400
// cycles per product instead of 2048); false: every product on the fp32 instruction (PGDVS_GNT_FP32=1).
template <bool STATS, bool SPLIT>
__global__ void __launch_bounds__(512, 1)
gnt_view_layer_kernel(const float *__restrict__ W_arg, const float *__restrict__ q_in,
                      const float *__restrict__ feat, const float *__restrict__ ray_diff,
                      const uint8_t *__restrict__ valid, int64_t N, int V, float *__restrict__ q_out,
                      float *__restrict__ stats) {
  // LDS image of the layer's weights: the 64 x 64 matrices and A1 lane-major (gnt_mfma.h: one ds_read_b128 per
  // K-step off one per-lane base), the small pieces as packed
  extern __shared__ __attribute__((aligned(16))) float s_w[];  // [kViewLdsFloats]
  float *s_wk = s_w, *s_wv = s_wk + 6144, *s_wq = s_wv + 6144, *s_wo = s_wq + 6144, *s_a1 = s_wo + 6144;
  float *s_small = s_a1 + 1024, *s_par = s_small + kSmallFloats;
  // per lane the second layers' A operands of the two small MLPs ([4 quads][lane][4]), per wavefront the tile's c1 ([4][lane][4]):
  // loop invariants that would otherwise hold 32 of the 256 registers through the view loop
  float *s_pw = s_par + 192, *s_c1 = s_pw + 1024;
  if (SPLIT) {
    stage_w64_bf16x3<512>(W_arg + VW_WK, s_wk);
    stage_w64_bf16x3<512>(W_arg + VW_WV, s_wv);
  } else {
    stage_w64_lanes<512>(W_arg + VW_WK, s_wk);
    stage_w64_lanes<512>(W_arg + VW_WV, s_wv);
  }
  {  // A1 [64 in][32 out, 8 used] -> [s>>2][lane][s&3] for the 16 output columns of the (padded) tile
    const int q = (int)threadIdx.x;  // 512 float4's
    const float4 v = reinterpret_cast<const float4 *>(W_arg + VW_A1)[q];
    const int in = q >> 3, out0 = (q & 7) * 4;
    if (out0 < 16) {
      const int s = 4 * (in >> 4) + (in & 3), hq = (in >> 2) & 3;
      float *d = s_a1 + (s >> 2) * 256 + (out0 + 16 * hq) * 4 + (s & 3);
      d[0] = v.x;
      d[4] = v.y;
      d[8] = v.z;
      d[12] = v.w;
    }
  }
  // s_small: P1 [4][32], P1B [32], P2 [8][64], P2B [64], A1B [32], A2 [8][64], A2B [64]
  stage_f4<kSmA1B / 4, 512>(W_arg + VW_P1, s_small, [](int q) { return 4 * q; });
  stage_f4<(kSmallFloats - kSmA1B) / 4, 512>(W_arg + VW_A1B, s_small + kSmA1B, [](int q) { return 4 * q; });
  // the per-tile pieces too (q_fc, out_fc, LayerNorm, out_fc bias): from global memory they
  // cost an L2 round trip per 8-MFMA chunk of a tile's prologue and epilogue
  if (SPLIT) {
    stage_w64_bf16x3<512>(W_arg + VW_WQ, s_wq);
    stage_w64_bf16x3<512>(W_arg + VW_WO, s_wo);
  } else {
    stage_w64_lanes<512>(W_arg + VW_WQ, s_wq);
    stage_w64_lanes<512>(W_arg + VW_WO, s_wo);
  }
  stage_f4<32, 512>(W_arg + VW_LN1_G, s_par, [](int q) { return 4 * q; });        // gamma[64], beta[64]
  stage_f4<16, 512>(W_arg + VW_WOB, s_par + 128, [](int q) { return 4 * q; });    // out_fc bias[64]
  __syncthreads();
  const int lane = threadIdx.x & 63, i = lane & 15, hq = lane >> 4;
  const int wave = threadIdx.x >> 6;
  // per-lane bases of the A operands (lane-major images)
  const float *wq = s_wq + 4 * lane, *wo = s_wo + 4 * lane;
  const float *wk = s_wk + 4 * lane;
  const float *wv = s_wv + 4 * lane;
  const float *wa1 = s_a1 + 4 * lane;
  const float *sP1 = s_small, *sP1b = s_small + kSmP1B, *sP2 = s_small + kSmP2, *sP2b = s_small + kSmP2B;
  const float *sA1b = s_small + kSmA1B, *sA2 = s_small + kSmA2, *sA2b = s_small + kSmA2B;
  floatx4 p1b, a1b;  // the hidden layers' biases: accumulator operands of their first MFMAs
  const float p1w = sP1[hq * 32 + i];
  if (threadIdx.x < 256) {
    const int q = (int)threadIdx.x >> 6, ln = (int)threadIdx.x & 63, li = ln & 15, lq = ln >> 4;
    const int lhu = 4 * (lq & 1) + (lq >> 1), u = q & 1;
    float4 v;
    const float *src = (q < 2 ? sP2 : sA2) + (lhu + 2 * u) * 64 + li;
    const float sc = q < 2 ? 1.0f : kLog2e;  // logits in log2 units: exp2 without a multiply
    v.x = src[0] * sc;
    v.y = src[16] * sc;
    v.z = src[32] * sc;
    v.w = src[48] * sc;
    reinterpret_cast<float4 *>(s_pw)[q * 64 + ln] = v;
  }
  __syncthreads();
  const floatx4 *pw4 = reinterpret_cast<const floatx4 *>(s_pw) + lane;      // [q * 64]: p2w[0], p2w[1], a2w[0], a2w[1] (x log2e)
  floatx4 *c1q = reinterpret_cast<floatx4 *>(s_c1) + wave * 256 + lane;      // [mt * 64]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    p1b[r] = sP1b[4 * hq + r];
    a1b[r] = sA1b[4 * hq + r];
  }

  const int64_t ntiles = (N + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 8) {
    const int64_t g_raw = tile * 16 + i;
    const bool g_ok = g_raw < N;
    const int64_t g = g_ok ? g_raw : N - 1;
    float w[8];
    // the first view's row, validity flag and direction travel while q' is being formed
    float f_nx[16], d_nx;
    uint8_t ok_nx;
    load_row16(feat + (g * V) * 64, f_nx, hq);
    ok_nx = valid[g * V];
    d_nx = ray_diff[(g * V) * 4 + hq];
    // q' = Wq LN(q) enters every view as (pos - q'): c1 = P2b - q' is the accumulator the
    // positional MLP's second layer starts from, so pq = pos - q' costs nothing; then
    // a = k + pq, and the value product starts from pq as well: sum_v attn (vv + pos - q'),
    // to which q' is added back once in the epilogue (the attention weights sum to one).
    {
      float q0[16], x[16];
      load_row16(q_in + g * 64, q0, hq);
      layer_norm64q(q0, s_par, s_par + 64, 1e-6f, x, hq);
      floatx4 qq[4] = {};
      if (SPLIT) {
        chain64_bf16x3(qq, wq, x);
      } else {
        ldq8v(w, wq, 0);
        chain64qv(qq, wq, x, w, [&](float (&d)[8]) { ldq8v(d, wk, 0); });
      }
      float b[16];
      load_row16(sP2b, b, hq);
#pragma unroll
      for (int t = 0; t < 16; ++t) qq[t >> 2][t & 3] = b[t] - qq[t >> 2][t & 3];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) c1q[mt * 64] = qq[mt];  // c1, parked in this wavefront's LDS block
    }
    // bm = (A2's bias - reference logit m) in log2 units: the logits leave the second layer's MFMAs already
    // relative to the reference (no subtraction, no bias reload per view); m itself is never needed
    float bm[16], l[16], acc[16];
    float sk[16], sk2[16], sabs[16], ue[16];
    load_row16(sA2b, bm, hq);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      bm[t] *= kLog2e;
      l[t] = 0.0f;
      acc[t] = 0.0f;
      if (STATS) {
        sk[t] = 0.0f;
        sk2[t] = 0.0f;
        sabs[t] = 0.0f;
        ue[t] = 0.0f;
      }
    }
    int nvalid = 0;
    for (int v = 0; v < V; ++v) {
      const int64_t row = g * V + v;
      float k[16];
      const bool ok = ok_nx != 0;
      const float dv = d_nx;
      // When no group of the tile sees this source view (samples of a ray leave a view's frustum
      // together) its 161 MFMAs would only be multiplied by zero attention and skipped statistics.
      const bool seen = __builtin_amdgcn_ballot_w64(ok) != 0;
      if (seen) {  // k = Wk f
        floatx4 c[4] = {};
        if (SPLIT) {
          chain64_bf16x3(c, wk, f_nx);
          ldq8v(w, wa1, 0);
        } else {
          chain64qv(c, wk, f_nx, w, [&](float (&d)[8]) { ldq8v(d, wa1, 0); });
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) k[t] = c[t >> 2][t & 3];
      }
      // the next view's row, validity flag and direction are requested as soon as this view's
      // row has been consumed: ~100 MFMAs (x2 wavefronts per SIMD) cover the HBM latency
      if (v + 1 < V) {
        load_row16(feat + (row + 1) * 64, f_nx, hq);
        ok_nx = valid[row + 1];
        d_nx = ray_diff[(row + 1) * 4 + hq];
      }
      if (seen) {
      float a[16], hid[4], hk[2];
      floatx4 pq[4];
      {  // pq = P2 relu(P1 d + b) + b - q'   (4 -> 8 -> 64)
        const floatx4 c = mfma16(p1w, dv, p1b);
#pragma unroll
        for (int r = 0; r < 4; ++r) hid[r] = vrelu(c[r]);
        pack_hidden(hid, hk);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
          for (int u = 0; u < 2; ++u) pq[mt] = mfma16(pw4[u * 64][mt], hk[u], u == 0 ? c1q[mt * 64] : pq[mt]);
        }
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) a[t] = k[t] + pq[t >> 2][t & 3];
      {  // hidden layer of the attention MLP (64 -> 8, M padded to 16)
        const floatx4 c = chain64n(wa1, a, w, a1b, [&](float (&d)[8]) {
          if (!SPLIT) ldq8v(d, wv, 0);
        });
#pragma unroll
        for (int r = 0; r < 4; ++r) hid[r] = vrelu(c[r]);
        pack_hidden(hid, hk);
      }
      if (STATS && ok) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          sk[t] += k[t];
          sk2[t] = __builtin_fmaf(k[t], k[t], sk2[t]);
          sabs[t] += fabsf(k[t]);
        }
      }
      floatx4 lv[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) lv[mt] = pq[mt];
      if (SPLIT)
        chain64_bf16x3(lv, wv, k);
      else
        chain64qv(lv, wv, k, w, [&](float (&d)[8]) { ldq8v(d, wk, 0); });
      float x[16];  // logits relative to the reference, in log2 units
      {
        floatx4 la[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) la[mt][r] = bm[4 * mt + r];
#pragma unroll
          for (int u = 0; u < 2; ++u) la[mt] = mfma16(pw4[(2 + u) * 64][mt], hk[u], la[mt]);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) x[t] = la[t >> 2][t & 3];
      }
      const bool first = ok && nvalid == 0;
      if (__builtin_amdgcn_ballot_w64(first) != 0) {
        // a group's first valid view defines the reference: x = 0, e = 1 below
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          bm[t] -= first ? x[t] : 0.0f;
          x[t] = first ? 0.0f : x[t];
        }
      }
      nvalid += ok ? 1 : 0;
      float xmax = -__builtin_inff();
#pragma unroll
      for (int t = 0; t < 16; ++t) xmax = vmax2(xmax, x[t]);  // (v_med3: no canonicalising move per MFMA result)
      if (__builtin_amdgcn_ballot_w64(ok && xmax > kRescaleGap * kLog2e) != 0) {
        // (rare) move the reference up by d = max(x, 0): every running sum scales by 2^-d.
        // ue = sum_v 2^(a_v - m) (a_v - m) follows the change of reference as
        // 2^-d (ue - d l); the entropy of the final softmax is log(l) - ln2 ue / l (epilogue).
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const float d = ok ? fmaxf(x[t], 0.0f) : 0.0f;
          const float sc = __builtin_amdgcn_exp2f(-d);
          if (STATS) ue[t] = sc * (ue[t] - d * l[t]);
          l[t] *= sc;
          acc[t] *= sc;
          bm[t] -= d;
          x[t] -= d;
        }
      }
      if (ok) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const float e = __builtin_amdgcn_exp2f(x[t]);
          if (STATS) ue[t] = __builtin_fmaf(e, x[t], ue[t]);
          l[t] += e;
          acc[t] = __builtin_fmaf(e, lv[t >> 2][t & 3], acc[t]);
        }
      }
          }
    }
    // x = Wo (acc / l + q') + bo + q ;  q_out = FF(LN(x)) + x
    float x1[16];
    {
      float qres[16];
      load_row16(q_in + g * 64, qres, hq);  // (long evicted: requested before the out_fc product)
      float xa[16], b[16];
      load_row16(sP2b, b, hq);
#pragma unroll
      for (int t = 0; t < 16; ++t) xa[t] = acc[t] / l[t] + (b[t] - c1q[(t >> 2) * 64][t & 3]);
      floatx4 o[4];
      load_row16(s_par + 128, b, hq);
#pragma unroll
      for (int t = 0; t < 16; ++t) o[t >> 2][t & 3] = b[t];
      if (SPLIT) {
        chain64_bf16x3(o, wo, xa);
      } else {
        ldq8v(w, wo, 0);
        chain64qv(o, wo, xa, w, [&](float (&d)[8]) {});
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) x1[t] = o[t >> 2][t & 3] + qres[t];
    }
    if (g_ok) store_row16(q_out + g * 64, x1, hq);
    if (STATS) {
      // Entropy of the normalised attention, sum_v -p_v log(p_v + 1e-8) upstream (:497-500).
      // With p_v = exp(a_v - m) / l:  -sum p log p = log(l) - ue / l, accumulated online above
      // (no second sweep, no logit scratch); the 1e-8 inside the log shifts each valid view's
      // term by -1e-8 + O(1e-16 / p_v), i.e. by less than 2e-7 in total -- far inside the
      // fp32 noise of the upstream expression and the 1e-4 tolerance.  Then the masked
      // unbiased std of k and its normalised form; means over the 64 features.
      float ent = 0.0f, sd = 0.0f, sdn = 0.0f;
      if (nvalid > 0) {
#pragma unroll
        for (int t = 0; t < 16; ++t) ent += logf(l[t]) - 0.6931471805599453f * ue[t] / l[t] - 1e-8f * (float)nvalid;
      }
      if (nvalid > 1) {
        const float n = (float)nvalid;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const float mean = sk[t] / n;
          const float var = (sk2[t] - n * mean * mean) / (n - 1.0f);
          const float sdev = sqrtf(fmaxf(var, 0.0f));
          sd += sdev;
          sdn += sdev / (sabs[t] / n + 1e-6f);
        }
      }
      ent = quad_sum(ent);
      sd = quad_sum(sd);
      sdn = quad_sum(sdn);
      if (g_ok && hq == 0) {
        stats[g * 3 + 0] = ent * (1.0f / 64.0f);
        stats[g * 3 + 1] = sd * (1.0f / 64.0f);
        stats[g * 3 + 2] = sdn * (1.0f / 64.0f);
      }
    }
  }
}

// q_out = F2 relu(F1 LN(x) + b1) + b2 + x  (FeedForward + ff_norm + residual, :44-55,:218-221),
// in place on the rows written by the attention kernel.
__global__ void __launch_bounds__(512, 1)
gnt_ff_kernel(const float *__restrict__ W_arg, float *__restrict__ x_io, int64_t N) {
  // both weight matrices (128 KB) live in LDS for the lifetime of the (persistent) workgroup:
  // read from global memory every 32-row tile would pull them through L2 8192 times per
  // launch at 1024 rays x 256 samples.  ds_read_b32 serves each 32-lane half in one cycle
  // (consecutive columns), so no padding is needed.
  extern __shared__ __attribute__((aligned(16))) float s_ff[];  // F1 [64][256], F2 [256][64]
  stage_f4<4096, 512>(W_arg + VW_F1, s_ff, [](int q) { return 4 * q; });
  stage_f4<4096, 512>(W_arg + VW_F2, s_ff + 16384, [](int q) { return 4 * q; });
  // the biases as well: a global load in front of an accumulator chain costs an L2 round trip
  // that the one-chunk-ahead weight pipeline cannot cover          [F1B 256 | F2B 64 | LN2 128]
  stage_f4<64, 512>(W_arg + VW_F1B, s_ff + 32768, [](int q) { return 4 * q; });
  stage_f4<16, 512>(W_arg + VW_F2B, s_ff + 32768 + 256, [](int q) { return 4 * q; });
  stage_f4<32, 512>(W_arg + VW_LN2_G, s_ff + 32768 + 320, [](int q) { return 4 * q; });  // gamma[64], beta[64]
  __syncthreads();
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int64_t ntiles = (N + 31) / 32;
  // the next tile's rows are requested before this tile's 512 MFMAs (nothing else covers the HBM
  // latency: the vector ALU and the fp32 matrix pipe do not run concurrently)
  const int64_t tstep = (int64_t)gridDim.x * nwave;
  float x_nx[32];
  {
    const int64_t t0 = (int64_t)blockIdx.x * nwave + wave;
    const int64_t g0 = t0 * 32 + i;
    if (t0 < ntiles) load_row32(x_io + (g0 < N ? g0 : N - 1) * 64, x_nx, h);
  }
  for (int64_t tile = (int64_t)blockIdx.x * nwave + wave; tile < ntiles; tile += tstep) {
    const int64_t g_raw = tile * 32 + i;
    const bool g_ok = g_raw < N;
    const int64_t g = g_ok ? g_raw : N - 1;
    float x1[32], xn[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) x1[t] = x_nx[t];
    if (tile + tstep < ntiles) {
      const int64_t gn = (tile + tstep) * 32 + i;
      load_row32(x_io + (gn < N ? gn : N - 1) * 64, x_nx, h);
    }
    layer_norm64(x1, s_ff + 32768 + 320, s_ff + 32768 + 384, 1e-6f, xn, h);
    floatx16 o0, o1;
    {
      const float *bb = s_ff + 32768 + 256 + 4 * h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o0[r] = bb[featc(r)];
        o1[r] = bb[featc(r + 16)];
      }
    }
    // Hidden features are produced 32 at a time (tile mt) and consumed right away by the second
    // layer.  The first-layer chain of tile mt+1 is interleaved with the second-layer MFMAs of
    // tile mt: independent accumulators alternate in the matrix pipe, and each chunk's weights
    // (8 + 8 floats) are requested one chunk ahead.
    const float *f1 = s_ff + (4 * h) * 256 + i;
    const float *f2 = s_ff + 16384 + (4 * h) * 64 + i;  // rows = hidden features of this lane half
    const float *b1 = s_ff + 32768 + 4 * h;
    auto ld_f1 = [&](float (&w)[8], int mt, int c) {
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = f1[featc(8 * c + u) * 256 + 32 * mt];
    };
    auto ld_f2 = [&](float (&w)[8], int mt, int c) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = 4 * c + u;
        w[u] = f2[(32 * mt + (r & 3) + 8 * (r >> 2)) * 64];
        w[4 + u] = f2[(32 * mt + (r & 3) + 8 * (r >> 2)) * 64 + 32];
      }
    };
    floatx16 hcur;
#pragma unroll
    for (int r = 0; r < 16; ++r) hcur[r] = b1[(r & 3) + 8 * (r >> 2)];
    {
      float wa[8], wb[8];
      ld_f1(wa, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; c += 2) {
        ld_f1(wb, 0, c + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) hcur = mfma(wa[u], xn[8 * c + u], hcur);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 2 < 4) ld_f1(wa, 0, c + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) hcur = mfma(wb[u], xn[8 * (c + 1) + u], hcur);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float pa[8], qa[8], pb[8], qb[8];
    ld_f1(pa, 1, 0);
    ld_f2(qa, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      floatx16 hnext;
      if (mt < 7) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hnext[r] = b1[32 * (mt + 1) + (r & 3) + 8 * (r >> 2)];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float(&p)[8] = (c & 1) ? pb : pa;
        float(&q)[8] = (c & 1) ? qb : qa;
        float(&pn)[8] = (c & 1) ? pa : pb;
        float(&qn)[8] = (c & 1) ? qa : qb;
        // next chunk: same tile pair, or chunk 0 of the next pair
        const int nmt = c < 3 ? mt : mt + 1, nc = c < 3 ? c + 1 : 0;
        if (nmt < 7) ld_f1(pn, nmt + 1, nc);
        if (nmt < 8) ld_f2(qn, nmt, nc);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (mt < 7) hnext = mfma(p[u], xn[8 * c + u], hnext);
          if (u < 4) {
            const float hv = vrelu(hcur[4 * c + u]);
            o0 = mfma(q[u], hv, o0);
            o1 = mfma(q[4 + u], hv, o1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (mt < 7) hcur = hnext;
    }
    if (g_ok) {
      float out[32];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        out[r] = o0[r] + x1[r];
        out[r + 16] = o1[r] + x1[r + 16];
      }
      store_row32(x_io + g * 64, out, h);
    }
  }
}

// ---------------------------------------------------------------------------------------
// The feed-forward block on the bf16 matrix pipe with fp32 results (round 4): every product as six partial products of
// exact bf16x3 pieces (gnt_mfma.h), on v_mfma_f32_32x32x16_bf16.  Three pieces of the 32 768 weights are 192 KB, LDS holds
// 160: the eight wavefronts of a workgroup take their eight tiles through TWO phases per round, with the pieces of half of
// the hidden units resident (96 KB: the rows of layer 1 and the columns of layer 2 that belong to those 128 units), the
// output accumulators kept across the phases; between the phases the other half is copied in from the weight blob's
// pre-split image (ops.ff_bf16x3_images), and the next round starts with the half that is resident.
// Layouts (32-row tiles as in gnt_ff_kernel: lane (i, h) keeps features featc(t) + 4 h of row i): A and B operands of the
// 32 x 32 x 16 instruction carry K index 8 (lane >> 5) + j in element j.  Layer 1, K-step c: the lane supplies its own
// features t = 8 c + j (split in registers), the image supplies W1[featc(8 c + j) + 4 kg][32 mt + m].  The accumulator leaves
// hidden unit (r & 3) + 8 (r >> 2) + 4 h + 32 mt in register r -- layer 2's K-step c' takes the lane's registers 8 c' + j.
// ---------------------------------------------------------------------------------------
typedef float floatx16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ floatx16v mfma32_bf16(uintx4 a, uintx4 b, floatx16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// six partial products of one (weight, activation) operand pair, small ones first
__device__ __forceinline__ floatx16v mfma32_bf16x3(const uintx4 (&w)[3], const uintx4 &xh, const uintx4 &xm, const uintx4 &xl,
                                                   floatx16v c) {
  c = mfma32_bf16(w[2], xh, c);
  c = mfma32_bf16(w[0], xl, c);
  c = mfma32_bf16(w[1], xm, c);
  c = mfma32_bf16(w[1], xh, c);
  c = mfma32_bf16(w[0], xm, c);
  c = mfma32_bf16(w[0], xh, c);
  return c;
}

constexpr int kFfHalfU4 = 6144;  // 16-byte units of one half image: 3 pieces x (1024 of layer 1 + 1024 of layer 2)

__global__ void __launch_bounds__(512, 1)
gnt_ff_bf16x3_kernel(const float *__restrict__ W_arg, float *__restrict__ x_io, int64_t N) {
  extern __shared__ __attribute__((aligned(16))) float s_ffb[];  // [24576: the resident half image][b1 256 | b2 64 | LN 128]
  uintx4 *s_img = reinterpret_cast<uintx4 *>(s_ffb);
  float *s_par = s_ffb + 4 * kFfHalfU4;
  const uintx4 *g_img = reinterpret_cast<const uintx4 *>(W_arg + VW_FFIMG);
  auto stage_half = [&](int hf) {
    uintx4 v[kFfHalfU4 / 512];
#pragma unroll
    for (int k = 0; k < kFfHalfU4 / 512; ++k) v[k] = g_img[hf * kFfHalfU4 + (int)threadIdx.x + 512 * k];
#pragma unroll
    for (int k = 0; k < kFfHalfU4 / 512; ++k) s_img[(int)threadIdx.x + 512 * k] = v[k];
  };
  stage_half(0);
  stage_f4<64, 512>(W_arg + VW_F1B, s_par, [](int q) { return 4 * q; });
  stage_f4<16, 512>(W_arg + VW_F2B, s_par + 256, [](int q) { return 4 * q; });
  stage_f4<32, 512>(W_arg + VW_LN2_G, s_par + 320, [](int q) { return 4 * q; });  // gamma[64], beta[64]
  __syncthreads();
  int resident = 0;
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const int64_t ntiles = (N + 31) / 32;
  // the next round's rows are requested before this round's 384 MFMAs (nothing else covers the HBM latency)
  float x_nx[32];
  {
    const int64_t g0 = ((int64_t)blockIdx.x * 8 + wave) * 32 + i;
    load_row32(x_io + (g0 < N ? g0 : N - 1) * 64, x_nx, h);
  }
  for (int64_t base = (int64_t)blockIdx.x * 8; base < ntiles; base += (int64_t)gridDim.x * 8) {  // (uniform over the workgroup)
    const int64_t tile = base + wave;
    const bool t_ok = tile < ntiles;
    const int64_t g_raw = tile * 32 + i;
    const bool g_ok = t_ok && g_raw < N;
    const int64_t g = g_raw < N ? g_raw : N - 1;
    float x1[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) x1[t] = x_nx[t];
    {
      const int64_t gn = (tile + (int64_t)gridDim.x * 8) * 32 + i;
      load_row32(x_io + (gn < N ? gn : N - 1) * 64, x_nx, h);
    }
    // the row, normalised, split once for both phases: K-step c of layer 1 takes registers 8 c .. 8 c + 7
    uintx4 xh[4], xm[4], xl[4];
    {
      float xn[32];
      layer_norm64(x1, s_par + 320, s_par + 384, 1e-6f, xn, h);
#pragma unroll
      for (int c = 0; c < 4; ++c) split8_bf16x3(&xn[8 * c], xh[c], xm[c], xl[c]);
    }
    floatx16v o0, o1;
    {
      const float *bb = s_par + 256 + 4 * h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o0[r] = bb[featc(r)];
        o1[r] = bb[featc(r + 16)];
      }
    }
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
      const int hf = ph == 0 ? resident : 1 - resident;
      if (ph == 1) {
        __syncthreads();  // every wavefront is through with the resident half
        stage_half(hf);
        __syncthreads();
      }
      const uintx4 *img = s_img + lane;
#pragma unroll 1
      for (int mtl = 0; mtl < 4; mtl += 2) {  // two hidden tiles at a time: their accumulators take turns in the matrix pipe
        const int mt = 4 * hf + mtl;
        floatx16v hacc0, hacc1;
        {
          const float *b1 = s_par + 32 * mt + 4 * h;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            hacc0[r] = b1[(r & 3) + 8 * (r >> 2)];
            hacc1[r] = b1[32 + (r & 3) + 8 * (r >> 2)];
          }
        }
        // layer 1: four K-steps, the next step's weight pieces requested while this step's twelve MFMAs run
        uintx4 wa0[3], wa1[3], wb0[3], wb1[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          wa0[p] = img[p * 2048 + (mtl * 4 + 0) * 64];
          wa1[p] = img[p * 2048 + ((mtl + 1) * 4 + 0) * 64];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (c + 1 < 4) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
              wb0[p] = img[p * 2048 + (mtl * 4 + c + 1) * 64];
              wb1[p] = img[p * 2048 + ((mtl + 1) * 4 + c + 1) * 64];
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          hacc0 = mfma32_bf16(wa0[2], xh[c], hacc0);
          hacc1 = mfma32_bf16(wa1[2], xh[c], hacc1);
          hacc0 = mfma32_bf16(wa0[0], xl[c], hacc0);
          hacc1 = mfma32_bf16(wa1[0], xl[c], hacc1);
          hacc0 = mfma32_bf16(wa0[1], xm[c], hacc0);
          hacc1 = mfma32_bf16(wa1[1], xm[c], hacc1);
          hacc0 = mfma32_bf16(wa0[1], xh[c], hacc0);
          hacc1 = mfma32_bf16(wa1[1], xh[c], hacc1);
          hacc0 = mfma32_bf16(wa0[0], xm[c], hacc0);
          hacc1 = mfma32_bf16(wa1[0], xm[c], hacc1);
          hacc0 = mfma32_bf16(wa0[0], xh[c], hacc0);
          hacc1 = mfma32_bf16(wa1[0], xh[c], hacc1);
          __builtin_amdgcn_sched_barrier(0);
          if (c + 1 < 4) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
              wa0[p] = wb0[p];
              wa1[p] = wb1[p];
            }
          }
        }
        // ReLU, split, layer 2: per hidden tile two K-steps x two output tiles
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          float hv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) hv[r] = vrelu(sub == 0 ? hacc0[r] : hacc1[r]);
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2) {
            uintx4 hh, hm, hl;
            split8_bf16x3(&hv[8 * c2], hh, hm, hl);
            uintx4 w0[3], w1[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
              w0[p] = img[p * 2048 + 1024 + (((mtl + sub) * 2 + c2) * 2 + 0) * 64];
              w1[p] = img[p * 2048 + 1024 + (((mtl + sub) * 2 + c2) * 2 + 1) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            // (the two output tiles take turns: consecutive MFMAs never wait for each other's result)
            o0 = mfma32_bf16(w0[2], hh, o0);
            o1 = mfma32_bf16(w1[2], hh, o1);
            o0 = mfma32_bf16(w0[0], hl, o0);
            o1 = mfma32_bf16(w1[0], hl, o1);
            o0 = mfma32_bf16(w0[1], hm, o0);
            o1 = mfma32_bf16(w1[1], hm, o1);
            o0 = mfma32_bf16(w0[1], hh, o0);
            o1 = mfma32_bf16(w1[1], hh, o1);
            o0 = mfma32_bf16(w0[0], hm, o0);
            o1 = mfma32_bf16(w1[0], hm, o1);
            o0 = mfma32_bf16(w0[0], hh, o0);
            o1 = mfma32_bf16(w1[0], hh, o1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      resident = hf;
    }
    if (g_ok) {
      float out[32];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        out[r] = o0[r] + x1[r];
        out[r + 16] = o1[r] + x1[r + 16];
      }
      store_row32(x_io + g * 64, out, h);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Ray transformer attention (Transformer + Attention, attn_mode="qk",
// transformer_network.py:231-338): per ray, x = LN(q); Q,K,V = W x; 4 heads x 16 dims;
// attn = softmax(Q K^T / 4) over the S samples of the ray; y = Wo (attn V) + bo + q.
// One workgroup per ray, one wavefront per tile of 16 query samples, 16x16x4 MFMA (layout:
// gnt_mfma.h).  A head is 16 features = exactly one M tile, so neither the score product
// (keys x dims) nor the P.V product (dims x keys) carries padding.  K ([feature][sample]) and
// V ([sample][feature]) of the whole ray live in LDS; the 16x16 score tile leaves the MFMA with
// 4 keys per lane and the query on the lane -- the B operand of the P.V product, so the
// probabilities never leave registers.  Q is pre-scaled by log2(e)/4: probabilities are
// exp2(score - m) with a per-query reference m that is only moved when a score exceeds it by
// kRayGap (see kRescaleGap above).  Also emits the head-averaged attention row of query
// sample 0, the "learned density" the renderer uses as sample weights (:336).
// Uses the VW_* weight offsets: LN1 = attn_norm, WQ/WK/WV, WO/WOB = out_fc (FF via gnt_ff).
// ---------------------------------------------------------------------------------------
constexpr int kRayKStride = 260;  // K rows [feature][sample] padded: 4 * 260 = 16 mod 32 banks
constexpr int kRayVStride = 68;   // V rows [sample][feature] padded likewise
constexpr int kRaySmax = 256;
constexpr float kRayGap = 20.0f;  // in log2 units
constexpr int kRayWStride = 68;   // the projection matrix staged in LDS, rows padded likewise
constexpr int kRayLdsFloats = 64 * kRayKStride + kRaySmax * kRayVStride + 72 + 192 + 64 * kRayWStride;

__device__ __forceinline__ float quad_max(float v, int lane) {  // over the lanes j, j+16, j+32, j+48
  const int iv = __float_as_int(v);
  const auto a = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);
  v = fmaxf(v, __int_as_float((lane & 16) ? a[0] : a[1]));
  const int iw = __float_as_int(v);
  const auto b = __builtin_amdgcn_permlane32_swap(iw, iw, false, false);
  return fmaxf(v, __int_as_float((lane & 32) ? b[0] : b[1]));
}

__global__ void __launch_bounds__(1024)
gnt_ray_attn_kernel(const float *__restrict__ W_arg, const float *__restrict__ q_in, int R, int S,
                    float *__restrict__ y_out, float *__restrict__ w_out) {
  extern __shared__ __attribute__((aligned(16))) float s_kv[];
  float *Ks = s_kv;                              // [64][kRayKStride]
  float *Vs = s_kv + 64 * kRayKStride;           // [kRaySmax][kRayVStride]
  float *s_row0 = Vs + kRaySmax * kRayVStride;   // [8 + 64]: (m,l) per head, scaled Q of sample 0
  float *s_par = s_row0 + 72;                    // LN gamma[64], beta[64], out_fc bias[64]
  float *s_wst = s_par + 192;                    // [64][kRayWStride]: the projection matrix in use
  const int ntile = (S + 15) / 16;
  const int lane = threadIdx.x & 63, i = lane & 15, hq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool has_tile = wave < ntile;
  // The four 64x64 matrices take turns in one LDS buffer: every thread carries one float4 of the
  // NEXT matrix in registers (requested a whole product earlier, so the L2 latency is long
  // gone) and drops it into the buffer between two barriers.  Read straight from global memory
  // by each of the 16 wavefronts, the weights cost an L2 round trip per 8-MFMA chunk.
  const int stage_dst = ((int)threadIdx.x >> 4) * kRayWStride + 4 * ((int)threadIdx.x & 15);
  auto fetch = [&](int off) { return reinterpret_cast<const float4 *>(W_arg + off)[threadIdx.x]; };
  if (threadIdx.x < 32) reinterpret_cast<float4 *>(s_par)[threadIdx.x] = reinterpret_cast<const float4 *>(W_arg + VW_LN1_G)[threadIdx.x];
  if (threadIdx.x < 16) reinterpret_cast<float4 *>(s_par + 128)[threadIdx.x] = reinterpret_cast<const float4 *>(W_arg + VW_WOB)[threadIdx.x];
  const float *wst = s_wst + (4 * hq) * kRayWStride + i;
  float4 pf = fetch(VW_WQ);
  for (int ray = blockIdx.x; ray < R; ray += gridDim.x) {
    const int s_raw = wave * 16 + i;
    const int smp = s_raw < S ? s_raw : S - 1;  // lanes past the end recompute the last sample: finite K/V
    const float *xrow = q_in + ((int64_t)ray * S + smp) * 64;
    float qv[16], xn[16], w[8];
    __syncthreads();  // previous ray: K/V and the weight buffer (Wo) fully consumed
    *reinterpret_cast<float4 *>(s_wst + stage_dst) = pf;  // Wq
    pf = fetch(VW_WK);
    if (has_tile) {
      float x[16];
      load_row16(xrow, x, hq);
      layer_norm64q(x, s_par, s_par + 64, 1e-6f, xn, hq);
    }
    __syncthreads();
    floatx4 c[4] = {};
    if (has_tile) {
      ldq8<kRayWStride>(w, wst, 0);
      chain64q<kRayWStride>(c, wst, xn, w, [&](float (&d)[8]) {});
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        qv[t] = c[t >> 2][t & 3] * (0.25f * kLog2e);
        c[t >> 2][t & 3] = 0.0f;
      }
    }
    __syncthreads();
    *reinterpret_cast<float4 *>(s_wst + stage_dst) = pf;  // Wk
    pf = fetch(VW_WV);
    __syncthreads();
    if (has_tile) {
      ldq8<kRayWStride>(w, wst, 0);
      chain64q<kRayWStride>(c, wst, xn, w, [&](float (&d)[8]) {});
      float *kw = Ks + (4 * hq) * kRayKStride + wave * 16 + i;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        kw[(16 * (t >> 2) + (t & 3)) * kRayKStride] = c[t >> 2][t & 3];
        c[t >> 2][t & 3] = 0.0f;
      }
    }
    __syncthreads();
    *reinterpret_cast<float4 *>(s_wst + stage_dst) = pf;  // Wv
    pf = fetch(VW_WO);
    __syncthreads();
    if (has_tile) {
      ldq8<kRayWStride>(w, wst, 0);
      chain64q<kRayWStride>(c, wst, xn, w, [&](float (&d)[8]) {});
      float vv[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) vv[t] = c[t >> 2][t & 3];
      store_row16(Vs + (wave * 16 + i) * kRayVStride, vv, hq);
    }
    __syncthreads();
    *reinterpret_cast<float4 *>(s_wst + stage_dst) = pf;  // Wo
    pf = fetch(VW_WQ);
    __syncthreads();  // K, V of the whole ray and Wo in place
    if (has_tile) {
      // y = Wo (attention output) + bo accumulates head by head: each head contributes its 16
      // features as 4 K-steps, so the attention output itself is never materialised
      floatx4 y[4];
      {
        float b[16];
        load_row16(s_par + 128, b, hq);
#pragma unroll
        for (int t = 0; t < 16; ++t) y[t >> 2][t & 3] = b[t];
      }
      const float *wo = wst;
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) {
        const float *kr = Ks + (16 * hh + 4 * hq) * kRayKStride + i;  // + r * stride + 16 * kt
        const float *vr = Vs + (4 * hq) * kRayVStride + 16 * hh + i;  // + (16 * kt + r) * stride
        float m = 0.0f, l = 0.0f;
        floatx4 O = {0.0f, 0.0f, 0.0f, 0.0f};
        // the score tile of key tile kt+1 is issued before the softmax of tile kt: the matrix
        // pipe works on it while the vector ALU exponentiates, and its K operands were
        // requested one tile earlier still.  From the second tile on the score product starts from
        // -m (the accumulator operand of its first MFMA), so the scores arrive relative to the
        // reference: no subtraction per score (the vector ALU and the fp32 matrix pipe do not overlap
        // on this chip: every vector instruction in this loop is paid in full).
        float ka[4], va[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ka[r] = kr[r * kRayKStride];
        floatx4 sa = {0.0f, 0.0f, 0.0f, 0.0f}, sb;
#pragma unroll
        for (int r = 0; r < 4; ++r) sa = mfma16(ka[r], qv[4 * hh + r], sa);
        const float *kp = kr + 16, *vp = vr;  // walking pointers: every read is pointer + immediate
        if (1 < ntile) {
#pragma unroll
          for (int r = 0; r < 4; ++r) ka[r] = kp[r * kRayKStride];
          kp += 16;
        }
        floatx4 negm = {0.0f, 0.0f, 0.0f, 0.0f};
        // one key tile: `cur` holds its scores, `nxt` receives the next tile's (the two trade places from tile
        // to tile: the loop below is unrolled by two so that no accumulator is ever copied)
        auto key_tile = [&](floatx4 &cur, floatx4 &nxt, int kt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) va[r] = vp[r * kRayVStride];
          vp += 16 * kRayVStride;
          nxt = negm;
          if (kt + 1 < ntile) {
#pragma unroll
            for (int r = 0; r < 4; ++r) nxt = mfma16(ka[r], qv[4 * hh + r], nxt);
            if (kt + 2 < ntile) {
#pragma unroll
              for (int r = 0; r < 4; ++r) ka[r] = kp[r * kRayKStride];
              kp += 16;
            }
          }
          if (16 * kt + 16 > S) {  // the last, partial key tile (never when S is a multiple of 16)
#pragma unroll
            for (int r = 0; r < 4; ++r) cur[r] = 16 * kt + 4 * hq + r < S ? cur[r] : -__builtin_inff();
          }
          const float xmax = vmax2(vmax2(cur[0], cur[1]), vmax2(cur[2], cur[3]));
          if (kt == 0) {
            // raw scores: their maximum over the tile becomes the reference (key 0 is always valid: finite)
            m = quad_max(xmax, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              cur[r] -= m;
              nxt[r] -= m;
              negm[r] = -m;
            }
          } else if (__builtin_amdgcn_ballot_w64(xmax > kRayGap) != 0) {
            // (rare) a score left the reference behind by more than the gap: move it up by d
            const float d = fmaxf(quad_max(xmax, lane), 0.0f);
            const float f = __builtin_amdgcn_exp2f(-d);
            l *= f;
            m += d;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              O[r] *= f;
              cur[r] -= d;
              nxt[r] -= d;
              negm[r] -= d;
            }
          }
          float e[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(cur[r]);
          l += (e[0] + e[1]) + (e[2] + e[3]);
#pragma unroll
          for (int r = 0; r < 4; ++r) O = mfma16(va[r], e[r], O);
        };
        int kt = 0;
        for (; kt + 1 < ntile; kt += 2) {
          key_tile(sa, sb, kt);
          key_tile(sb, sa, kt + 1);
        }
        if (kt < ntile) key_tile(sa, sb, kt);
        l = quad_sum(l);
        if (wave == 0 && i == 0) {  // query sample 0: softmax statistics + its scaled Q for the weight row
          if (hq == 0) {
            s_row0[hh * 2 + 0] = m;
            s_row0[hh * 2 + 1] = l;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) s_row0[8 + hh * 16 + 4 * hq + r] = qv[4 * hh + r];
        }
        const float inv_l = 1.0f / l;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float av = O[r] * inv_l;  // feature 16*hh + 4*hq + r of the attention output
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) y[mt] = mfma16(wo[(16 * hh + r) * kRayWStride + 16 * mt], av, y[mt]);
        }
      }
      float xres[16];
      load_row16(xrow, xres, hq);
      if (s_raw < S) {
#pragma unroll
        for (int t = 0; t < 16; ++t) xres[t] += y[t >> 2][t & 3];
        store_row16(y_out + ((int64_t)ray * S + s_raw) * 64, xres, hq);
      }
    }
    if (w_out != nullptr) {
      // attention row of query sample 0 (head average) for all keys: one key per thread,
      // scores recomputed on the vector ALU from K in LDS and the saved (m, l, Q) of sample 0
      __syncthreads();
      const int key = threadIdx.x;
      if (key < S) {
        float wsum = 0.0f;
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
          float sdot = 0.0f;
#pragma unroll
          for (int d = 0; d < 16; ++d) sdot += s_row0[8 + hh * 16 + d] * Ks[(16 * hh + d) * kRayKStride + key];
          wsum += __builtin_amdgcn_exp2f(sdot - s_row0[hh * 2]) / s_row0[hh * 2 + 1];
        }
        w_out[(int64_t)ray * S + key] = wsum * 0.25f;  // mean over the 4 heads
      }
    }
  }
}

}  // namespace pgdvs

using namespace pgdvs;

PGDVS_API int64_t pgdvs_gnt_view_weight_floats(void) { return VW_TOTAL; }

// feed-forward block in place on x[N,64]: one persistent 8-wave workgroup per CU
static int launch_ff(const float *weights, float *x, int64_t N, hipStream_t st) {
  const bool fp32_path = option_int(options().gnt_fp32) != 0;  // (see pgdvs_gnt_view_layer)
  const size_t lds = fp32_path ? (2 * 16384 + 256 + 64 + 128) * sizeof(float) : (4 * (size_t)kFfHalfU4 + 256 + 64 + 128) * sizeof(float);
  static bool configured = false;
  if (!configured) {
    for (const void *fn : {reinterpret_cast<const void *>(gnt_ff_kernel), reinterpret_cast<const void *>(gnt_ff_bf16x3_kernel)}) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((2 * 16384 + 256 + 64 + 128) * sizeof(float)));
      if (e != hipSuccess) {
        set_error("gnt_ff: cannot reserve LDS: %s", hipGetErrorString(e));
        return PGDVS_ERR_LAUNCH;
      }
    }
    configured = true;
  }
  const int64_t ntiles = cdiv(N, 32);
  const unsigned grid = (unsigned)(cdiv(ntiles, 8) < 256 ? cdiv(ntiles, 8) : 256);
  if (fp32_path) {
    PGDVS_LAUNCH("gnt_ff", gnt_ff_kernel, dim3(grid), dim3(512), lds, st, weights, x, N);
  } else {
    PGDVS_LAUNCH("gnt_ff", gnt_ff_bf16x3_kernel, dim3(grid), dim3(512), lds, st, weights, x, N);
  }
  return PGDVS_OK;
}

PGDVS_API int pgdvs_gnt_view_layer(const float *weights, const float *q_in, const float *feat,
                                   const float *ray_diff, const uint8_t *valid, int64_t N, int V,
                                   float *q_out, float *stats, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(weights && q_in && feat && ray_diff && valid && q_out, "pgdvs_gnt_view_layer: null pointer");
  PGDVS_REQUIRE(N >= 0 && V >= 1, "pgdvs_gnt_view_layer: bad shape");
  if (N == 0) return PGDVS_OK;
  const int64_t vtiles = cdiv(N, 16);
  const unsigned grid = (unsigned)(cdiv(vtiles, 8) < 256 ? cdiv(vtiles, 8) : 256);
  const size_t lds = (size_t)kViewLdsFloats * sizeof(float);
  hipStream_t st = as_stream(stream);
  static bool configured = false;
  if (!configured) {
    for (const void *fn : {reinterpret_cast<const void *>(gnt_view_layer_kernel<true, true>),
                           reinterpret_cast<const void *>(gnt_view_layer_kernel<false, true>),
                           reinterpret_cast<const void *>(gnt_view_layer_kernel<true, false>),
                           reinterpret_cast<const void *>(gnt_view_layer_kernel<false, false>)}) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) {
        set_error("gnt_view_layer: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
        return PGDVS_ERR_LAUNCH;
      }
    }
    configured = true;
  }
  // option gnt_fp32 (PGDVS_GNT_FP32=1 at load time, pgdvs_option_set afterwards): the k / v products on the fp32 matrix
  // instruction as well (exact fp32 products; the default splits both operands into three bf16 pieces: six partial
  // products, fp32 accumulation -- see gnt_mfma.h)
  const bool fp32_env = option_int(options().gnt_fp32) != 0;
#define PGDVS_VIEW_LAUNCH(ST, SP)                                                                                        \
  PGDVS_LAUNCH("gnt_view_layer", (gnt_view_layer_kernel<ST, SP>), dim3(grid), dim3(512), lds, st, weights, q_in, feat, \
               ray_diff, valid, N, V, q_out, stats)
  if (stats) {
    if (fp32_env) {
      PGDVS_VIEW_LAUNCH(true, false);
    } else {
      PGDVS_VIEW_LAUNCH(true, true);
    }
  } else {
    if (fp32_env) {
      PGDVS_VIEW_LAUNCH(false, false);
    } else {
      PGDVS_VIEW_LAUNCH(false, true);
    }
  }
#undef PGDVS_VIEW_LAUNCH
  if (launch_ff(weights, q_out, N, st) != PGDVS_OK) return PGDVS_ERR_LAUNCH;
  return check_launch("gnt_view_layer");
}

PGDVS_API int pgdvs_gnt_ray_layer(const float *weights, const float *q_in, int R, int S, float *q_out,
                                  float *sample_weights, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(weights && q_in && q_out, "pgdvs_gnt_ray_layer: null pointer");
  PGDVS_REQUIRE(R >= 0 && S >= 1 && S <= 256, "pgdvs_gnt_ray_layer: samples per ray must be in [1, 256]");
  if (R == 0) return PGDVS_OK;
  hipStream_t st = as_stream(stream);
  const size_t lds = (size_t)kRayLdsFloats * sizeof(float);
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gnt_ray_attn_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("gnt_ray_layer: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
      return PGDVS_ERR_LAUNCH;
    }
    configured = true;
  }
  const unsigned grid = (unsigned)(R < 1024 ? R : 1024);
  PGDVS_LAUNCH("gnt_ray_attn", gnt_ray_attn_kernel, dim3(grid), dim3(1024), lds, st, weights, q_in, R, S, q_out,
               sample_weights);
  const int64_t N = (int64_t)R * S;
  if (launch_ff(weights, q_out, N, st) != PGDVS_OK) return PGDVS_ERR_LAUNCH;
  return check_launch("gnt_ray_layer");
}
