// A14 (view transformer): one fused kernel per GNT layer for Transformer2D + Attention2D
// (pgdvs/models/gnt/models/transformer_network.py:59-169,197-223) on the fp32 matrix cores.
//
//   x = LN(q); q' = Wq x
//   for every source view v:  k = Wk f_v ; vv = Wv k ; pos = P2 relu(P1 d_v + b) + b ;
//                             a = A2 relu(A1 (k - q' + pos) + b) + b          (64 -> 8 -> 64)
//   attn = softmax_v(a) (masked);  x = Wo sum_v (vv + pos) * attn + bo + q
//   q_out = F2 relu(F1 LN(x) + b) + b + x
//
// MI355X mapping.  Everything is computed TRANSPOSED: the 64 features run along the MFMA M
// dimension, 32 (ray,sample) groups along N, so that
//   * weights are the A operand: lane (i, h) reads Wt[in = feature(t,h)][out = 32*mt + i],
//     a contiguous 128-byte row segment per half-wave (weights are stored input-major);
//   * activations are the B operand: lane (j, h) keeps, for ITS group j, the 32 features
//     feature(t,h) = (t&3) + 8*((t&15)>>2) + 4*h + 32*(t>>4), t = 0..31, in registers;
//   * the accumulator layout of v_mfma_f32_32x32x2_f32 (row = (r&3) + 8*(r>>2) + 4*h) is the
//     same feature(t,h) map, so a layer's output registers ARE the next layer's B operand --
//     the whole chain (k -> vv, k -> attention MLP, FF) never leaves registers, no LDS
//     transposes;
//   * the softmax over views is a per-lane online softmax (running max / sum / weighted sum
//     per feature), no cross-lane traffic; the view loop streams each f_v row once.
// f32-input MFMA (exact fp32 products, fp32 accumulate): TF32 is off in the reference
// (pgdvs/run.py:21-24) and outputs must agree to 1e-4.
#include "common.h"

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
constexpr int VW_TOTAL = VW_F2B + 64;
// the per-view hot weights are staged in LDS
constexpr int VW_LDS_BEGIN = VW_WK;
constexpr int VW_LDS_END = VW_WO;
constexpr int VW_LDS_FLOATS = VW_LDS_END - VW_LDS_BEGIN;

__device__ __forceinline__ int feat_of(int t, int h) {
  return (t & 3) + 8 * ((t & 15) >> 2) + 4 * h + 32 * (t >> 4);
}

__device__ __forceinline__ floatx16 mfma(float a, float b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// y[32] = W x (+ bias): 64 -> 64.  Wt input-major [64][64]; x, y in feature(t,h) layout.
__device__ __forceinline__ void lin64x64(const float *__restrict__ Wt, const float *__restrict__ bias,
                                         const float (&x)[32], float (&y)[32], int i, int h) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias ? bias[feat_of(r + 16 * mt, h)] : 0.0f;
#pragma unroll
    for (int t0 = 0; t0 < 32; t0 += 8) {
      // keep at most 8 weight registers live: the scheduler must not hoist all 64 loads
      float w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = Wt[feat_of(t0 + u, h) * 64 + mt * 32 + i];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = mfma(w[u], x[t0 + u], acc);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r + 16 * mt] = acc[r];
  }
}

// hid[4] = relu(W x + b): 64 -> 8 (M padded to 32).  Wt [64][32]; lane-half h gets hidden 4h..4h+3
__device__ __forceinline__ void lin64x8_relu(const float *__restrict__ Wt, const float *__restrict__ bias,
                                             const float (&x)[32], float (&hid)[4], int i, int h) {
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = r < 4 ? bias[r + 4 * h] : 0.0f;
#pragma unroll
  for (int t0 = 0; t0 < 32; t0 += 8) {
    float w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = Wt[feat_of(t0 + u, h) * 32 + i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = mfma(w[u], x[t0 + u], acc);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) hid[r] = fmaxf(acc[r], 0.0f);
}

// y[32] = W hid + b: 8 -> 64.  Wt [8][64]
__device__ __forceinline__ void lin8x64(const float *__restrict__ Wt, const float *__restrict__ bias,
                                        const float (&hid)[4], float (&y)[32], int i, int h) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias[feat_of(r + 16 * mt, h)];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = mfma(Wt[(t + 4 * h) * 64 + mt * 32 + i], hid[t], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r + 16 * mt] = acc[r];
  }
}

// hid[4] = relu(W d + b): 4 -> 8.  Wt [4][32]; lane-half h supplies d[2t+h]
__device__ __forceinline__ void lin4x8_relu(const float *__restrict__ Wt, const float *__restrict__ bias,
                                            const float (&d2)[2], float (&hid)[4], int i, int h) {
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = r < 4 ? bias[r + 4 * h] : 0.0f;
#pragma unroll
  for (int t = 0; t < 2; ++t) acc = mfma(Wt[(2 * t + h) * 32 + i], d2[t], acc);
#pragma unroll
  for (int r = 0; r < 4; ++r) hid[r] = fmaxf(acc[r], 0.0f);
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
#pragma unroll
  for (int t = 0; t < 32; ++t) {
    int f = feat_of(t, h);
    y[t] = (x[t] - mean) * rstd * g[f] + b[f];
  }
}

__device__ __forceinline__ void load_row32(const float *__restrict__ row, float (&x)[32], int h) {
  // features feature(t,h): groups of 4 contiguous floats -> 8 x dwordx4
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float4 v = *reinterpret_cast<const float4 *>(row + feat_of(c * 4, h));
    x[c * 4 + 0] = v.x;
    x[c * 4 + 1] = v.y;
    x[c * 4 + 2] = v.z;
    x[c * 4 + 3] = v.w;
  }
}

template <bool STATS>
__global__ void __launch_bounds__(256, 1)
gnt_view_layer_kernel(const float *__restrict__ W, const float *__restrict__ q_in,
                      const float *__restrict__ feat, const float *__restrict__ ray_diff,
                      const uint8_t *__restrict__ valid, int64_t N, int V, float *__restrict__ q_out,
                      float *__restrict__ stats, float *__restrict__ logit_scratch) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];  // [VW_LDS_FLOATS]
  for (int k = threadIdx.x; k < VW_LDS_FLOATS; k += blockDim.x) s_w[k] = W[VW_LDS_BEGIN + k];
  __syncthreads();
  const float *sWk = s_w + (VW_WK - VW_LDS_BEGIN), *sWv = s_w + (VW_WV - VW_LDS_BEGIN);
  const float *sP1 = s_w + (VW_P1 - VW_LDS_BEGIN), *sP1b = s_w + (VW_P1B - VW_LDS_BEGIN);
  const float *sP2 = s_w + (VW_P2 - VW_LDS_BEGIN), *sP2b = s_w + (VW_P2B - VW_LDS_BEGIN);
  const float *sA1 = s_w + (VW_A1 - VW_LDS_BEGIN), *sA1b = s_w + (VW_A1B - VW_LDS_BEGIN);
  const float *sA2 = s_w + (VW_A2 - VW_LDS_BEGIN), *sA2b = s_w + (VW_A2B - VW_LDS_BEGIN);

  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const int64_t ntiles = (N + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t g_raw = tile * 32 + i;
    const bool g_ok = g_raw < N;
    const int64_t g = g_ok ? g_raw : N - 1;
    float qq[32];
    {
      float q0[32], x[32];
      load_row32(q_in + g * 64, q0, h);
      layer_norm64(q0, W + VW_LN1_G, W + VW_LN1_B, 1e-6f, x, h);
      lin64x64(W + VW_WQ, nullptr, x, qq, i, h);
    }
    float m[32], l[32], acc[32];
    float sk[32], sk2[32], sabs[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      m[t] = -__builtin_inff();
      l[t] = 0.0f;
      acc[t] = 0.0f;
      if (STATS) {
        sk[t] = 0.0f;
        sk2[t] = 0.0f;
        sabs[t] = 0.0f;
      }
    }
    int nvalid = 0;
    for (int v = 0; v < V; ++v) {
      const int64_t row = g * V + v;
      float f[32], k[32];
      load_row32(feat + row * 64, f, h);
      lin64x64(sWk, nullptr, f, k, i, h);
      const bool ok = valid[row] != 0;
      float d2[2] = {ray_diff[row * 4 + h], ray_diff[row * 4 + 2 + h]};
      float hid[4], pos[32], a[32];
      lin4x8_relu(sP1, sP1b, d2, hid, i, h);
      lin8x64(sP2, sP2b, hid, pos, i, h);
#pragma unroll
      for (int t = 0; t < 32; ++t) a[t] = k[t] - qq[t] + pos[t];
      lin64x8_relu(sA1, sA1b, a, hid, i, h);
      if (STATS && ok) {
        ++nvalid;
#pragma unroll
        for (int t = 0; t < 32; ++t) {
          sk[t] += k[t];
          sk2[t] += k[t] * k[t];
          sabs[t] += fabsf(k[t]);
        }
      }
      // logits and values are produced 16 features (one M-tile) at a time and folded into the
      // online softmax right away, which keeps the live register set small
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        floatx16 la, lv;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          la[r] = sA2b[feat_of(r + 16 * mt, h)];
          lv[r] = 0.0f;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) la = mfma(sA2[(t + 4 * h) * 64 + mt * 32 + i], hid[t], la);
#pragma unroll
        for (int t0 = 0; t0 < 32; t0 += 8) {
          float w[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) w[u] = sWv[feat_of(t0 + u, h) * 64 + mt * 32 + i];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 8; ++u) lv = mfma(w[u], k[t0 + u], lv);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (STATS) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            *reinterpret_cast<float4 *>(logit_scratch + row * 64 + feat_of(16 * mt + c * 4, h)) =
                make_float4(la[c * 4], la[c * 4 + 1], la[c * 4 + 2], la[c * 4 + 3]);
        }
        if (ok) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int t = r + 16 * mt;
            float mn = fmaxf(m[t], la[r]);
            float sc = expf(m[t] - mn);  // exp(-inf) = 0 on the first valid view
            float e = expf(la[r] - mn);
            l[t] = l[t] * sc + e;
            acc[t] = acc[t] * sc + e * (lv[r] + pos[t]);
            m[t] = mn;
          }
        }
      }
    }
    // x = Wo (acc / l) + bo + q ;  q_out = FF(LN(x)) + x
    float x1[32];
    {
      float xa[32];
#pragma unroll
      for (int t = 0; t < 32; ++t) xa[t] = acc[t] / l[t];
      lin64x64(W + VW_WO, W + VW_WOB, xa, x1, i, h);
      float qres[32];
      load_row32(q_in + g * 64, qres, h);
#pragma unroll
      for (int t = 0; t < 32; ++t) x1[t] += qres[t];
    }
    if (g_ok) {
#pragma unroll
      for (int c = 0; c < 8; ++c)
        *reinterpret_cast<float4 *>(q_out + g * 64 + feat_of(c * 4, h)) =
            make_float4(x1[c * 4], x1[c * 4 + 1], x1[c * 4 + 2], x1[c * 4 + 3]);
    }
    if (STATS) {
      // entropy of the normalised attention (second sweep over the stored logits), masked
      // unbiased std of k and its normalised form; means over the 64 features
      float ent = 0.0f, sd = 0.0f, sdn = 0.0f;
      for (int v = 0; v < V; ++v) {
        const int64_t row = g * V + v;
        if (valid[row] == 0) continue;
        float a[32];
        load_row32(logit_scratch + row * 64, a, h);
#pragma unroll
        for (int t = 0; t < 32; ++t) {
          float p = expf(a[t] - m[t]) / l[t];
          ent += -p * logf(p + 1e-8f);
        }
      }
      if (nvalid > 1) {
        float n = (float)nvalid;
#pragma unroll
        for (int t = 0; t < 32; ++t) {
          float mean = sk[t] / n;
          float var = (sk2[t] - n * mean * mean) / (n - 1.0f);
          float s = sqrtf(fmaxf(var, 0.0f));
          sd += s;
          sdn += s / (sabs[t] / n + 1e-6f);
        }
      }
      ent += __shfl_xor(ent, 32, 64);
      sd += __shfl_xor(sd, 32, 64);
      sdn += __shfl_xor(sdn, 32, 64);
      if (g_ok && h == 0) {
        stats[g * 3 + 0] = ent * (1.0f / 64.0f);
        stats[g * 3 + 1] = sd * (1.0f / 64.0f);
        stats[g * 3 + 2] = sdn * (1.0f / 64.0f);
      }
    }
  }
}

// q_out = F2 relu(F1 LN(x) + b1) + b2 + x  (FeedForward + ff_norm + residual, :44-55,:218-221),
// in place on the rows written by the attention kernel.
__global__ void __launch_bounds__(256)
gnt_ff_kernel(const float *__restrict__ W, float *__restrict__ x_io, int64_t N) {
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const int64_t ntiles = (N + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t g_raw = tile * 32 + i;
    const bool g_ok = g_raw < N;
    const int64_t g = g_ok ? g_raw : N - 1;
    float x1[32], xn[32];
    load_row32(x_io + g * 64, x1, h);
    layer_norm64(x1, W + VW_LN2_G, W + VW_LN2_B, 1e-6f, xn, h);
    floatx16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o0[r] = W[VW_F2B + feat_of(r, h)];
      o1[r] = W[VW_F2B + feat_of(r + 16, h)];
    }
    for (int mt = 0; mt < 8; ++mt) {  // hidden features 32*mt .. 32*mt+31
      floatx16 hacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) hacc[r] = W[VW_F1B + (r & 3) + 8 * (r >> 2) + 4 * h + 32 * mt];
#pragma unroll
      for (int t0 = 0; t0 < 32; t0 += 8) {
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = W[VW_F1 + feat_of(t0 + u, h) * 256 + mt * 32 + i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) hacc = mfma(w[u], xn[t0 + u], hacc);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += 4) {
        float w0[4], w1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int r = r0 + u;
          int fin = (r & 3) + 8 * (r >> 2) + 4 * h + 32 * mt;  // hidden feature held by this lane half
          w0[u] = W[VW_F2 + fin * 64 + i];
          w1[u] = W[VW_F2 + fin * 64 + 32 + i];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float hv = fmaxf(hacc[r0 + u], 0.0f);
          o0 = mfma(w0[u], hv, o0);
          o1 = mfma(w1[u], hv, o1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (g_ok) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        *reinterpret_cast<float4 *>(x_io + g * 64 + feat_of(c * 4, h)) =
            make_float4(o0[c * 4] + x1[c * 4], o0[c * 4 + 1] + x1[c * 4 + 1], o0[c * 4 + 2] + x1[c * 4 + 2],
                        o0[c * 4 + 3] + x1[c * 4 + 3]);
        *reinterpret_cast<float4 *>(x_io + g * 64 + feat_of(16 + c * 4, h)) =
            make_float4(o1[c * 4] + x1[16 + c * 4], o1[c * 4 + 1] + x1[16 + c * 4 + 1],
                        o1[c * 4 + 2] + x1[16 + c * 4 + 2], o1[c * 4 + 3] + x1[16 + c * 4 + 3]);
      }
    }
  }
}

}  // namespace pgdvs

using namespace pgdvs;

PGDVS_API int64_t pgdvs_gnt_view_weight_floats(void) { return VW_TOTAL; }

PGDVS_API int pgdvs_gnt_view_layer(const float *weights, const float *q_in, const float *feat,
                                   const float *ray_diff, const uint8_t *valid, int64_t N, int V,
                                   float *q_out, float *stats, float *logit_scratch,
                                   pgdvs_stream_t stream) {
  PGDVS_REQUIRE(weights && q_in && feat && ray_diff && valid && q_out, "pgdvs_gnt_view_layer: null pointer");
  PGDVS_REQUIRE(N >= 0 && V >= 1, "pgdvs_gnt_view_layer: bad shape");
  PGDVS_REQUIRE((stats == nullptr) == (logit_scratch == nullptr),
                "pgdvs_gnt_view_layer: stats and logit_scratch go together");
  if (N == 0) return PGDVS_OK;
  const int64_t ntiles = cdiv(N, 32);
  const unsigned grid = (unsigned)(cdiv(ntiles, 4) < 256 ? cdiv(ntiles, 4) : 256);
  const size_t lds = (size_t)VW_LDS_FLOATS * sizeof(float);
  hipStream_t st = as_stream(stream);
  if (stats) {
    PGDVS_LAUNCH("gnt_view_layer", gnt_view_layer_kernel<true>, dim3(grid), dim3(256), lds, st, weights, q_in,
                 feat, ray_diff, valid, N, V, q_out, stats, logit_scratch);
  } else {
    PGDVS_LAUNCH("gnt_view_layer", gnt_view_layer_kernel<false>, dim3(grid), dim3(256), lds, st, weights, q_in,
                 feat, ray_diff, valid, N, V, q_out, stats, logit_scratch);
  }
  const unsigned gff = (unsigned)(cdiv(ntiles, 4) < 2048 ? cdiv(ntiles, 4) : 2048);
  PGDVS_LAUNCH("gnt_ff", gnt_ff_kernel, dim3(gff), dim3(256), 0, st, weights, q_out, N);
  return check_launch("gnt_view_layer");
}
