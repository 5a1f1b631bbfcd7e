// A14 (entry of GNT.forward, pgdvs/models/gnt/models/transformer_network.py:455-474):
//   feat = rgbfeat_fc(rgb_feat)              (Linear(3+C, 64) -> ReLU -> Linear(64, 64))
//   q0   = max over the source views of feat
//   view_std_list[0] = mean_f std_v(feat),  view_std_normalized_list[0] = mean_f std / (mean_v |feat| + 1e-6)
// One pass: a wavefront owns 16 (ray,sample) groups and walks their V source views; both
// layers run on the 16x16x4 fp32 MFMA (layout: gnt_mfma.h), the feature row is written once
// and the per-group maximum / moments are carried in registers, so the [N,V,64] tensor is
// never re-read for the reductions (upstream: two GEMMs, a ReLU, a max, two std passes and a
// mean-abs over a 1.6 GB tensor at 1080p chunk sizes).
#include <type_traits>

#include "gnt_mfma.h"

namespace pgdvs {

constexpr int kEmbStride = 68;  // padded LDS rows (see gnt_view.hip: 4 * 68 = 16 mod 32 banks)
// the first layer's K-steps take input rows 4*s + hq: the two quarters of a ds_read_b32 lane group
// are ONE row apart, so its rows are padded to 80 floats (80 = 16 mod 32)
constexpr int kEmb1Stride = 80;

// packed weights (floats): W1t [4*KS in (zero padded)][64 out], b1 [64], W2t [64 in][64 out], b2 [64]
template <int KS>
__global__ void __launch_bounds__(256, 2)
gnt_embed_kernel(const float *__restrict__ W_arg, const float *__restrict__ rgb_feat, int64_t N, int V, int Cin,
                 float *__restrict__ feat, float *__restrict__ q0, float *__restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];
  float *s_w1 = s_w, *s_w2 = s_w + 4 * KS * kEmb1Stride;
  const float *b1g = W_arg + 4 * KS * 64, *w2g = b1g + 64, *b2g = w2g + 4096;
  stage_f4<4 * KS * 16, 256>(W_arg, s_w1, [](int q) { return (q >> 4) * kEmb1Stride + 4 * (q & 15); });
  stage_f4<1024, 256>(w2g, s_w2, [](int q) { return (q >> 4) * kEmbStride + 4 * (q & 15); });
  __syncthreads();
  const int lane = threadIdx.x & 63, i = lane & 15, hq = lane >> 4;
  const int wave = threadIdx.x >> 6;
  // layer 1 takes the raw row as B operand: K-step s <-> input channel 4*s + hq
  const float *w1 = s_w1 + hq * kEmb1Stride + i;
  // layer 2 takes layer 1's accumulators: K-step (c,r) <-> hidden unit 16*c + 4*hq + r
  const float *w2 = s_w2 + (4 * hq) * kEmbStride + i;
  // the biases as accumulator operands of each product's first MFMAs (register quads: no copies)
  floatx4 b1v[4], b2v[4];
  {
    float b1[16], b2[16];
    load_row16(b1g, b1, hq);
    load_row16(b2g, b2, hq);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      b1v[t >> 2][t & 3] = b1[t];
      b2v[t >> 2][t & 3] = b2[t];
    }
  }

  const int64_t ntiles = (N + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t g_raw = tile * 16 + i;
    const bool g_ok = g_raw < N;
    const int64_t g = g_ok ? g_raw : N - 1;
    // running maximum and moments over the views (all of them: the mask plays no part here); the moments are
    // taken about the first view's value, which keeps the one-pass variance as accurate as the two-pass form
    // for features whose spread is small against their mean.  Pairs of features: packed vector instructions.
    float qmax[16];
    floatx2 f0[8], s1[8], s2[8];
    float sa[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) sa[t] = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      s1[t] = floatx2{0.0f, 0.0f};
      s2[t] = floatx2{0.0f, 0.0f};
    }
    // channel 4*s + hq of the row: only the last K-step can reach past Cin (Cin > 4*KS - 4)
    const bool last_ok = 4 * (KS - 1) + hq < Cin;
    auto load_x = [&](float (&x)[KS], const float *row) {
#pragma unroll
      for (int s = 0; s + 1 < KS; ++s) x[s] = row[4 * s + hq];
      x[KS - 1] = last_ok ? row[4 * (KS - 1) + hq] : 0.0f;
    };
    float x_nx[KS];
    load_x(x_nx, rgb_feat + (g * V) * Cin);
    float w[8];
    ldq8<kEmbStride>(w, w2, 0);
    // one source view; `first` (a compile-time flag: the loop's first trip is peeled) initialises the running values
    auto view = [&](int v, auto first) {
      const int64_t r = g * V + v;
      float x[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) x[s] = x_nx[s];
      if (v + 1 < V) load_x(x_nx, rgb_feat + (r + 1) * Cin);
      // hidden = relu(W1 x + b1)
      floatx4 hacc[4];
      {
        float wa[4], wb[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) wa[mt] = w1[16 * mt];
#pragma unroll
        for (int s = 0; s < KS; s += 2) {
          if (s + 1 < KS) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) wb[mt] = w1[(4 * (s + 1)) * kEmb1Stride + 16 * mt];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) hacc[mt] = mfma16(wa[mt], x[s], s == 0 ? b1v[mt] : hacc[mt]);
          __builtin_amdgcn_sched_barrier(0);
          if (s + 2 < KS) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) wa[mt] = w1[(4 * (s + 2)) * kEmb1Stride + 16 * mt];
          }
          __builtin_amdgcn_sched_barrier(0);
          if (s + 1 < KS) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) hacc[mt] = mfma16(wb[mt], x[s + 1], hacc[mt]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      float hid[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) hid[t] = vrelu(hacc[t >> 2][t & 3]);
      // feat = W2 hidden + b2
      floatx4 o[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) o[mt] = b2v[mt];
      chain64q<kEmbStride>(o, w2, hid, w, [&](float (&d)[8]) { ldq8<kEmbStride>(d, w2, 0); });
      float f[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) f[t] = o[t >> 2][t & 3];
      if (g_ok) store_row16(feat + r * 64, f, hq);
#pragma unroll
      for (int t = 0; t < 16; ++t) qmax[t] = first ? f[t] : vmax2(qmax[t], f[t]);
      if (stats != nullptr) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const floatx2 fv = {f[2 * t], f[2 * t + 1]};
          if (first) {
            f0[t] = fv;  // d = 0: the sums stay zero
          } else {
            const floatx2 d = fv - f0[t];
            s1[t] += d;
            s2[t] = __builtin_elementwise_fma(d, d, s2[t]);
          }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) sa[t] += fabsf(f[t]);
      }
    };
    view(0, std::true_type{});
    for (int v = 1; v < V; ++v) view(v, std::false_type{});
    if (g_ok) store_row16(q0 + g * 64, qmax, hq);
    if (stats != nullptr) {
      const float n = (float)V;
      float sd = 0.0f, sdn = 0.0f;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        // torch.std (unbiased): NaN for a single view, like upstream
        const float var = (s2[t >> 1][t & 1] - s1[t >> 1][t & 1] * s1[t >> 1][t & 1] / n) / (n - 1.0f);
        const float sdev = sqrtf(fmaxf(var, 0.0f));
        sd += sdev;
        sdn += sdev / (sa[t] / n + 1e-6f);
      }
      sd = quad_sum(sd);
      sdn = quad_sum(sdn);
      if (g_ok && hq == 0) {
        stats[g * 2 + 0] = V > 1 ? sd * (1.0f / 64.0f) : __builtin_nanf("");
        stats[g * 2 + 1] = V > 1 ? sdn * (1.0f / 64.0f) : __builtin_nanf("");
      }
    }
  }
}

}  // namespace pgdvs

using namespace pgdvs;

PGDVS_API int64_t pgdvs_gnt_embed_weight_floats(int Cin) { return (int64_t)((Cin + 3) / 4) * 4 * 64 + 64 + 4096 + 64; }

PGDVS_API int pgdvs_gnt_embed(const float *weights, const float *rgb_feat, int64_t N, int V, int Cin, float *feat,
                              float *q0, float *stats, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(weights && rgb_feat && feat && q0, "pgdvs_gnt_embed: null pointer");
  PGDVS_REQUIRE(N >= 0 && V >= 1, "pgdvs_gnt_embed: bad shape");
  PGDVS_REQUIRE(Cin > 32 && Cin <= 36, "pgdvs_gnt_embed: %d input channels (built for 3 + 32 feature channels)", Cin);
  if (N == 0) return PGDVS_OK;
  constexpr int KS = 9;
  const int64_t tiles = cdiv(N, 16);
  const unsigned grid = (unsigned)(cdiv(tiles, 4) < 512 ? cdiv(tiles, 4) : 512);
  const size_t lds = (size_t)(4 * KS * kEmb1Stride + 64 * kEmbStride) * sizeof(float);
  PGDVS_LAUNCH("gnt_embed", gnt_embed_kernel<KS>, dim3(grid), dim3(256), lds, as_stream(stream), weights, rgb_feat, N,
               V, Cin, feat, q0, stats);
  return check_launch("gnt_embed");
}

// ---------------------------------------------------------------------------------------
// Positional re-embedding of the even layers (transformer_network.py:482-486):
//   q <- q_fc(cat(q, posenc(pts), posenc(viewdir))),  q_fc = Linear(64+P+P', 64) -> ReLU -> Linear(64, 64)
// The first layer is linear in the three concatenated blocks, so its position part
// T = W1[:, 64:64+P] posenc(pts) (one GEMM for all even layers at once) and its direction part
// tv = W1[:, 64+P:] posenc(viewdir) + b1 (per ray) are formed once per forward pass by the
// caller; this kernel does  q <- W2 relu(W1[:, :64] q + T + tv) + b2  per row on the MFMA,
// without materialising the concatenation.
//   weights (floats): W1q input-major [64][64], W2 input-major [64][64], b2 [64]
// ---------------------------------------------------------------------------------------
namespace pgdvs {

__global__ void __launch_bounds__(256, 2)
gnt_posfc_kernel(const float *__restrict__ W_arg, const float *__restrict__ q_in, const float *__restrict__ T,
                 int64_t t_stride, const float *__restrict__ tv, int64_t tv_stride, int64_t N, int S,
                 float *__restrict__ q_out) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];
  float *s_w1 = s_w, *s_w2 = s_w + 64 * kEmbStride;
  stage_f4<1024, 256>(W_arg, s_w1, [](int q) { return (q >> 4) * kEmbStride + 4 * (q & 15); });
  stage_f4<1024, 256>(W_arg + 4096, s_w2, [](int q) { return (q >> 4) * kEmbStride + 4 * (q & 15); });
  __syncthreads();
  const int lane = threadIdx.x & 63, i = lane & 15, hq = lane >> 4;
  const int wave = threadIdx.x >> 6;
  const float *w1 = s_w1 + (4 * hq) * kEmbStride + i, *w2 = s_w2 + (4 * hq) * kEmbStride + i;
  float b2[16];
  load_row16(W_arg + 8192, b2, hq);
  const int64_t ntiles = (N + 15) / 16;
  float w[8];
  ldq8<kEmbStride>(w, w1, 0);
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t g_raw = tile * 16 + i;
    const bool g_ok = g_raw < N;
    const int64_t g = g_ok ? g_raw : N - 1;
    float x[16], t[16], u[16];
    load_row16(q_in + g * 64, x, hq);
    load_row16(T + g * t_stride, t, hq);
    load_row16(tv + (g / S) * tv_stride, u, hq);
    floatx4 c[4];
#pragma unroll
    for (int k = 0; k < 16; ++k) c[k >> 2][k & 3] = t[k] + u[k];
    chain64q<kEmbStride>(c, w1, x, w, [&](float (&d)[8]) { ldq8<kEmbStride>(d, w2, 0); });
    float hid[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      hid[k] = fmaxf(c[k >> 2][k & 3], 0.0f);
      c[k >> 2][k & 3] = b2[k];
    }
    chain64q<kEmbStride>(c, w2, hid, w, [&](float (&d)[8]) { ldq8<kEmbStride>(d, w1, 0); });
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = c[k >> 2][k & 3];
    if (g_ok) store_row16(q_out + g * 64, x, hq);
  }
}

}  // namespace pgdvs

PGDVS_API int pgdvs_gnt_posfc(const float *weights, const float *q_in, const float *T, int64_t t_stride,
                              const float *tv, int64_t tv_stride, int64_t N, int S, float *q_out,
                              pgdvs_stream_t stream) {
  PGDVS_REQUIRE(weights && q_in && T && tv && q_out, "pgdvs_gnt_posfc: null pointer");
  PGDVS_REQUIRE(N >= 0 && S >= 1 && t_stride >= 64 && tv_stride >= 64 && t_stride % 4 == 0 && tv_stride % 4 == 0,
                "pgdvs_gnt_posfc: bad shape");
  if (N == 0) return PGDVS_OK;
  const int64_t tiles = cdiv(N, 16);
  const unsigned grid = (unsigned)(cdiv(tiles, 4) < 512 ? cdiv(tiles, 4) : 512);
  const size_t lds = (size_t)2 * 64 * kEmbStride * sizeof(float);
  PGDVS_LAUNCH("gnt_posfc", gnt_posfc_kernel, dim3(grid), dim3(256), lds, as_stream(stream), weights, q_in, T, t_stride,
               tv, tv_stride, N, S, q_out);
  return check_launch("gnt_posfc");
}

// ---------------------------------------------------------------------------------------
// Exit of GNT.forward (transformer_network.py:533-535): rgb = rgb_fc(mean_samples(LayerNorm(q))),
// LayerNorm eps 1e-5 (nn.LayerNorm default).  One workgroup per ray, one thread per sample (a
// thread owns whole rows, so the normalisation needs no cross-lane traffic); the per-thread sums
// are combined in a fixed order through LDS.
//   weights (floats): gamma[64], beta[64], rgb_fc weight [3][64], rgb_fc bias[3]
// ---------------------------------------------------------------------------------------
namespace pgdvs {

__global__ void __launch_bounds__(256)
gnt_head_kernel(const float *__restrict__ W, const float *__restrict__ q, int R, int S, float *__restrict__ out) {
  __shared__ float s_part[4][64];
  __shared__ float s_h[64];
  const int ray = blockIdx.x;
  float acc[64];
#pragma unroll
  for (int f = 0; f < 64; ++f) acc[f] = 0.0f;
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    const float4 *row = reinterpret_cast<const float4 *>(q + ((int64_t)ray * S + s) * 64);
    float x[64];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float4 v = row[c];
      x[4 * c] = v.x;
      x[4 * c + 1] = v.y;
      x[4 * c + 2] = v.z;
      x[4 * c + 3] = v.w;
    }
    float sum = 0.0f;
#pragma unroll
    for (int f = 0; f < 64; ++f) sum += x[f];
    const float mean = sum * (1.0f / 64.0f);
    float var = 0.0f;
#pragma unroll
    for (int f = 0; f < 64; ++f) var += (x[f] - mean) * (x[f] - mean);
    const float rstd = 1.0f / sqrtf(var * (1.0f / 64.0f) + 1e-5f);
#pragma unroll
    for (int f = 0; f < 64; ++f) acc[f] += (x[f] - mean) * rstd;
  }
  // wavefront sums (xor butterfly: every lane ends with the same value, fixed order), then the four wavefronts
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int f = 0; f < 64; ++f) {
    float v = acc[f];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == f) s_part[wave][f] = v;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int f = threadIdx.x;
    const float m = (s_part[0][f] + s_part[1][f] + s_part[2][f] + s_part[3][f]) / (float)S;
    s_h[f] = m * W[f] + W[64 + f];  // gamma * mean(normalised) + beta = mean(LayerNorm(q))
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float o = W[128 + 192 + threadIdx.x];
    for (int f = 0; f < 64; ++f) o += W[128 + threadIdx.x * 64 + f] * s_h[f];
    out[(int64_t)ray * 3 + threadIdx.x] = o;
  }
}

}  // namespace pgdvs

PGDVS_API int pgdvs_gnt_head(const float *weights, const float *q, int R, int S, float *rgb_out, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(weights && q && rgb_out && R >= 0 && S >= 1, "pgdvs_gnt_head: bad arguments");
  if (R == 0) return PGDVS_OK;
  PGDVS_LAUNCH("gnt_head", gnt_head_kernel, dim3((unsigned)R), dim3(256), 0, as_stream(stream), weights, q, R, S, rgb_out);
  return check_launch("gnt_head");
}
