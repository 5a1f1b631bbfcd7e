// Shared pieces of the GNT kernels that run on the 16x16x4 fp32 MFMA (gnt_view.hip,
// gnt_embed.hip).  Tiles of 16 rows per wavefront: lane = j + 16*hq (row j of the tile,
// quarter hq), and a lane keeps, for its row, the 16 features
//   F(t,hq) = 16*(t>>2) + 4*hq + (t&3),  t = 0..15
// -- the accumulator layout of v_mfma_f32_16x16x4_f32 (row = 4*hq + r of output tile mt
// <-> t = 4*mt + r), which is also the B-operand layout of the next product (K-step (c,r)
// <-> t = 4*c + r, k index = hq): chained layers never leave registers.
#pragma once
#include "common.h"

namespace pgdvs {

// Weight pointers are loop-invariant across the persistent tile / ray loops; left alone, LICM
// hoists hundreds of 64-bit load addresses out of the loop and they end up in scratch.  Passing
// the (wave-uniform) base through an empty asm per iteration keeps the address math local.
__device__ __forceinline__ const float *opaque_uniform(const float *p) {
  asm volatile("" : "+s"(p));
  return p;
}

typedef float floatx4 __attribute__((ext_vector_type(4)));

// Weight staging global -> LDS for a workgroup of NT threads: NF4 float4's, every load issued
// before the first store (a load / wait / store loop pays the L2 latency once per iteration:
// 16 round trips for the 128 KB of the feed-forward block).  dst_of maps a float4 index to
// its float offset in LDS (row re-striding).
template <int NF4, int NT, class DstOf>
__device__ __forceinline__ void stage_f4(const float *__restrict__ src, float *__restrict__ dst, DstOf dst_of) {
  constexpr int IT = (NF4 + NT - 1) / NT;
  float4 v[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int q = (int)threadIdx.x + it * NT;
    if (NF4 % NT == 0 || q < NF4) v[it] = reinterpret_cast<const float4 *>(src)[q];
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int q = (int)threadIdx.x + it * NT;
    if (NF4 % NT == 0 || q < NF4) *reinterpret_cast<float4 *>(dst + dst_of(q)) = v[it];
  }
}


__device__ __forceinline__ floatx4 mfma16(float a, float b, floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Maxima without the canonicalising `v_max x, x` the compiler puts in front of fmaxf on values it cannot prove
// quiet (MFMA results): the median of (a, b, +inf) IS the maximum, and v_med3_f32 is one instruction (a NaN loses
// against a number, as in fmaxf).  Builtins, not inline assembly: the compiler inserts the wait states an MFMA
// result needs before a vector instruction may read it only for instructions it knows.  They matter: the vector
// ALU and the fp32 matrix pipe do not overlap on this chip (profiles/r01_mfma_calibration.txt), every vector
// instruction between MFMAs is paid in full.
__device__ __forceinline__ float vmax2(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
__device__ __forceinline__ float vrelu(float a) { return __builtin_amdgcn_fmed3f(a, 0.0f, __builtin_inff()); }

typedef float floatx2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void load_row16(const float *__restrict__ row, float (&x)[16], int hq) {
  const float *rb = row + 4 * hq;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float4 v = *reinterpret_cast<const float4 *>(rb + 16 * c);
    x[4 * c + 0] = v.x;
    x[4 * c + 1] = v.y;
    x[4 * c + 2] = v.z;
    x[4 * c + 3] = v.w;
  }
}

__device__ __forceinline__ void store_row16(float *__restrict__ row, const float (&x)[16], int hq) {
  float *rb = row + 4 * hq;
#pragma unroll
  for (int c = 0; c < 4; ++c)
    *reinterpret_cast<float4 *>(rb + 16 * c) = make_float4(x[4 * c], x[4 * c + 1], x[4 * c + 2], x[4 * c + 3]);
}

__device__ __forceinline__ float quad_sum(float s) {  // over the four lanes j, j+16, j+32, j+48
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}

// weights of K-steps 2q, 2q+1 of a 64 -> 64 product for the four output tiles; wb is the
// lane's base Wt + (4*hq)*STRIDE + i, so every address is base + immediate
template <int STRIDE>
__device__ __forceinline__ void ldq8(float (&w)[8], const float *__restrict__ wb, int q) {
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int s = 2 * q + s2;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) w[4 * s2 + mt] = wb[(16 * (s >> 2) + (s & 3)) * STRIDE + 16 * mt];
  }
}

__device__ __forceinline__ void mmq8(floatx4 (&acc)[4], const float (&w)[8], const float (&x)[16], int q) {
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma16(w[4 * s2 + mt], x[2 * q + s2], acc[mt]);
}

// acc[mt] += W x for the four 16-row output tiles of a 64 -> 64 product.  `w` holds the
// weights of K-steps 0,1 on entry; while 8 MFMAs run, the next 8 weights are on their way, and
// the last chunk covers `next(w)`, the first weights of whatever product follows.
template <int STRIDE, class Next>
__device__ __forceinline__ void chain64q(floatx4 (&acc)[4], const float *__restrict__ wb, const float (&x)[16],
                                         float (&w)[8], Next &&next) {
  float w2[8];
#pragma unroll
  for (int q = 0; q < 8; q += 2) {
    ldq8<STRIDE>(w2, wb, q + 1);
    __builtin_amdgcn_sched_barrier(0);
    mmq8(acc, w, x, q);
    __builtin_amdgcn_sched_barrier(0);
    if (q + 2 < 8)
      ldq8<STRIDE>(w, wb, q + 2);
    else
      next(w);
    __builtin_amdgcn_sched_barrier(0);
    mmq8(acc, w2, x, q + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- lane-major weight images -----------------------------------------------------------
// The A operands of a 64 -> 64 product in the order the lanes consume them: K-step s (0..15) is one float4 per
// lane -- the operands of the four output tiles -- at image[s][lane][mt].  One ds_read_b128 per K-step with an
// immediate offset off a single per-lane base (consecutive lanes, consecutive 16 bytes: no bank conflicts, no
// padding, no address arithmetic in the loop), where the row-major image takes four ds_read_b32 at four strides.
// Source: the packed input-major matrix Wt[in][out]; lane (i, hq) multiplies in = 16*(s>>2) + (s&3) + 4*hq,
// out = 16*mt + i.
template <int NT>
__device__ __forceinline__ void stage_w64_lanes(const float *__restrict__ src, float *__restrict__ dst) {
  constexpr int IT = 1024 / NT;
  float4 v[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) v[it] = reinterpret_cast<const float4 *>(src)[(int)threadIdx.x + it * NT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int q = (int)threadIdx.x + it * NT;
    const int in = q >> 4, out0 = (q & 15) * 4;
    const int s = 4 * (in >> 4) + (in & 3), hq = (in >> 2) & 3;
    float *d = dst + s * 256 + ((out0 & 15) + 16 * hq) * 4 + (out0 >> 4);
    d[0] = v[it].x;
    d[4] = v[it].y;
    d[8] = v[it].z;
    d[12] = v[it].w;
  }
}

// K-steps 2q, 2q+1 (64 -> 64 image: w[4*s2 + mt]) or, for a 64 -> 16 image [s>>2][lane][s&3], K-steps
// 8q .. 8q+7; lb = image + 4 * lane
__device__ __forceinline__ void ldq8v(float (&w)[8], const float *__restrict__ lb, int q) {
  const float4 a = *reinterpret_cast<const float4 *>(lb + (2 * q) * 256);
  const float4 b = *reinterpret_cast<const float4 *>(lb + (2 * q + 1) * 256);
  w[0] = a.x;
  w[1] = a.y;
  w[2] = a.z;
  w[3] = a.w;
  w[4] = b.x;
  w[5] = b.y;
  w[6] = b.z;
  w[7] = b.w;
}

// chain64q on a lane-major image
template <class Next>
__device__ __forceinline__ void chain64qv(floatx4 (&acc)[4], const float *__restrict__ lb, const float (&x)[16],
                                          float (&w)[8], Next &&next) {
  float w2[8];
#pragma unroll
  for (int q = 0; q < 8; q += 2) {
    ldq8v(w2, lb, q + 1);
    __builtin_amdgcn_sched_barrier(0);
    mmq8(acc, w, x, q);
    __builtin_amdgcn_sched_barrier(0);
    if (q + 2 < 8)
      ldq8v(w, lb, q + 2);
    else
      next(w);
    __builtin_amdgcn_sched_barrier(0);
    mmq8(acc, w2, x, q + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- fp32-faithful products on the bf16 matrix pipe ------------------------------------------------------------------
// v_mfma_f32_16x16x32_bf16 runs at ~16 cycles for 16 x 16 x 32 (profiles/r01_mfma_calibration.txt, tools/mfma_bf16_probe.hip)
// where the fp32 instruction takes 32 for 16 x 16 x 4: a 64 -> 64 product of a 16-row tile is 8 of them per operand piece
// pair instead of 64.  An fp32 value is the exact sum of three bf16 pieces (8 + 8 + 8 mantissa bits, by truncation:
// hi = x & 0xffff0000, the remainders are exact fp32 subtractions), so w * x = sum over the nine piece pairs; the three
// pairs below 2^-24 of the product (mid*lo, lo*mid, lo*lo) are dropped like an fp32 multiply drops its low bits, the other
// six are accumulated in fp32, small ones first: 48 MFMAs = 768 matrix cycles + ~5.5 vector instructions per activation
// value to split it, against 2048.  Weights are split once per workgroup when they are staged.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ floatx4 mfma_bf16(uintx4 a, uintx4 b, floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// eight floats -> three registers quads of eight bf16 each (element j in the low half of word j / 2 for even j)
__device__ __forceinline__ void split8_bf16x3(const float *x, uintx4 &hi, uintx4 &mid, uintx4 &lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = x[2 * j], b = x[2 * j + 1];
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    hi[j] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);  // (ub & 0xffff0000) | (ua >> 16)
    const float ra = a - __uint_as_float(ua & 0xffff0000u), rb = b - __uint_as_float(ub & 0xffff0000u);
    const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
    mid[j] = __builtin_amdgcn_perm(vb, va, 0x07060302u);
    const float sa = ra - __uint_as_float(va & 0xffff0000u), sb = rb - __uint_as_float(vb & 0xffff0000u);
    lo[j] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
  }
}

// Lane-major bf16x3 image of a 64 x 64 matrix for the products above: img[piece][mt * 2 + s][lane] (16 bytes each) holds,
// for output tile mt and K-step s, the eight weights W[in = F(8 s + j, hq)][out = 16 mt + i] of lane (i, hq) -- the in-features
// the lane's activation operand carries in the same slots (F(t, hq) = 16 (t >> 2) + 4 hq + (t & 3), t = 8 s + j).  6144 floats.
// Source: the packed input-major matrix Wt[in][out]; NT threads share the 512 slots.
template <int NT>
__device__ __forceinline__ void stage_w64_bf16x3(const float *__restrict__ src, float *__restrict__ dst) {
#pragma unroll
  for (int slot = (int)threadIdx.x; slot < 512; slot += NT) {  // 512 slots = (output tile, K-step, lane)
    const int mt = slot >> 7, s = (slot >> 6) & 1, lane = slot & 63, i = lane & 15, hq = lane >> 4;
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = 8 * s + j;
      wv[j] = src[(16 * (t >> 2) + 4 * hq + (t & 3)) * 64 + 16 * mt + i];
    }
    uintx4 h, m, l;
    split8_bf16x3(wv, h, m, l);
    uintx4 *d = reinterpret_cast<uintx4 *>(dst) + (mt * 2 + s) * 64 + lane;
    d[0] = h;
    d[8 * 64] = m;
    d[16 * 64] = l;
  }
}

// acc[mt] += W x for the four output tiles, W as a bf16x3 image (lb = image + 4 * lane floats), x the lane's 16 fp32 features.
// Per K-step three phases -- the low weight piece (one partial product per tile), the middle one (two), the high one
// (three) -- each holding its four register quads (one per output tile) while the next phase's are on their way: every
// weight quad is read once, and consecutive MFMAs go to different accumulators (the four tiles in turn; a chain of six
// into one accumulator waits for each result: 16.5 ms per chunk instead of 14.6 with the fp32 instruction).
__device__ __forceinline__ void chain64_bf16x3(floatx4 (&acc)[4], const float *__restrict__ lb, const float (&x)[16]) {
  const uintx4 *img = reinterpret_cast<const uintx4 *>(lb);
  auto quads = [&](uintx4 (&w)[4], int piece, int s) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) w[mt] = img[(piece * 8 + mt * 2 + s) * 64];
  };
  uintx4 wa[4], wb[4];
  quads(wa, 2, 0);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    uintx4 xh, xm, xl;
    split8_bf16x3(&x[8 * s], xh, xm, xl);
    quads(wb, 1, s);  // (middle pieces requested while the low ones work)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma_bf16(wa[mt], xh, acc[mt]);
    __builtin_amdgcn_sched_barrier(0);
    quads(wa, 0, s);  // (high pieces)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma_bf16(wb[mt], xm, acc[mt]);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma_bf16(wb[mt], xh, acc[mt]);
    __builtin_amdgcn_sched_barrier(0);
    if (s == 0) quads(wb, 2, 1);  // (the next K-step's low pieces)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma_bf16(wa[mt], xl, acc[mt]);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma_bf16(wa[mt], xm, acc[mt]);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma_bf16(wa[mt], xh, acc[mt]);
    __builtin_amdgcn_sched_barrier(0);
    if (s == 0) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) wa[mt] = wb[mt];
    }
  }
}

}  // namespace pgdvs
