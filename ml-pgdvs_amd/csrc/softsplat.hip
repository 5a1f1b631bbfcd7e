// A7: softmax splatting (forward warp), the gfx950 statement of kernel
// `softsplat_out` (pgdvs/utils/softsplat.py:352-402) with the wrapper's pre/post
// processing (softsplat.py:294-333) fused in, plus the renderer-level fusion
// A6+A7+A8+A11 used by PGDVSDynamicRenderer / PGDVSRenderer.
//
// One thread per SOURCE PIXEL handles all channels (the reference launches one
// thread per (channel, pixel) and re-reads the flow and re-derives the weights C
// times).  Accumulators are planar [C][H][W] so that a wavefront of neighbouring
// source pixels with coherent flow issues each atomic as one contiguous 256-byte
// row segment -- the shape the memory-side f32 atomic units run at full rate on.
// Corners whose bilinear weight is exactly zero are skipped: adding +-0 never
// changes an accumulator, so the result is identical and the zero-flow (static)
// majority costs one atomic per channel instead of four.
#include "common.h"
#include "fused.h"

namespace pgdvs {

__device__ __forceinline__ void atomic_add_f32(float *p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// fp32 adds onto LDS accumulators.  The LDS's own float atomic (ds_add_f32) takes ~80 ns per wavefront instruction and CU
// on gfx950 -- forty times an integer LDS atomic (tools/lds_atomic_rate.hip, profiles/r03_lds_atomic_rate.txt) -- so the
// add is a compare-and-swap on the bit pattern: the kN accumulators of one lane are distinct, so they are read together,
// swapped together, and only a lane that lost a race (another lane of the tile hit the same accumulator in between) goes
// round again.  ~6 ns per accumulator and wavefront.
template <int kN>
__device__ __forceinline__ void lds_add_f32(uint32_t *const (&cell)[kN], const float (&add)[kN], const bool (&on)[kN]) {
  uint32_t seen[kN], want[kN];
#pragma unroll
  for (int i = 0; i < kN; ++i)
    if (on[i]) seen[i] = __hip_atomic_load(cell[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  bool again = false;
#pragma unroll
  for (int i = 0; i < kN; ++i)
    if (on[i]) {
      want[i] = seen[i];
      seen[i] = atomicCAS(cell[i], want[i], __float_as_uint(__uint_as_float(want[i]) + add[i]));
    }
#pragma unroll
  for (int i = 0; i < kN; ++i) again = again || (on[i] && seen[i] != want[i]);
  if (again) {
#pragma unroll
    for (int i = 0; i < kN; ++i)
      if (on[i])
        while (seen[i] != want[i]) {
          want[i] = seen[i];
          seen[i] = atomicCAS(cell[i], want[i], __float_as_uint(__uint_as_float(want[i]) + add[i]));
        }
  }
}

struct SplatCorners {
  int idx[4];   // flat destination index (y*W+x) or -1; order nw, ne, sw, se
  float w[4];
  int x0, y0;   // target pixel of the nw corner (may lie outside the image)
  bool any;
};

// corner indices + weights exactly as softsplat.py:365-401
__device__ __forceinline__ SplatCorners splat_corners(int x, int y, float fx, float fy, int H,
                                                      int W) {
  SplatCorners c;
  c.any = false;
  c.x0 = 0;
  c.y0 = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) c.idx[k] = -1;
  float X = (float)x + fx;
  float Y = (float)y + fy;
  if (!isfinite(X) || !isfinite(Y)) return c;
  float flx = floorf(X), fly = floorf(Y);
  // (int) floor() of a value outside [-2, size] can only fail the bounds tests
  if (flx < -2.0f || flx > (float)W || fly < -2.0f || fly > (float)H) return c;
  int nwx = (int)flx, nwy = (int)fly;
  c.x0 = nwx;
  c.y0 = nwy;
  int nex = nwx + 1, ney = nwy;
  int swx = nwx, swy = nwy + 1;
  int sex = nwx + 1, sey = nwy + 1;
  c.w[0] = ((float)sex - X) * ((float)sey - Y);
  c.w[1] = (X - (float)swx) * ((float)swy - Y);
  c.w[2] = ((float)nex - X) * (Y - (float)ney);
  c.w[3] = (X - (float)nwx) * (Y - (float)nwy);
  if (nwx >= 0 && nwx < W && nwy >= 0 && nwy < H) c.idx[0] = nwy * W + nwx;
  if (nex >= 0 && nex < W && ney >= 0 && ney < H) c.idx[1] = ney * W + nex;
  if (swx >= 0 && swx < W && swy >= 0 && swy < H) c.idx[2] = swy * W + swx;
  if (sex >= 0 && sex < W && sey >= 0 && sey < H) c.idx[3] = sey * W + sex;
  c.any = c.idx[0] >= 0 || c.idx[1] >= 0 || c.idx[2] >= 0 || c.idx[3] >= 0;
  return c;
}

__device__ __forceinline__ void splat_value(float *plane, const SplatCorners &c, float v) {
  const bool fin = isfinite(v);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (c.idx[k] >= 0 && !(c.w[k] == 0.0f && fin)) atomic_add_f32(plane + c.idx[k], v * c.w[k]);
  }
}

// generic op ---------------------------------------------------------------
// mode: 0 sum, 1 avg, 2 linear, 3 soft.  acc has Cacc = C (+1 if mode != 0) planes.
template <int MODE>
__global__ void __launch_bounds__(256)
softsplat_scatter_kernel(const float *__restrict__ in, const float *__restrict__ flow,
                         const float *__restrict__ metric, float *__restrict__ acc, int C, int H,
                         int W) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  int P = H * W;
  if (p >= P) return;
  int b = blockIdx.y;
  int y = p / W, x = p - y * W;
  const float *fl = flow + (size_t)b * 2 * P;
  SplatCorners c = splat_corners(x, y, fl[p], fl[(size_t)P + p], H, W);
  if (!c.any) return;
  const int Cacc = MODE == 0 ? C : C + 1;
  float *a = acc + (size_t)b * Cacc * P;
  const float *src = in + (size_t)b * C * P;
  float m = 1.0f;
  if (MODE == 2) m = metric[(size_t)b * P + p];
  if (MODE == 3) m = expf(metric[(size_t)b * P + p]);
  for (int ch = 0; ch < C; ++ch) {
    float v = src[(size_t)ch * P + p];
    if (MODE >= 2) v = v * m;
    splat_value(a + (size_t)ch * P, c, v);
  }
  if (MODE != 0) splat_value(a + (size_t)C * P, c, m);
}

// out[:, :C] = acc[:, :C] / norm(acc[:, C]) -- softsplat.py:313-330
__global__ void __launch_bounds__(256)
softsplat_normalize_kernel(const float *__restrict__ acc, float *__restrict__ out, int C, int H,
                           int W, int eps) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  int P = H * W;
  if (p >= P) return;
  int b = blockIdx.y;
  const float *a = acc + (size_t)b * (C + 1) * P;
  float nrm = a[(size_t)C * P + p];
  if (eps == 0) nrm = nrm + 0.0000001f;
  else if (eps == 1) nrm = (nrm == 0.0f) ? 1.0f : nrm;
  else nrm = nrm < 0.0000001f ? 0.0000001f : nrm;   // clip(min): NaN stays NaN
  float *o = out + (size_t)b * C * P;
  for (int ch = 0; ch < C; ++ch) o[(size_t)ch * P + p] = a[(size_t)ch * P + p] / nrm;
}

// renderer-level fusion -------------------------------------------------------
// torch.linspace(-1,1,steps)[i] as ATen computes it
__device__ __forceinline__ float linspace_pm1(int i, int steps) {
  if (steps == 1) return -1.0f;
  float step = (1.0f - (-1.0f)) / (float)(steps - 1);
  if (i < steps / 2) return -1.0f + step * (float)i;
  return 1.0f - step * (float)(steps - 1 - i);
}

// Which target pixels can receive dynamic content at all: the corners of every source pixel with
// a non-zero mask.  The rendered colour and mask are multiplied by (splatted mask / norm > 1e-3)
// afterwards (pgdvs_renderer_dyn.py:200-202), so whatever the ~85 % static source pixels add to
// a target pixel outside this set is multiplied by zero: the scatter below skips a static source
// pixel none of whose corners is flagged, before its backwarp, exponential and 16 atomics (the
// L2 atomic rate, ~270 G/s, is what bounds the scatter).  Identical results for finite inputs (a
// non-finite static colour no longer turns an untouched target pixel into NaN).
// The same set is the only place where the accumulators are ever READ (the finish kernel writes the
// static composite elsewhere without looking at them), so only its pixels are zeroed -- here, by whoever
// flags them (idempotent) -- instead of 20 bytes per pixel of memset; what the scatter adds onto the
// uninitialised accumulators of unflagged pixels is never looked at.
__global__ void __launch_bounds__(256)
dyn_splat_flag_kernel(int H, int W, const float *__restrict__ flow_1_to_tgt,
                      const float *__restrict__ valid_mask, uint8_t *__restrict__ flags, float *__restrict__ acc) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int P = H * W;
  if (p >= P) return;
  if (valid_mask[p] == 0.0f) return;
  int y = p / W, x = p - y * W;
  SplatCorners c = splat_corners(x, y, flow_1_to_tgt[p], flow_1_to_tgt[(size_t)P + p], H, W);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (c.idx[k] >= 0) {
      flags[c.idx[k]] = 1;
#pragma unroll
      for (int pl = 0; pl < 5; ++pl) acc[(size_t)pl * P + c.idx[k]] = 0.0f;
    }
}

// The per-view call's variant (round 6): A5's projection of the kept points into the target view
// (project_flow_dense_kernel, dyn.hip: flow = projection - pixel, planar, zeros and mask 0 where nothing is kept) inside the
// flag pass -- the same operations on the same values, one launch and one read of the flow planes less.
__global__ void __launch_bounds__(256)
dyn_project_flag_kernel(int H, int W, const float *__restrict__ cam_tgt, const float *__restrict__ pcl,
                        const uint8_t *__restrict__ keep, float *__restrict__ flow_1_to_tgt, float *__restrict__ valid_mask,
                        uint8_t *__restrict__ flags, float *__restrict__ acc) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int P = H * W;
  if (p >= P) return;
  float fx = 0.0f, fy = 0.0f, vm = 0.0f;
  int y = p / W, x = p - y * W;
  const bool kept = keep[p] != 0;
  if (kept) {
    float u, v;
    project_point(cam_tgt + PGDVS_CAM_P, pcl[(size_t)p * 3], pcl[(size_t)p * 3 + 1], pcl[(size_t)p * 3 + 2], u, v);
    fx = u - (float)x;
    fy = v - (float)y;
    vm = 1.0f;
  }
  if (flow_1_to_tgt != nullptr) {  // (null: the scatter pass projects the kept points itself -- see dyn_splat_scatter_kernel<true>)
    flow_1_to_tgt[p] = fx;
    flow_1_to_tgt[(size_t)P + p] = fy;
    valid_mask[p] = vm;
  }
  if (!kept) return;
  SplatCorners c = splat_corners(x, y, fx, fy, H, W);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (c.idx[k] >= 0) {
      flags[c.idx[k]] = 1;
#pragma unroll
      for (int pl = 0; pl < 5; ++pl) acc[(size_t)pl * P + c.idx[k]] = 0.0f;
    }
}

// acc planes: 0..2 rgb*e, 3 e, 4 mask*e
// One workgroup per 16x16 tile of source pixels.  The scatter is bound by the L2 atomic rate,
// and neighbouring source pixels land on the same target pixels (every target pixel of a smooth
// flow field collects ~4 bilinear contributions per plane): the tile's contributions are first
// summed in an LDS window anchored at the tile's smallest target coordinates (lds_add_f32 above),
// then each touched target pixel receives ONE global atomic per plane.  Corners that fall outside
// the window (flows that diverge by more than kSplatWin - 17 pixels inside a tile) go to global
// memory directly.
constexpr int kSplatTile = 16, kSplatWin = 40;

// torch.randn_like(rgb_src_1) of the reference (pgdvs_renderer_dyn.py:177-182), drawn where it is consumed: the
// noise only shows on static source pixels that splat next to dynamic content (a few per cent of the frame), so
// a Philox4x32-10 block per such pixel (key = the caller's seed, counter = (draw number, pixel)) replaces a
// 25 MB normal field written by one kernel and read back by this one.  Three of the block's four uniforms pairs:
// Box-Muller, channels 0 / 1 from the first pair, channel 2 from the second.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}
__device__ __forceinline__ void splat_noise3(const unsigned long long *__restrict__ rng, int p, float (&nz)[3]) {
  const unsigned long long seed = rng[0], draw = rng[1];
  const uint4 r = philox4x32_10(make_uint4((uint32_t)p, 0u, (uint32_t)draw, (uint32_t)(draw >> 32)),
                                make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  const float u0 = ((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = ((float)(r.y >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(r.z >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = ((float)(r.w >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
  nz[0] = ra * cosf(6.283185307179586f * u1);
  nz[1] = ra * sinf(6.283185307179586f * u1);
  nz[2] = rb * cosf(6.283185307179586f * u3);
}
// the field dyn_splat_scatter_kernel draws for the state `rng` (tests; [3,H,W], un-clamped like torch.randn)
__global__ void __launch_bounds__(256) splat_noise_field_kernel(int P, const unsigned long long *__restrict__ rng,
                                                                float *__restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float nz[3];
  splat_noise3(rng, p, nz);
#pragma unroll
  for (int k = 0; k < 3; ++k) out[(size_t)k * P + p] = nz[k];
}

// kProject (the per-view call, round 6): the flow to the target and the validity of a source pixel are not read from the dense
// planes A5 wrote (8 + 4 bytes per pixel written by one launch and read back by this one: 48 MB per view at 1080p) but worked
// out here -- keep[p] ? projection of pcl[p] through project_point, the same operations on the same values as
// project_flow_dense_kernel : zero flow, mask 0 -- for the 15 % of the pixels that are kept; the others cost one byte.
template <bool kProject>
__global__ void __launch_bounds__(256)
dyn_splat_scatter_kernel(int H, int W, const float *__restrict__ rgb1,
                         const float *__restrict__ rgb2, const float *__restrict__ flow12,
                         const float *__restrict__ flow_1_to_tgt,
                         const float *__restrict__ valid_mask, const float *__restrict__ noise,
                         const unsigned long long *__restrict__ rng, float alpha, float *__restrict__ acc,
                         const uint8_t *__restrict__ flags, const float *__restrict__ cam_tgt,
                         const float *__restrict__ pcl, const uint8_t *__restrict__ keep) {
  __shared__ float s_acc[5][kSplatWin * kSplatWin];
  __shared__ int s_org[2];
  const int P = H * W;
  const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int x = blockIdx.x * kSplatTile + lx, y = blockIdx.y * kSplatTile + ly;
  const int p = y * W + x;
  if (threadIdx.x == 0) {
    s_org[0] = 0x7fffffff;
    s_org[1] = 0x7fffffff;
  }
  SplatCorners c;
  c.any = false;
  float m = 0.0f;
  bool part = false;
  if (x < W && y < H) {
    float fx, fy, vm;
    if (kProject) {
      fx = 0.0f;
      fy = 0.0f;
      vm = 0.0f;
      if (keep[p]) {
        float u, v;
        project_point(cam_tgt + PGDVS_CAM_P, pcl[(size_t)p * 3], pcl[(size_t)p * 3 + 1], pcl[(size_t)p * 3 + 2], u, v);
        fx = u - (float)x;
        fy = v - (float)y;
        vm = 1.0f;
      }
    } else {
      fx = flow_1_to_tgt[p];
      fy = flow_1_to_tgt[(size_t)P + p];
      vm = valid_mask[p];
    }
    c = splat_corners(x, y, fx, fy, H, W);
    if (c.any) {
      m = vm;
      part = true;
      if (m == 0.0f) {
        bool wanted = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) wanted = wanted || (c.idx[k] >= 0 && flags[c.idx[k]] != 0);
        part = wanted;
      }
    }
  }
  if (!__syncthreads_or(part ? 1 : 0)) return;  // nothing of this tile can matter
  {  // window origin: the smallest in-image target coordinates of the tile (one LDS atomic per wavefront, not per lane:
     // 64 lanes on one LDS word are served one after the other)
    int mx = part ? (c.x0 < 0 ? 0 : c.x0) : 0x7fffffff, my = part ? (c.y0 < 0 ? 0 : c.y0) : 0x7fffffff;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      mx = min(mx, __shfl_xor(mx, d));
      my = min(my, __shfl_xor(my, d));
    }
    if ((threadIdx.x & 63) == 0 && mx != 0x7fffffff) {
      atomicMin(&s_org[0], mx);
      atomicMin(&s_org[1], my);
    }
  }
  for (int i = threadIdx.x; i < 5 * kSplatWin * kSplatWin / 4; i += 256)
    reinterpret_cast<float4 *>(&s_acc[0][0])[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const int ox = s_org[0], oy = s_org[1];
  if (part) {
    // rgb_src_1 = rgb*mask + clamp(randn,0,1)*(1-mask)   (pgdvs_renderer_dyn.py:177-182)
    float c1[3], drawn[3] = {0.0f, 0.0f, 0.0f};
    if (noise == nullptr && rng != nullptr && m != 1.0f) splat_noise3(rng, p, drawn);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float nz = clampf(noise ? noise[(size_t)k * P + p] : drawn[k], 0.0f, 1.0f);
      c1[k] = rgb1[(size_t)p * 3 + k] * m + nz * (1.0f - m);
    }
    // backwarp rgb2 by flow12, align_corners=True (pgdvs_renderer_base.py:91-138)
    float2 f12 = reinterpret_cast<const float2 *>(flow12)[p];
    const float hw_x = ((float)W - 1.0f) / 2.0f, hw_y = ((float)H - 1.0f) / 2.0f;
    float gx = linspace_pm1(x, W) + f12.x / hw_x;
    float gy = linspace_pm1(y, H) + f12.y / hw_y;
    float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
    float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    bool fin = isfinite(ix) && isfinite(iy) && fabsf(ix) < 1e9f && fabsf(iy) < 1e9f;
    float x0f = floorf(ix), y0f = floorf(iy);
    int x0 = fin ? (int)x0f : -10, y0 = fin ? (int)y0f : -10, x1 = x0 + 1, y1 = y0 + 1;
    float wnw = ((float)x1 - ix) * ((float)y1 - iy);
    float wne = (ix - (float)x0) * ((float)y1 - iy);
    float wsw = ((float)x1 - ix) * (iy - (float)y0);
    float wse = (ix - (float)x0) * (iy - (float)y0);
    bool inx0 = x0 >= 0 && x0 < W, inx1 = x1 >= 0 && x1 < W;
    bool iny0 = y0 >= 0 && y0 < H, iny1 = y1 >= 0 && y1 < H;
    float sabs = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float a = 0.0f;
      if (inx0 && iny0) a = a + rgb2[((size_t)y0 * W + x0) * 3 + k] * wnw;
      if (inx1 && iny0) a = a + rgb2[((size_t)y0 * W + x1) * 3 + k] * wne;
      if (inx0 && iny1) a = a + rgb2[((size_t)y1 * W + x0) * 3 + k] * wsw;
      if (inx1 && iny1) a = a + rgb2[((size_t)y1 * W + x1) * 3 + k] * wse;
      sabs = sabs + fabsf(c1[k] - a);
    }
    float l1 = sabs / 3.0f;
    // metric = clip(-alpha*L1, -alpha, alpha); soft mode weight exp(metric)
    float e = expf(clampf(-alpha * l1, -alpha, alpha));
    const float me = m * e;
    const float val[5] = {c1[0] * e, c1[1] * e, c1[2] * e, e, me};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (c.idx[k] < 0) continue;
      const int wx = c.x0 + (k & 1) - ox, wy = c.y0 + (k >> 1) - oy;
      const bool in_win = wx >= 0 && wx < kSplatWin && wy >= 0 && wy < kSplatWin;
      uint32_t *cell[5];
      float add[5];
      bool on[5];
#pragma unroll
      for (int pl = 0; pl < 5; ++pl) {
        const float v = val[pl];
        // (the mask plane of a static pixel; adding +-0 changes nothing -- splat_value's rule)
        const bool skip = (pl == 4 && me == 0.0f) || (c.w[k] == 0.0f && isfinite(v));
        add[pl] = v * c.w[k];
        on[pl] = in_win && !skip;
        cell[pl] = reinterpret_cast<uint32_t *>(&s_acc[pl][in_win ? wy * kSplatWin + wx : 0]);
        if (!in_win && !skip) atomic_add_f32(acc + (size_t)pl * P + c.idx[k], add[pl]);
      }
      lds_add_f32(cell, add, on);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kSplatWin * kSplatWin; i += 256) {
    const int ty = oy + i / kSplatWin, tx = ox + i % kSplatWin;
    if (tx >= W || ty >= H) continue;
#pragma unroll
    for (int pl = 0; pl < 5; ++pl) {
      const float v = s_acc[pl][i];
      if (v != 0.0f) atomic_add_f32(acc + (size_t)pl * P + (size_t)ty * W + tx, v);
    }
  }
}

// normalise, threshold, mask and composite (pgdvs_renderer_dyn.py:200-202,
// pgdvs_renderer.py:169-178)
__global__ void __launch_bounds__(256)
dyn_splat_finish_kernel(int P, const float *__restrict__ acc, const uint8_t *__restrict__ flags,
                        const float *__restrict__ static_rgb,
                        float *__restrict__ dyn_rgb, float *__restrict__ dyn_mask,
                        float *__restrict__ comb, float *__restrict__ comb_st,
                        float *__restrict__ comb_dy, unsigned long long *__restrict__ rng) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  if (p == 0 && rng != nullptr) rng[1] = rng[1] + 1ull;  // the next call draws a fresh field (the scatter launch is done)
  // a pixel no dynamic source pixel reaches has splatted mask 0 -> below the 1e-3 threshold -> (.) * 0:
  // its accumulators (never initialised) are not read
  const bool live = flags[p] != 0;
  float nrm = live ? acc[(size_t)3 * P + p] + 0.0000001f : 1.0f;
  float mk = live ? acc[(size_t)4 * P + p] / nrm : 0.0f;
  float dm = mk > 1e-3f ? 1.0f : 0.0f;
  dyn_mask[p] = dm;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float v = live ? (acc[(size_t)k * P + p] / nrm) * dm : 0.0f;
    dyn_rgb[(size_t)k * P + p] = v;
    if (static_rgb) {
      float a = (1.0f - dm) * static_rgb[(size_t)k * P + p];
      float b = dm * v;
      if (comb_st) comb_st[(size_t)k * P + p] = a;
      if (comb_dy) comb_dy[(size_t)k * P + p] = b;
      if (comb) comb[(size_t)k * P + p] = a + b;
    }
  }
}

// Backward of the raw splat (softsplat.py:459-617): one thread per source pixel computes the
// four corner weights and their flow derivatives once and walks the channels, producing the
// input gradient of every channel and both flow gradients from a single read of outgrad (the
// reference launches softsplat_ingrad over B*C*H*W and softsplat_flowgrad over B*2*H*W threads,
// the latter re-reading every channel).  A gather: no atomics, deterministic.
__global__ void __launch_bounds__(256)
softsplat_bwd_kernel(const float *__restrict__ in, const float *__restrict__ flow,
                     const float *__restrict__ outgrad, float *__restrict__ ingrad,
                     float *__restrict__ flowgrad, int C, int H, int W) {
  const int64_t P = (int64_t)H * W;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int n = blockIdx.y;
  const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
  const float X = (float)x + flow[((int64_t)n * 2 + 0) * P + p];
  const float Y = (float)y + flow[((int64_t)n * 2 + 1) * P + p];
  bool ok = isfinite(X) && isfinite(Y);
  const float flx = floorf(X), fly = floorf(Y);
  if (ok && (flx < -2.0f || flx > (float)W || fly < -2.0f || fly > (float)H)) ok = false;
  const int nwx = ok ? (int)flx : -8, nwy = ok ? (int)fly : -8;
  const int cx[4] = {nwx, nwx + 1, nwx, nwx + 1}, cy[4] = {nwy, nwy, nwy + 1, nwy + 1};
  const float sex = (float)(nwx + 1), sey = (float)(nwy + 1), wx = (float)nwx, wy = (float)nwy;
  const float w[4] = {(sex - X) * (sey - Y), (X - wx) * (sey - Y), (sex - X) * (Y - wy), (X - wx) * (Y - wy)};
  const float dx[4] = {-1.0f * (sey - Y), +1.0f * (sey - Y), -1.0f * (Y - wy), +1.0f * (Y - wy)};
  const float dy[4] = {(sex - X) * -1.0f, (X - wx) * -1.0f, (sex - X) * +1.0f, (X - wx) * +1.0f};
  bool inb[4];
  int64_t off[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    inb[k] = ok && cx[k] >= 0 && cx[k] < W && cy[k] >= 0 && cy[k] < H;
    off[k] = inb[k] ? (int64_t)cy[k] * W + cx[k] : 0;
  }
  float gfx = 0.0f, gfy = 0.0f;
  for (int ch = 0; ch < C; ++ch) {
    const float *og = outgrad + ((int64_t)n * C + ch) * P;
    const float v = in[((int64_t)n * C + ch) * P + p];
    float gi = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!inb[k]) continue;
      const float g = og[off[k]];
      gi = gi + g * w[k];
      gfx = gfx + g * v * dx[k];
      gfy = gfy + g * v * dy[k];
    }
    if (ingrad) ingrad[((int64_t)n * C + ch) * P + p] = gi;
  }
  if (flowgrad) {
    flowgrad[((int64_t)n * 2 + 0) * P + p] = gfx;
    flowgrad[((int64_t)n * 2 + 1) * P + p] = gfy;
  }
}

}  // namespace pgdvs


using namespace pgdvs;

PGDVS_API int64_t pgdvs_softsplat_workspace_bytes(int B, int C, int H, int W, int mode) {
  if (mode == 0) return 256;
  return align_up((int64_t)B * (C + 1) * H * W * (int64_t)sizeof(float), 256);
}

PGDVS_API int pgdvs_softsplat_fwd(const float *in, const float *flow, const float *metric,
                                  float *out, int B, int C, int H, int W, int mode, int eps,
                                  void *workspace, int64_t workspace_bytes, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(in && flow && out, "pgdvs_softsplat_fwd: null pointer");
  PGDVS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 31) && B < 65536,
                "pgdvs_softsplat_fwd: bad shape");
  PGDVS_REQUIRE(mode >= 0 && mode <= 3 && eps >= 0 && eps <= 2, "pgdvs_softsplat_fwd: bad mode");
  // softsplat.py:285-292
  PGDVS_REQUIRE((mode >= 2) == (metric != nullptr),
                "pgdvs_softsplat_fwd: metric must be given for linear/soft and absent for sum/avg");
  hipStream_t st = as_stream(stream);
  const int64_t P = (int64_t)H * W;
  dim3 grid((unsigned)cdiv(P, 256), B), block(256);
  hipError_t e;
  if (mode == 0) {
    e = fill_async(out, 0, (size_t)B * C * P * sizeof(float), st);
    if (e != hipSuccess) {
      set_error("softsplat memset: %s", hipGetErrorString(e));
      return PGDVS_ERR_LAUNCH;
    }
    PGDVS_LAUNCH("softsplat_scatter", softsplat_scatter_kernel<0>, grid, block, 0, st, in, flow, metric, out, C, H, W);
    return check_launch("softsplat_scatter");
  }
  if (!workspace || workspace_bytes < pgdvs_softsplat_workspace_bytes(B, C, H, W, mode)) {
    set_error("pgdvs_softsplat_fwd: workspace too small");
    return PGDVS_ERR_WORKSPACE;
  }
  float *acc = reinterpret_cast<float *>(workspace);
  e = fill_async(acc, 0, (size_t)B * (C + 1) * P * sizeof(float), st);
  if (e != hipSuccess) {
    set_error("softsplat memset: %s", hipGetErrorString(e));
    return PGDVS_ERR_LAUNCH;
  }
  if (mode == 1)
    PGDVS_LAUNCH("softsplat_scatter", softsplat_scatter_kernel<1>, grid, block, 0, st, in, flow, metric, acc, C, H, W);
  else if (mode == 2)
    PGDVS_LAUNCH("softsplat_scatter", softsplat_scatter_kernel<2>, grid, block, 0, st, in, flow, metric, acc, C, H, W);
  else
    PGDVS_LAUNCH("softsplat_scatter", softsplat_scatter_kernel<3>, grid, block, 0, st, in, flow, metric, acc, C, H, W);
  PGDVS_LAUNCH("softsplat_normalize", softsplat_normalize_kernel, grid, block, 0, st, acc, out, C, H, W, eps);
  return check_launch("softsplat_fwd");
}

PGDVS_API int64_t pgdvs_dyn_splat_workspace_bytes(int H, int W) {
  // five accumulator planes + one flag byte per target pixel
  return align_up((int64_t)5 * H * W * (int64_t)sizeof(float) + (int64_t)H * W, 256);
}

static int dyn_splat_composite_impl(int H, int W, const float *rgb1, const float *rgb2, const float *flow12,
                                    const float *flow_1_to_tgt, const float *valid_dyn_mask_1, const float *noise,
                                    unsigned long long *rng, float alpha, const float *static_rgb, float *render_dyn_rgb,
                                    float *render_dyn_mask, float *combined, float *combined_static, float *combined_dyn,
                                    void *workspace, int64_t workspace_bytes, pgdvs_stream_t stream);
namespace pgdvs {
int dyn_splat_scatter_part(int H, int W, const float *rgb1, const float *rgb2, const float *flow12, const float *flow_1_to_tgt,
                           const float *valid_dyn_mask_1, const float *noise, const unsigned long long *rng, float alpha,
                           void *workspace, hipStream_t st);
int dyn_splat_finish_part(int H, int W, unsigned long long *rng, const float *static_rgb, float *render_dyn_rgb,
                          float *render_dyn_mask, float *combined, float *combined_static, float *combined_dyn, void *workspace,
                          hipStream_t st);
}  // namespace pgdvs

PGDVS_API int pgdvs_dyn_splat_composite(int H, int W, const float *rgb1, const float *rgb2,
                                        const float *flow12, const float *flow_1_to_tgt,
                                        const float *valid_dyn_mask_1, const float *noise,
                                        float alpha, const float *static_rgb, float *render_dyn_rgb,
                                        float *render_dyn_mask, float *combined,
                                        float *combined_static, float *combined_dyn,
                                        void *workspace, int64_t workspace_bytes,
                                        pgdvs_stream_t stream) {
  return dyn_splat_composite_impl(H, W, rgb1, rgb2, flow12, flow_1_to_tgt, valid_dyn_mask_1, noise, nullptr, alpha, static_rgb,
                                  render_dyn_rgb, render_dyn_mask, combined, combined_static, combined_dyn, workspace,
                                  workspace_bytes, stream);
}

PGDVS_API int pgdvs_dyn_splat_composite_rng(int H, int W, const float *rgb1, const float *rgb2, const float *flow12,
                                            const float *flow_1_to_tgt, const float *valid_dyn_mask_1,
                                            uint64_t *rng_state, float alpha, const float *static_rgb,
                                            float *render_dyn_rgb, float *render_dyn_mask, float *combined,
                                            float *combined_static, float *combined_dyn, void *workspace,
                                            int64_t workspace_bytes, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(rng_state, "pgdvs_dyn_splat_composite_rng: null rng_state");
  return dyn_splat_composite_impl(H, W, rgb1, rgb2, flow12, flow_1_to_tgt, valid_dyn_mask_1, nullptr,
                                  reinterpret_cast<unsigned long long *>(rng_state), alpha, static_rgb, render_dyn_rgb,
                                  render_dyn_mask, combined, combined_static, combined_dyn, workspace, workspace_bytes, stream);
}

PGDVS_API int pgdvs_splat_noise_field(int H, int W, const uint64_t *rng_state, float *noise_out, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(H > 0 && W > 0 && (int64_t)H * W < (1ll << 31) && rng_state && noise_out, "pgdvs_splat_noise_field: bad argument");
  const int P = H * W;
  PGDVS_LAUNCH("splat_noise_field", splat_noise_field_kernel, dim3((unsigned)cdiv(P, 256)), dim3(256), 0, as_stream(stream), P,
               reinterpret_cast<const unsigned long long *>(rng_state), noise_out);
  return check_launch("splat_noise_field");
}

static int dyn_splat_composite_impl(int H, int W, const float *rgb1, const float *rgb2, const float *flow12,
                                    const float *flow_1_to_tgt, const float *valid_dyn_mask_1, const float *noise,
                                    unsigned long long *rng, float alpha, const float *static_rgb, float *render_dyn_rgb,
                                    float *render_dyn_mask, float *combined, float *combined_static, float *combined_dyn,
                                    void *workspace, int64_t workspace_bytes, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(H > 0 && W > 0 && (int64_t)H * W < (1ll << 31), "pgdvs_dyn_splat_composite: bad H/W");
  PGDVS_REQUIRE(rgb1 && rgb2 && flow12 && flow_1_to_tgt && valid_dyn_mask_1 && render_dyn_rgb &&
                    render_dyn_mask,
                "pgdvs_dyn_splat_composite: null pointer");
  PGDVS_REQUIRE(alpha >= 0.0f, "pgdvs_dyn_splat_composite: alpha must be >= 0");
  if (!workspace || workspace_bytes < pgdvs_dyn_splat_workspace_bytes(H, W)) {
    set_error("pgdvs_dyn_splat_composite: workspace too small");
    return PGDVS_ERR_WORKSPACE;
  }
  int rc = pgdvs::dyn_splat_scatter_part(H, W, rgb1, rgb2, flow12, flow_1_to_tgt, valid_dyn_mask_1, noise, rng, alpha, workspace,
                                         as_stream(stream));
  if (rc != PGDVS_OK) return rc;
  return pgdvs::dyn_splat_finish_part(H, W, rng, static_rgb, render_dyn_rgb, render_dyn_mask, combined, combined_static,
                                      combined_dyn, workspace, as_stream(stream));
}

namespace pgdvs {
// The two halves of the composite (the per-view call runs the first beside the static branch: it needs the flows, not the
// static image).  Arguments as validated by dyn_splat_composite_impl / pgdvs_view_geo_forward; workspace >=
// pgdvs_dyn_splat_workspace_bytes(H, W).
int dyn_splat_scatter_part(int H, int W, const float *rgb1, const float *rgb2, const float *flow12, const float *flow_1_to_tgt,
                           const float *valid_dyn_mask_1, const float *noise, const unsigned long long *rng, float alpha,
                           void *workspace, hipStream_t st) {
  const int P = H * W;
  float *acc = reinterpret_cast<float *>(workspace);
  uint8_t *flags = reinterpret_cast<uint8_t *>(acc + (size_t)5 * P);
  hipError_t e = fill_async(flags, 0, (size_t)P, st);  // (the accumulators are zeroed where they will be read)
  if (e != hipSuccess) {
    set_error("dyn_splat memset: %s", hipGetErrorString(e));
    return PGDVS_ERR_LAUNCH;
  }
  dim3 grid((unsigned)cdiv(P, 256)), block(256);
  PGDVS_LAUNCH("dyn_splat_flag", dyn_splat_flag_kernel, grid, block, 0, st, H, W, flow_1_to_tgt, valid_dyn_mask_1,
               flags, acc);
  const dim3 tgrid((unsigned)cdiv(W, kSplatTile), (unsigned)cdiv(H, kSplatTile));
  PGDVS_LAUNCH("dyn_splat_scatter", dyn_splat_scatter_kernel<false>, tgrid, block, 0, st, H, W, rgb1, rgb2, flow12,
                     flow_1_to_tgt, valid_dyn_mask_1, noise, rng, alpha, acc, (const uint8_t *)flags, (const float *)nullptr,
                     (const float *)nullptr, (const uint8_t *)nullptr);
  return check_launch("dyn_splat_scatter");
}

uint8_t *dyn_splat_flag_map(void *workspace, int H, int W) {
  return reinterpret_cast<uint8_t *>(reinterpret_cast<float *>(workspace) + (size_t)5 * H * W);
}

int dyn_splat_scatter_part_fused(int H, int W, const float *rgb1, const float *rgb2, const float *flow12, const float *cam_tgt,
                                 const float *pcl, const uint8_t *keep, float *flow_1_to_tgt, float *valid_mask,
                                 const float *noise, const unsigned long long *rng, float alpha, void *workspace,
                                 bool flags_cleared, hipStream_t st) {
  const int P = H * W;
  float *acc = reinterpret_cast<float *>(workspace);
  uint8_t *flags = dyn_splat_flag_map(workspace, H, W);
  if (!flags_cleared) {
    hipError_t e = fill_async(flags, 0, (size_t)P, st);
    if (e != hipSuccess) {
      set_error("dyn_splat memset: %s", hipGetErrorString(e));
      return PGDVS_ERR_LAUNCH;
    }
  }
  dim3 grid((unsigned)cdiv(P, 256)), block(256);
  // (flow_1_to_tgt / valid_mask: null = not materialised -- both passes project the kept points themselves)
  PGDVS_LAUNCH("dyn_project_flag", dyn_project_flag_kernel, grid, block, 0, st, H, W, cam_tgt, pcl, keep, flow_1_to_tgt, valid_mask,
               flags, acc);
  const dim3 tgrid((unsigned)cdiv(W, kSplatTile), (unsigned)cdiv(H, kSplatTile));
  PGDVS_LAUNCH("dyn_splat_scatter", dyn_splat_scatter_kernel<true>, tgrid, block, 0, st, H, W, rgb1, rgb2, flow12,
               (const float *)nullptr, (const float *)nullptr, noise, rng, alpha, acc, (const uint8_t *)flags, cam_tgt, pcl, keep);
  return check_launch("dyn_splat_scatter");
}

int dyn_splat_finish_part(int H, int W, unsigned long long *rng, const float *static_rgb, float *render_dyn_rgb,
                          float *render_dyn_mask, float *combined, float *combined_static, float *combined_dyn, void *workspace,
                          hipStream_t st) {
  const int P = H * W;
  float *acc = reinterpret_cast<float *>(workspace);
  const uint8_t *flags = reinterpret_cast<const uint8_t *>(acc + (size_t)5 * P);
  dim3 grid((unsigned)cdiv(P, 256)), block(256);
  PGDVS_LAUNCH("dyn_splat_finish", dyn_splat_finish_kernel, grid, block, 0, st, P, acc, flags, static_rgb,
                     render_dyn_rgb, render_dyn_mask, combined, combined_static, combined_dyn, rng);
  return check_launch("dyn_splat_composite");
}
}  // namespace pgdvs

PGDVS_API int pgdvs_softsplat_bwd(const float *in, const float *flow, const float *outgrad, float *ingrad,
                                  float *flowgrad, int B, int C, int H, int W, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(in && flow && outgrad && (ingrad || flowgrad), "pgdvs_softsplat_bwd: null pointer");
  PGDVS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 31) && B < 65536,
                "pgdvs_softsplat_bwd: bad shape");
  const int64_t P = (int64_t)H * W;
  PGDVS_LAUNCH("softsplat_bwd", softsplat_bwd_kernel, dim3((unsigned)cdiv(P, 256), B), dim3(256), 0,
               as_stream(stream), in, flow, outgrad, ingrad, flowgrad, C, H, W);
  return check_launch("softsplat_bwd");
}
