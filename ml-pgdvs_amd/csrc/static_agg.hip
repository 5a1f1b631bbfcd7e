// A12: static point-cloud aggregation across the S source frames of a video
// (pgdvs/datasets/nvidia_eval_pure_geo.py:183-277, _compute_pcl at
// pgdvs/datasets/nvidia_eval.py:840-847, rays at pgdvs/datasets/base.py:507-546).
//
// Upstream this is single-threaded numpy at dataset construction.  Frame i may only add
// the static pixels that the cloud accumulated from frames < i does not already cover
// (integer-truncated projection occupancy), so frames are processed in order; inside a
// frame everything is data-parallel.
//   frame 0  : `select` appends every static pixel (count, ordered offsets inside the launch: tagged
//            8-byte count granules gathered from all predecessor tiles, dynamic tile tickets for forward
//            progress; rows in row-major pixel order -- the order numpy's boolean indexing produces; point
//            ids matter: the rasteriser breaks z ties by id), `push0` projects those points ONCE into every
//            later frame and stamps that frame's own occupancy map (one byte per pixel and frame, zeroed
//            per call).  This launch carries ~3/4 of all projections and fills the chip.
//   step i   : ONE launch per later frame (i = 1 .. S-1), a latency chain of S-1 links: a workgroup reads its
//            share of the frame's mask and of the frame's own occupancy map (complete: every earlier frame has
//            stamped it), lists the selected pixels in LDS, leaves their selection BITS for the end, unprojects
//            them and stamps the maps of the frames behind.  Neither order nor offsets exist inside the chain
//            (round 2: a selection launch that listed the pixels tile by tile + a push launch that scanned the
//            tile counts, built the rows and stamped: 3 + 11 us per frame; now 7.6):
//   rows     : at the end the selection bits of all later frames are counted per (frame, tile) and ONE
//            chip-filling launch builds their rows in the reference's order.
//            (Tried first: candidate bits cleared by memory-side atomic AND instead of occupancy bytes -- an eighth
//            of the map bytes, but a wavefront's 64 clears fall on scattered dwords of one or two bit rows, and
//            the atomic units take such an instruction lane by lane: 88 us per link.)
//   (PGDVS_AGG_ORDERED=1 keeps round 2's chain -- per frame an ordered selection and a byte-stamping push
//   that also builds the rows -- as an independent second implementation for the tests.)
//            numpy projects in fp64 (no z>0 test, no epsilon, closed bounds, astype(int)
//            truncation).  Decisions only change where a quotient crosses an integer, so a cheap
//            evaluation decides whenever its rigorous error bound keeps the quotient away from
//            every integer: first in fp32 (12 FMAs, v_rcp_f32; bound ~2e-3 px), the ~1 % it cannot
//            decide are parked in an LDS queue and re-evaluated densely in fp64 (screening form,
//            bound ~1e-9 px), and the one in ~1e9 still open takes the reference operation order
//            with correctly rounded divisions.
// No host synchronisation: the running point counts live on the device.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

namespace pgdvs {

struct ProjF64 {
  double K3[9];
  double w2c[16];
  int affine;  // w2c's last row is exactly (0,0,0,1): vc[3] == 1 and x/vc[3] == x bit for bit
  // K3 = [[fx,0,cx],[0,fy,cy],[0,0,1]] exactly: the skipped terms are +-0 (finite points) and
  // 1*z, so only the sign of an exact zero can differ -- which no decision below depends on
  int ksparse;
  // Screening form: M = K3 . w2c[0:3] composed on the host.  q~ = (M X)_k / (M X)_2 evaluated
  // with 9 FMAs differs from the value of the reference operation order by at most
  // tol_scale * max(|x|,|y|,|z|,1) * (1 + |q|) / |(M X)_2| (both are fp64 evaluations of the same
  // real number; tol_scale = 64 ulp x the largest row sum of |K3| |w2c|).  Decisions only change
  // at integers, so a q~ farther than that from every integer decides like the reference; the
  // others (one in ~1e9) take the reference operation order below.
  double M[12];
  double tol_scale;
  int screen;  // affine w2c and finite matrices
  // fp32 screening form of the same M: |s_k - S_k| <= e32[k] * max(|x|,|y|,|z|,1) for the fp32 FMA
  // evaluation s_k of row k (4.5 x 2^-24 x the row sum of |M|: entries rounded to fp32, three fused
  // multiply-adds), with the fp64 form's own tolerance folded in
  // (stored with rows 0 and 1 interleaved -- M00 M10 M01 M11 M02 M12 M03 M13, then row 2 -- so that one scalar load
  // leaves the pairs of the two-wide FMAs in aligned SGPR pairs)
  float M32[12];
  float e32[3];
  int screen32;
};

// Decides `q >= 0 && q <= hi` and trunc(q) for q = a / b (correctly rounded fp64 division,
// what numpy computes) without the division in the common case: q' = a * (1/b) with a
// Newton-refined reciprocal is within ~1e-15 relative of q; unless q' lies within 1e-13
// relative of an integer (all decision boundaries are integers) the decisions on q' and q
// coincide.  Otherwise fall back to the exact division.
__device__ __forceinline__ double refined_rcp(double b) {
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  return __builtin_fma(r, e, r);
}

// r = refined_rcp(b): both image coordinates divide by the same depth
__device__ __forceinline__ bool trunc_div_in_range(double a, double b, double r, int hi, int &out) {
  double q = a * r;
  double f = floor(q);
  double dist = fmin(q - f, (f + 1.0) - q);
  if (!(dist > fabs(q) * 1e-13 + 1e-290)) q = a / b;  // also taken for inf / NaN
  if (!(q >= 0.0 && q <= (double)hi)) return false;
  out = (int)q;
  return true;
}

// per-frame projection constants live in the workspace (written by agg_params_kernel): the
// marking kernel keeps only the screening form of its frames in registers and reads the
// reference form through this pointer in the rare doubtful case
constexpr int kProjChunk = 10;  // (kernel arguments are limited to 4 KB)
struct ProjChunk {
  ProjF64 p[kProjChunk];
};
static_assert(sizeof(ProjChunk) + 64 <= 4096, "agg_params_kernel's arguments");

struct PushConsts;
__global__ void agg_params_kernel(ProjChunk c, ProjF64 *__restrict__ dst, float *__restrict__ pc32, int first, int n, float q_hi) {
  const int words = (int)(sizeof(ProjF64) / 4);
  for (int k = threadIdx.x; k < n * words; k += blockDim.x)
    reinterpret_cast<uint32_t *>(dst + first)[k] = reinterpret_cast<const uint32_t *>(&c.p[0])[k];
  // the fp32 screening record (PushConsts: M32[12], e[3], pad), 16 floats per frame.  e[0], e[1]: the error bound of BOTH
  // image coordinates as one affine function of |r| amax for every quotient of magnitude <= q_hi = max(W, H) + 1 (see
  // screen_frames), rounded up; e[2] = 1/8 - e[1], rounded down
  for (int k = threadIdx.x; k < n * 16; k += blockDim.x) {
    const ProjF64 &pj = c.p[k >> 4];
    const int j = k & 15;
    float v = 0.0f;
    if (j < 12) {
      v = pj.M32[j];
    } else if (!pj.screen32) {
      v = j < 15 ? INFINITY : 0.0f;
    } else if (j == 12) {
      const double e01 = pj.e32[0] > pj.e32[1] ? (double)pj.e32[0] : (double)pj.e32[1];
      // (x 1.001: the fp32 roundings of |r| amax and of the product with it)
      v = (float)((e01 + ((double)q_hi + 1.0) * (double)pj.e32[2]) * 1.001 * 1.000001);
    } else if (j == 13) {
      v = (float)(((double)q_hi * 2.6e-7 + 1e-7) * 1.000001);
    } else if (j == 14) {
      v = (float)((0.125 - ((double)q_hi * 2.6e-7 + 1e-7) * 1.000001) * 0.999999);
    }
    pc32[(size_t)(first + (k >> 4)) * 16 + j] = v;
  }
}

// _compute_pcl_proj_mask, nvidia_eval_pure_geo.py:257-277 in the reference operation order (no
// z>0 test, no epsilon, closed bounds, astype(int) truncation)
// How a decided projection marks pixel q = row * W + col of frame f.
struct ByteStamp {  // one occupancy byte per (frame, pixel): frame 0's chip-filling push (coalesced byte stores)
  uint8_t *occ_all;
  int64_t P;
  // counters of the call (pgdvs_view_geo_counters): [0] projections that went all the way to the reference operation order,
  // [1 .. 64] points whose projections the fp32 form left to the fp64 queue -- summed per workgroup in LDS and added once per
  // workgroup and frame group to word 1 + (workgroup & 63): one hot word cost every link of the chain 4 us
  unsigned *stat;
#ifdef PGDVS_AB_CHAIN  // tools/r05_chain_cost.sh: a duplicate chain that computes everything and stores nothing
  __device__ __forceinline__ void operator()(int f, int q) const { if (occ_all) occ_all[(int64_t)f * P + q] = 1; }
#else
  // (frame base on the scalar unit, the pixel as an unsigned 32-bit offset: the store takes them as they are -- a 64-bit
  // f * P + q per lane cost three vector instructions per stamp)
  __device__ __forceinline__ void operator()(int f, int q) const { (occ_all + (int64_t)f * P)[(unsigned)q] = 1; }
#endif
};
// (experiment, tools/r06_ab.sh with EXTRA=-DPGDVS_AB_TBS: the chain's links look at the byte before they set it -- 70 % of
// their stamps hit bytes that frame 0's push or an earlier link has set already)
struct ByteStampTest {
  uint8_t *occ_all;
  int64_t P;
  unsigned *stat;
  __device__ __forceinline__ void operator()(int f, int q) const {
    uint8_t *p = occ_all + (int64_t)f * P;
    if (p[(unsigned)q] == 0) p[(unsigned)q] = 1;
  }
};
template <class Stamp>
__device__ __attribute__((noinline)) void mark_reference_order(const ProjF64 *__restrict__ pj, double x, double y,
                                                               double z, int H, int W, int f, const Stamp &stamp) {
  double vc[4];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double s = pj->w2c[k * 4 + 0] * x;
    s = s + pj->w2c[k * 4 + 1] * y;
    s = s + pj->w2c[k * 4 + 2] * z;
    s = s + pj->w2c[k * 4 + 3];
    vc[k] = s;
  }
  double cx = vc[0], cy = vc[1], cz = vc[2];
  if (!pj->affine) {
    double s = pj->w2c[12] * x;
    s = s + pj->w2c[13] * y;
    s = s + pj->w2c[14] * z;
    s = s + pj->w2c[15];
    cx = vc[0] / s;
    cy = vc[1] / s;
    cz = vc[2] / s;
  }
  double pp[3];
  if (pj->ksparse) {
    pp[0] = pj->K3[0] * cx + pj->K3[2] * cz;
    pp[1] = pj->K3[4] * cy + pj->K3[5] * cz;
    pp[2] = cz;
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double s = pj->K3[k * 3 + 0] * cx;
      s = s + pj->K3[k * 3 + 1] * cy;
      s = s + pj->K3[k * 3 + 2] * cz;
      pp[k] = s;
    }
  }
  int row, col;
  const double rz = refined_rcp(pp[2]);
  if (!trunc_div_in_range(pp[1], pp[2], rz, H - 1, row)) return;
  if (!trunc_div_in_range(pp[0], pp[2], rz, W - 1, col)) return;
  stamp(f, row * W + col);
}

// fp64 decision for one (point, frame): the screening form M = K3 . w2c decides unless a quotient lies
// within its rounding distance of an integer; then the reference operation order does
template <class Stamp>
__device__ __attribute__((noinline)) void mark_fp64(const ProjF64 *__restrict__ pj, float xf, float yf, float zf, int H,
                                                    int W, int f, const Stamp &stamp) {
  const double x = (double)xf, y = (double)yf, z = (double)zf;
  if (pj->screen) {
    const double *M = pj->M;
    const double amax = fmax(fmax(fabs(x), fabs(y)), fmax(fabs(z), 1.0));
    const double s0 = __builtin_fma(M[0], x, __builtin_fma(M[1], y, __builtin_fma(M[2], z, M[3])));
    const double s1 = __builtin_fma(M[4], x, __builtin_fma(M[5], y, __builtin_fma(M[6], z, M[7])));
    const double s2 = __builtin_fma(M[8], x, __builtin_fma(M[9], y, __builtin_fma(M[10], z, M[11])));
    const double r = refined_rcp(s2);
    const double qx = s0 * r, qy = s1 * r;
    const double tol = pj->tol_scale * amax * fabs(r);
    const double fx = floor(qx), fy = floor(qy);
    const double dx = fmin(qx - fx, (fx + 1.0) - qx), dy = fmin(qy - fy, (fy + 1.0) - qy);
    // (false for NaN / inf as well: those take the reference path)
    if (dx > tol * (1.0 + fabs(qx)) && dy > tol * (1.0 + fabs(qy))) {
      if (qy >= 0.0 && qy <= (double)(H - 1) && qx >= 0.0 && qx <= (double)(W - 1)) stamp(f, (int)qy * W + (int)qx);
      return;
    }
  }
  atomicAdd(stamp.stat, 1u);  // (one projection in ~1e9)
  mark_reference_order(pj, x, y, z, H, W, f, stamp);
}

// the fp32 screening constants of one frame: a 64-byte record of its own (one s_load_dwordx16, indexed by a shift)
typedef float f2v __attribute__((ext_vector_type(2)));
struct PushConsts {
  float M[12];  // in ProjF64::M32's order: (row 0, row 1) column pairs, then row 2
  float e[3];   // e[0] |r| amax + e[1] bounds the error of both quotients (agg_params_kernel); +inf without an fp32 form
  float pad;
};
static_assert(sizeof(PushConsts) == 64, "PushConsts");

// One row of the cloud from pixel p of a frame: unprojection exactly as _compute_pcl (fp32, fixed operation
// order), (xyz, rgb) to the cloud and the packed xyz copy; returns the point.  Shared by the selection of
// frame 0 (which appends in place) and by the push launches of the later frames (which append the pixels
// their frame's selection listed).
struct AppendSrc {
  const float *depth;  // frame's depth [P]
  const float *rgb;    // frame's colours [P,3]
  float *cloud;        // [capacity,6]
  float *xyz;          // [capacity,3]
  int P, W;
};
struct f3 { float x, y, z; };  // 12-byte loads / stores in one instruction

// (d_in: the pixel's depth if the caller already has it, NaN = load it here; *d_out receives the depth used)
__device__ __forceinline__ f3 append_row(const AppendSrc &a, const CamBlock &cam, int p, int64_t pos, bool write,
                                         float d_in = __builtin_nanf(""), float *d_out = nullptr, const f3 *c_in = nullptr) {
  // row / column of the pixel: float reciprocal + one correction step (exact for P < 2^24),
  // integer division otherwise
  int r, col;
  if (a.P < (1 << 24)) {
    r = (int)(((float)p + 0.5f) * (1.0f / (float)a.W));
    col = p - r * a.W;
    if (col < 0) {
      --r;
      col += a.W;
    } else if (col >= a.W) {
      ++r;
      col -= a.W;
    }
  } else {
    r = p / a.W;
    col = p - r * a.W;
  }
  const float u = (float)col, v = (float)r;
  const float d = d_in == d_in ? d_in : a.depth[p];
  if (d_out) *d_out = d;
  float X[3];
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    float dir = cam.v[PGDVS_CAM_M + ax * 3 + 0] * u;
    dir = dir + cam.v[PGDVS_CAM_M + ax * 3 + 1] * v;
    dir = dir + cam.v[PGDVS_CAM_M + ax * 3 + 2];
    X[ax] = cam.v[PGDVS_CAM_O + ax] + dir * d;
  }
  f3 xq;
  xq.x = X[0];
  xq.y = X[1];
  xq.z = X[2];
  if (write) {
    const f3 c = c_in ? *c_in : *reinterpret_cast<const f3 *>(a.rgb + (size_t)p * 3);
    // a cloud row is 24 bytes at an 8-byte aligned address: three 8-byte stores
    float2 *o = reinterpret_cast<float2 *>(a.cloud + pos * 6);
    o[0] = make_float2(X[0], X[1]);
    o[1] = make_float2(X[2], c.x);
    o[2] = make_float2(c.y, c.z);
    *reinterpret_cast<f3 *>(a.xyz + pos * 3) = xq;
  }
  return xq;
}

constexpr int kSelThreads = 512;
constexpr int kSelItems = 16;
constexpr int kSelTile = kSelThreads * kSelItems;
constexpr int kPushThreads = 256;
// points with frames the fp32 form could not decide, queued per workgroup: 1024 entries for frame 0's chip-filling
// launch (drained every fourth round); 512 (every second round) for the later frames, whose launches are a few
// rounds long
constexpr int kPushQueue = 1024;
constexpr int kPushQueueSmall = 512;
constexpr int kPushMaxFpg = 32;    // frames per workgroup row (one bit each in a queue entry)

// One point against the frames [fa, fb): fp32 screening of each projection, stamp where it decides; returns the
// bitmask (bit f - fa) of the frames it could not decide (lanes that are not `live`: all bits).  Per-frame constants
// are wave-uniform scalar loads.
//   q~ = (M32 X)_k / (M32 X)_2 differs from the reference's fp64 value by at most T (see ProjF64 and below).  With
//   sure = T < 1/8, clear = min(dx, dy) > T (dx, dy: distance of q~ to the nearest integer) and
//   mi = min(qx, W-1 - qx, qy, H-1 - qy):   stamp  <=>  sure and clear and mi >= 0        (in range, both coordinates)
//                                          decided <=>  sure and (clear or mi < -1/2)    (else: fp64)
//   "sure" implies that every quantity is finite (NaN / inf coordinates, a zero denominator: its left-hand side is NaN
//   or inf and the comparison false).
//   The decisions are arithmetic (min / max on the vector unit) with four comparisons at the end: written as twelve
//   comparisons combined in scalar registers the loop spent 42 scalar against 50 vector instructions per frame, and
//   the scalar unit -- one per CU for four SIMDs -- set the pace of frame 0's launch (38 M scalar wave-instructions
//   = 62 of its 72 us).
// (`rec(f)` returns frame f's record: wave-uniform scalar loads from the record array in agg_push_kernel -- every workgroup
// walks many rounds of points and the records stay in the scalar cache --, broadcast reads of an LDS copy in the chain's
// links, where a wavefront screens ONE round and a record fetched when its frame comes up is a trip to the L2 per frame)
template <class Stamp, class Rec>
__device__ __forceinline__ unsigned screen_frames(const Rec &rec, const int fa, const int fb,
                                                  const bool live, float x, float y, float z, const float wm1,
                                                  const float hm1, const int W, const Stamp &stamp) {
  if (!live) x = y = z = __builtin_nanf("");  // decides nothing, stamps nothing
  const float amax = fmaxf(fmaxf(fabsf(x), fabsf(y)), fmaxf(fabsf(z), 1.0f));
  unsigned dmask = 0;
  auto frame = [&](const int f, const PushConsts &c) {
    const float *M = c.M;
    // rows 0 and 1 as one two-wide FMA chain (v_pk_fma_f32 on the SGPR pairs as loaded); the same three fused
    // multiply-adds per row
    const f2v p0 = {M[0], M[1]}, p1 = {M[2], M[3]}, p2 = {M[4], M[5]}, p3 = {M[6], M[7]};
    const f2v xx = {x, x}, yy = {y, y}, zz = {z, z};
    const f2v s01 = __builtin_elementwise_fma(p0, xx, __builtin_elementwise_fma(p1, yy, __builtin_elementwise_fma(p2, zz, p3)));
    const float s0 = s01.x, s1 = s01.y;
    const float s2 = __builtin_fmaf(M[8], x, __builtin_fmaf(M[9], y, __builtin_fmaf(M[10], z, M[11])));
    const float r = __builtin_amdgcn_rcpf(s2);  // v_rcp_f32: 1 ulp
    const float qx = s0 * r, qy = s1 * r;
    // |q~ - q_ref| <= (E_row + (|q~| + 1) E_2) |r| + 2^-22 |q~|, E_k = e32[k] amax (see ProjF64): for |q~| <= Q = max(W, H) + 1
    // at most T = e0 |r| amax + e[1] with e0 = max(e32[0], e32[1]) + (Q + 1) e32[2], e[1] = 2.6e-7 Q + 1e-7 (the rounding of
    // dx, dy and of the subtraction below included) -- one bound for both coordinates.  A quotient beyond Q can be off by
    // more than T, but it never stamps (mi < 0) and "decided" is right for it: sure means e0 |r| amax < 1/8, hence its true error is below
    // 1/8 + (|q~| + 1) / (8 (Q + 1)) + 2.5e-7 |q~| and the true quotient still lies more than 1/2 outside [0, max(W, H) - 1].
    // The two constants stay in scalar registers: u = e[0] |r| amax (e[0] carries the rounding allowance of this product),
    // "sure" <=> u < e[2] = 1/8 - e[1], "clear of every integer" <=> min(dx, dy) - u > e[1].
    const float u = c.e[0] * (fabsf(r) * amax);
    // distance to the nearest integer: v_fract_f32 = qx - floor(qx) in ONE instruction (round 6; floor + subtract until then).
    // The two agree bit for bit wherever a projection can stamp (qx >= 0: the difference is exact); the instruction clamps
    // its result below 1, which only shows for -6e-8 < qx < 0 -- 5.96e-8 from an integer instead of 0: "not clear" either
    // way.  1 - that is the same real number as (floor + 1) - qx.
    const float gx = __builtin_amdgcn_fractf(qx), gy = __builtin_amdgcn_fractf(qy);
    const float dx = fminf(gx, 1.0f - gx), dy = fminf(gy, 1.0f - gy);
    // (0 * (qx + qy): NaN for an infinite or NaN quotient, so that "sure" still implies finite coordinates)
    // (|s2| < 2^100: beyond that v_rcp_f32's result is denormal -- flushed or short of 24 significant bits -- and the
    // 1-ulp assumption behind the bound does not hold; such projections, coordinates around 1e30, take the fp64 queue)
    const bool sure = (__builtin_fmaf(qx + qy, 0.0f, u) < c.e[2]) & (fabsf(s2) < 0x1p100f);
    const bool clear = fminf(dx, dy) - u > c.e[1];
    const float mi = fminf(fminf(qx, wm1 - qx), fminf(qy, hm1 - qy));
    if (sure & clear & (mi >= 0.0f)) stamp(f, (int)qy * W + (int)qx);
    dmask |= ((sure & (clear | (mi < -0.5f))) ? 0u : 1u) << (f - fa);
  };
  // Two frames per trip with two sets of constants: each set is requested (scalar loads, wave-uniform) while the
  // other frame is evaluated and lands in its own registers -- a single set renamed per frame cost ~14 scalar
  // moves per projection.
  // (the record array has two spare entries behind the last frame: the requests ahead need no clamping)
  PushConsts ca = rec(fa);
  for (int f = fa; f < fb; f += 2) {
    const PushConsts cb = rec(f + 1);
    frame(f, ca);
    if (f + 1 < fb) {
      ca = rec(f + 2);
      frame(f + 1, cb);
    }
  }
  return live ? dmask : 0u;
}

// The (point, frame) pairs the fp32 form could not decide: one LDS queue entry per point (the point itself and a
// frame bitmask), drained densely through the fp64 forms when the next round could overflow the queue (every
// thread adds at most one entry per round) and after the last round.  Called by every thread of the workgroup.
constexpr int kAggStatWords = 65;
template <int kQueue, int kThreads, class Stamp>
__device__ __forceinline__ void queue_doubtful(uint4 *s_q, int *s_qn, const unsigned dmask, const float x, const float y,
                                               const float z, const int fa, const bool last,
                                               const ProjF64 *__restrict__ proj, const int H, const int W, const Stamp &stamp) {
  if (dmask != 0) {
    const int slot = atomicAdd(s_qn, 1);
    if (slot < kQueue) {
      s_q[slot] = make_uint4(__float_as_uint(x), __float_as_uint(y), __float_as_uint(z), dmask);
    } else {  // queue full (degenerate views): decide in place
      atomicAdd(stamp.stat + 1 + (blockIdx.x & 63), 1u);
      for (unsigned m = dmask; m; m &= m - 1) {
        const int f = fa + __builtin_ctz(m);
        mark_fp64(proj + f, x, y, z, H, W, f, stamp);
      }
    }
  }
  __syncthreads();
  const int qn = *s_qn < kQueue ? *s_qn : kQueue;
  // every wavefront has read the count before any of them can add to it again (next round): the decision to
  // drain -- and with it the barriers inside -- is the same on all of them
  __syncthreads();
  if (last || qn + kThreads > kQueue) {
    if (threadIdx.x == 0) s_qn[1] += qn;
    for (int e = threadIdx.x; e < qn; e += kThreads) {
      const uint4 q = s_q[e];
      const float px = __uint_as_float(q.x), py = __uint_as_float(q.y), pz = __uint_as_float(q.z);
      for (unsigned m = q.w; m; m &= m - 1) {
        const int f = fa + __builtin_ctz(m);
        mark_fp64(proj + f, px, py, pz, H, W, f, stamp);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      *s_qn = 0;
      if (last && s_qn[1] > 0) {
        atomicAdd(stamp.stat + 1 + (blockIdx.x & 63), (unsigned)s_qn[1]);
        s_qn[1] = 0;
      }
    }
    __syncthreads();
  }
}

// Projects the points [cnts[src], cnts[src+1]) -- what frame `src` appended -- into the frames
// f_lo + blockIdx.y * fpg ... (at most fpg of them, below f_hi) and stamps their occupancy maps
// occ[f][P].  One thread per point, the frame loop inside.  Frame 0's launch (the chip-filling one) on the
// default path; with PGDVS_AGG_ORDERED=1 every frame's, and then a later frame's launch also builds the rows of
// the pixels its ordered selection listed in sel_pix (cloud order).
template <int kQueue>
__global__ void __launch_bounds__(kPushThreads)
agg_push_kernel(const float *__restrict__ xyz, const int64_t *__restrict__ cnts, int src,
                const ProjF64 *__restrict__ proj, const PushConsts *__restrict__ pc, int f_lo, int f_hi, int fpg, int H, int W,
                uint8_t *__restrict__ occ_all, const int32_t *__restrict__ sel_pix, AppendSrc app, CamBlock cam,
                unsigned *__restrict__ stat) {
  __shared__ uint4 s_q[kQueue];
  __shared__ int s_qn[2];  // queue length; points queued since the last report to the statistics words
  const int64_t begin = cnts[src], end = cnts[src + 1];
  if (begin >= end) return;
  const ByteStamp stamp{occ_all, (int64_t)H * W, stat};
  const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
  if (threadIdx.x == 0) s_qn[0] = s_qn[1] = 0;
  __syncthreads();
  // frame groups side by side (gridDim.y covers them; a smaller gridDim.y walks them one after the other:
  // tried for frame 0 so that all workgroups stamp the same <= fpg maps at a time -- 98 us against 81 us,
  // and the same 4.8 bytes written per stamp)
  // (with a deferred append, frame group 0 runs once even when there is no later frame to stamp)
  bool first_iter = true;
  for (int fa = f_lo + (int)blockIdx.y * fpg; fa < f_hi || (first_iter && sel_pix != nullptr && blockIdx.y == 0);
       fa += (int)gridDim.y * fpg, first_iter = false) {
  const int fb = fa >= f_hi ? fa : (fa + fpg < f_hi ? fa + fpg : f_hi);
  // XCD-aware order: workgroups b, b+8, ... share an XCD (and its L2), so each XCD walks one contiguous
  // eighth of the points -- a band of source rows whose stamps fall into a band of the maps.  Interleaved
  // 256-point chunks made neighbouring chunks (same row, adjacent columns) dirty the same 128-byte lines
  // in different L2s: every such line went to memory twice.
  const int64_t n_chunks = (end - begin + kPushThreads - 1) / kPushThreads;
  const int64_t per_xcd = (n_chunks + 7) / 8;
  const int64_t c_lo = (int64_t)(blockIdx.x & 7) * per_xcd;
  const int64_t c_hi = c_lo + per_xcd < n_chunks ? c_lo + per_xcd : n_chunks;
  const int64_t c_step = gridDim.x >> 3 ? gridDim.x >> 3 : 1;
  // whole workgroups stay in the loop so that the queue can be drained between rounds
  for (int64_t ch = c_lo + (int64_t)(blockIdx.x >> 3); ch < c_hi; ch += c_step) {
    const int64_t i = begin + ch * kPushThreads + threadIdx.x;
    const bool live = i < end;
    float x = 0.f, y = 0.f, z = 0.f;
    if (live && sel_pix != nullptr) {
      // deferred append: the frame's selection only listed its pixels (in cloud order); the row is built
      // here, where the work is spread over the whole grid however the pixels cluster (frame group 0 writes)
      const f3 X = append_row(app, cam, sel_pix[i - begin], i, blockIdx.y == 0 && first_iter);
      x = X.x;
      y = X.y;
      z = X.z;
    } else if (live) {
      x = xyz[i * 3 + 0];
      y = xyz[i * 3 + 1];
      z = xyz[i * 3 + 2];
    }
    unsigned dmask = 0;
    if (fa < fb) dmask = screen_frames([pc](const int f) { return pc[f]; }, fa, fb, live, x, y, z, wm1, hm1, W, stamp);
    queue_doubtful<kQueue, kPushThreads>(s_q, s_qn, dmask, x, y, z, fa, ch + c_step >= c_hi, proj, H, W, stamp);
  }
  }
}

// ---- the later frames: one `step` launch per frame, rows at the end -------------------------------------------------

constexpr int kBitTileWords = kSelTile / 32;  // a tile of 8192 pixels = 256 words, one per thread
constexpr int kStepThreads = 256;
constexpr int kStepPx = 16;        // pixels per thread of a step launch
constexpr int kStepChunkPx = 128;  // ... in chunks of 128 consecutive pixels (8 threads), dealt round-robin to the workgroups
constexpr int kStepChunks = kStepThreads * kStepPx / kStepChunkPx;  // chunks per workgroup (32: 4096 pixels)

// Which 128-pixel chunk of the frame is chunk r (0 .. kStepChunks-1) of a link's workgroup b, gx workgroups per link (a
// multiple of 8), and the inverse.  Workgroup b runs on XCD b mod 8 (round-robin dispatch).  Until round 6 chunk g belonged to
// workgroup g mod gx: neighbouring chunks -- whose new points stamp neighbouring pixels of the later frames, i.e. the same
// 128-byte lines of their maps -- sat on eight different XCDs, and every such line went to memory from several L2s.  Now the
// frame is cut into stripes of kStepChunks = 32 chunks (4096 pixels, ~2 rows at 1080p), stripe s belongs to XCD s mod 8, and
// the stripes of an XCD are dealt chunk by chunk over its gx / 8 workgroups as before (every workgroup still holds the same
// sample of the image, every XCD every eighth stripe).  A bijection on [0, 32 gx) for every gx that is a multiple of 8.
__device__ __forceinline__ int64_t step_chunk_of(const int b, const int r, const int gx) {
  const int L = gx >> 3, xcd = b & 7, l = b >> 3;
  const int j = r * L + l;  // the chunk's number among its XCD's chunks
  return ((int64_t)(j / kStepChunks) * 8 + xcd) * kStepChunks + (j % kStepChunks);
}
__device__ __forceinline__ void step_owner_of(const int64_t g, const int gx, int &b, int &r) {
  const int L = gx >> 3;
  const int64_t stripe = g / kStepChunks;
  const int xcd = (int)(stripe & 7);
  const int j = (int)(stripe >> 3) * kStepChunks + (int)(g % kStepChunks);
  r = j / L;
  b = (j - r * L) * 8 + xcd;
}

// block-wide exclusive offset of `c` (256 threads) and the block total
__device__ __forceinline__ int block_excl_256(const int c, int *s_wsum, int &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = c;
  for (int off = 1; off < 64; off <<= 1) {
    const int y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) s_wsum[wave] = x;
  __syncthreads();
  int wave_off = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) wave_off += s_wsum[w];
    total += s_wsum[w];
  }
  return wave_off + x - c;
}

constexpr unsigned kSelSpinLimit = 1u << 22;

// tile count granule: [63:48] frame tag, [47:46] status (unused, 1), [45:0] value
__device__ __forceinline__ unsigned long long sel_desc(int tag, int status, long long v) {
  return ((unsigned long long)tag << 48) | ((unsigned long long)status << 46) | (unsigned long long)v;
}

struct SelArgs {
  const uint8_t *dyn_mask;   // frame's mask [P]
  const uint8_t *occ;        // this frame's occupancy map [P], written by the earlier frames' push launches
  const float *depth;        // frame's depth [P]
  const float *rgb;          // frame's colours [P,3]
  float *cloud;              // [capacity,6]
  float *xyz;                // [capacity,3] packed copy for the mark pass
  int64_t *cnts;             // [S+1] cloud size before each frame; cnts[S] = final
  unsigned long long *desc;  // [tiles]
  int32_t *ticket;           // [S]
  int32_t *error;            // set when a look-back spin gives up
  int32_t *sel_pix;          // deferred append (ordered chain, frames >= 1): the selected pixels in cloud order, [P]
  int64_t capacity;
  int frame, P, W, tiles;
};

// selection flags of 16 consecutive pixels: static and (frame 0 or) not stamped by this
// frame's mark pass (tmp_st_mask & ~tmp_proj_mask, :224-245)
__device__ __forceinline__ unsigned sel_flags16(const SelArgs &a, int base) {
  unsigned flags = 0;
  if (base + kSelItems <= a.P && ((reinterpret_cast<uintptr_t>(a.dyn_mask + base) & 15) == 0)) {
    const uint4 m = *reinterpret_cast<const uint4 *>(a.dyn_mask + base);
    const unsigned mw[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
    for (int k = 0; k < 16; ++k) flags |= (((mw[k >> 2] >> ((k & 3) * 8)) & 0xffu) == 0u ? 1u : 0u) << k;
    if (a.frame > 0 && ((reinterpret_cast<uintptr_t>(a.occ + base) & 15) == 0)) {
      const uint4 o = *reinterpret_cast<const uint4 *>(a.occ + base);
      const unsigned ow[4] = {o.x, o.y, o.z, o.w};
      unsigned stamped = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) stamped |= (((ow[k >> 2] >> ((k & 3) * 8)) & 0xffu) != 0u ? 1u : 0u) << k;
      flags &= ~stamped;
    } else if (a.frame > 0) {
      unsigned stamped = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) stamped |= (a.occ[base + k] != 0 ? 1u : 0u) << k;
      flags &= ~stamped;
    }
  } else {
    for (int k = 0; k < kSelItems; ++k) {
      const int p = base + k;
      if (p >= a.P) break;
      bool st = a.dyn_mask[p] == 0;
      if (a.frame > 0) st = st && a.occ[p] == 0;
      flags |= (st ? 1u : 0u) << k;
    }
  }
  return flags;
}

// one bit per byte of four dwords: bit k = (byte k is zero)
__device__ __forceinline__ unsigned zero_bytes16(const uint4 v) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
  unsigned bits = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // high bit of every non-zero byte, gathered into four adjacent bits by one multiplication
    const unsigned nz = ((((w[k] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w[k]) >> 7) & 0x01010101u;
    bits |= ((nz * 0x10204080u) >> 28) << (4 * k);
  }
  return ~bits & 0xffffu;
}

__global__ void __launch_bounds__(kSelThreads) agg_select_kernel(SelArgs a, CamBlock cam) {
  __shared__ int s_tile;
  __shared__ int wave_sums[kSelThreads / kWave];
  __shared__ long long s_excl;
  __shared__ uint16_t s_list[kSelTile];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // dynamic tile id (one atomic per block): a tile's predecessors are always owned by blocks that
  // already run, whatever order the dispatcher starts blocks in (blockIdx order is NOT dispatch
  // order: workgroups go round-robin over the 8 XCDs, and other views' kernels share the CUs)
  if (tid == 0) s_tile = atomicAdd(&a.ticket[a.frame], 1);
  __syncthreads();
  const int tile = s_tile;
  const int base = tile * kSelTile + tid * kSelItems;
  const unsigned flags = base < a.P ? sel_flags16(a, base) : 0u;
  const int c = __popc(flags);
  int x = c;
  for (int off = 1; off < 64; off <<= 1) {
    int y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) wave_sums[wave] = x;
  __syncthreads();
  int wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kSelThreads / kWave; ++w) {
    if (w < wave) wave_off += wave_sums[w];
    total += wave_sums[w];
  }
  // Ordered offsets without a second launch: every tile publishes its count as one tagged
  // 8-byte granule, then sums the granules of ALL its predecessors (each thread polls its own
  // few words, one visibility round trip in total -- every tile of a frame is resident at the
  // same time, so a chained look-back would serialise instead).
  const int tag = a.frame + 1;
  if (tid == 0)
    __hip_atomic_store(&a.desc[tile], sel_desc(tag, 1, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // a tile that selects nothing appends nothing: it needs no offset and leaves at once instead of polling
  // beside the other views' kernels (in the later frames most tiles are like this); the last tile stays to
  // record the new cloud size
  if (total == 0 && tile != a.tiles - 1) return;
  long long part = 0;
  for (int j = tid; j < tile; j += kSelThreads) {
    unsigned spins = 0;
    unsigned long long d;
    while (true) {
      d = __hip_atomic_load(&a.desc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((int)(d >> 48) == tag) break;
      if (++spins > kSelSpinLimit) {  // never expected; keeps a protocol bug from hanging the GPU
        atomicExch(a.error, 1);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    part += (long long)(d & ((1ull << 46) - 1));
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  __shared__ long long wave_part[kSelThreads / kWave];
  if (lane == 0) wave_part[wave] = part;
  __syncthreads();
  if (tid == 0) {
    long long e = 0;
#pragma unroll
    for (int w = 0; w < kSelThreads / kWave; ++w) e += wave_part[w];
    s_excl = e;
  }
  __syncthreads();
  const int64_t cloud_base = a.cnts[a.frame];
  if (tile == a.tiles - 1 && tid == 0) {
    int64_t n = cloud_base + s_excl + total;
    n = n > a.capacity ? a.capacity : n;
    a.cnts[a.frame + 1] = n;
  }
  // Ordered append (tmp_pcl[tmp_st_mask], tmp_img[tmp_st_mask] :247-251): the tile's selected
  // pixels are first compacted into an LDS list (row-major rank order), then the whole block
  // walks the list -- consecutive threads write consecutive cloud rows and every thread has
  // several independent loads in flight, however the selected pixels cluster.
  if (total == 0) return;
  {
    int slot = wave_off + x - c;
    unsigned f = flags;
    while (f) {
      const int k = __builtin_ctz(f);
      f &= f - 1;
      s_list[slot++] = (uint16_t)(tid * kSelItems + k);
    }
  }
  __syncthreads();
  const int64_t pos0 = cloud_base + s_excl;
  const int tile_px = tile * kSelTile;
  if (a.sel_pix != nullptr) {
    // later frames: only the list of selected pixels, in cloud order; the rows are built by the frame's push
    // launch, whose grid shares the work evenly (here a few tiles hold all of a disocclusion's pixels)
    for (int e = tid; e < total; e += kSelThreads) {
      const int64_t pos = pos0 + e;
      if (pos >= a.capacity) break;
      a.sel_pix[pos - cloud_base] = tile_px + (int)s_list[e];
    }
    return;
  }
  AppendSrc app;
  app.depth = a.depth;
  app.rgb = a.rgb;
  app.cloud = a.cloud;
  app.xyz = a.xyz;
  app.P = a.P;
  app.W = a.W;
#pragma unroll 4
  for (int e = tid; e < total; e += kSelThreads) {
    const int64_t pos = pos0 + e;
    if (pos >= a.capacity) break;
    append_row(app, cam, tile_px + (int)s_list[e], pos, true);
  }
}

// selection flags of 16 pixels of a later frame, both 16-byte loads in flight together (sel_flags16 waits for each in turn:
// dependent round trips at the head of every chain link); selected = mask byte zero and map byte zero = (mask | map) zero
__device__ __forceinline__ unsigned sel_flags16_pair(const SelArgs &a, const int64_t base) {
  if (base + 16 <= a.P && (((reinterpret_cast<uintptr_t>(a.dyn_mask + base) | reinterpret_cast<uintptr_t>(a.occ + base)) & 15) == 0)) {
    const uint4 m = *reinterpret_cast<const uint4 *>(a.dyn_mask + base);
    const uint4 o = *reinterpret_cast<const uint4 *>(a.occ + base);
    return zero_bytes16(make_uint4(m.x | o.x, m.y | o.y, m.z | o.z, m.w | o.w));
  }
  return base < a.P ? sel_flags16(a, (int)base) : 0u;
}

// Rows of the later frames on their way from the chain links to agg_rows: the selected pixels of a link's workgroup in list
// order -- chunk after chunk, pixel order inside a chunk -- as (depth, r, g, b): one 16-byte store per pixel in the link, DENSE
// per workgroup (round 5: rows[((frame * gx + workgroup) * 4096 + list position) * 4]; until then every 128-pixel chunk packed
// its rows at the chunk's own pixel base, 60 k partial cache lines per link), and the list position of every chunk's first
// selected pixel (cst[(frame * gx + workgroup) * 32 + chunk], 16 bits) so that agg_rows finds them.  (The finished row, 24
// bytes in three stores, cost the link 1.4 us: more than agg_rows gained; a cursor shared by the 512 workgroups of a link
// cost the link 5 us.)  null: nothing staged, agg_rows gathers depth and colour itself.
struct RowStage {
  float *rows;
  uint16_t *cst;
  int gx;  // workgroups per link (chunk g of a frame belongs to workgroup g % gx, its chunk g / gx)
};

// One link of the chain: frame `src`'s selection = static and not stamped (tmp_st_mask & ~tmp_proj_mask, :224-245);
// its own map is complete when the launch starts.  Workgroup b takes the 128-pixel chunks b, b + gridDim.x, ... (32 of
// them, 16 pixels per thread: selected pixels cluster -- whole rows at an image border, bands around depth edges --
// and chunks dealt round-robin give every workgroup the same sample of the image; with 64-word groups the slowest
// workgroup ran 10+ rounds of the loop below and WAS the launch: 21 us), lists the selected pixels in LDS and leaves
// their bits in sel[src] (row blockIdx.y == 0); every thread then unprojects its pixels (as _compute_pcl) and screens
// them against the frames src + 1 + blockIdx.y * fpg ... like agg_push_kernel.  gridDim.x is a multiple of 8: the
// workgroups (x, 0), (x, 1), ... land on one XCD and share its L2's copy of the chunk (the rows re-read the same 4 MB).
// The last frame's launch only leaves its bits.
template <int kQueue>
__global__ void __launch_bounds__(kStepThreads)
agg_step_kernel(SelArgs a, uint8_t *__restrict__ occ_all, uint16_t *__restrict__ sel16, int64_t Wd, int src,
                const ProjF64 *__restrict__ proj, const PushConsts *__restrict__ pc, int f_hi, int fpg, int H, int W,
                AppendSrc app, CamBlock cam, unsigned *__restrict__ stat, RowStage stage) {
  __shared__ uint16_t s_list[kStepThreads * kStepPx];  // (thread << 4 | pixel) of every selected pixel, 8 KB
  __shared__ int s_cstart[kStepChunks];  // list position of every chunk's first selected pixel
  __shared__ uint4 s_q[kQueue];
  __shared__ int s_qn[2];
  __shared__ int s_wsum[4];
  __shared__ PushConsts s_pc[kPushMaxFpg + 2];  // the row's screening records (+ the two spare entries screen_frames may request)
  const int tid = threadIdx.x;
  auto pixel_base = [&](const int t) {
    return step_chunk_of((int)blockIdx.x, t >> 3, (int)gridDim.x) * kStepChunkPx + (t & 7) * kStepPx;
  };
  const int64_t base = pixel_base(tid);
  // the row's records ride on the first round trip, beside the mask and map bytes (16 bytes per thread: four per record):
  // fetched one frame ahead through the scalar cache they were a trip to the L2 per frame, in a loop a wavefront runs ONCE
  const int fa = src + 1 + (int)blockIdx.y * fpg;
  float4 rec4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool rec_mine = tid < (kPushMaxFpg + 2) * 4 && fa + (tid >> 2) < f_hi + 2;
  if (rec_mine) rec4 = reinterpret_cast<const float4 *>(pc + fa)[tid];
  const uint32_t bits = sel_flags16_pair(a, base);
  {
    // the chunk's 128 selection bits as ONE 16-byte store by its first thread, and only where a pixel is selected (sel is
    // zeroed per call with the maps): 2-byte stores from every thread were partial writes of ~130 k cache lines per link
    const uint32_t pair = bits | ((uint32_t)__shfl_down((int)bits, 1, 8) << 16);
    const uint32_t p1 = (uint32_t)__shfl_down((int)pair, 2, 8), p2 = (uint32_t)__shfl_down((int)pair, 4, 8), p3 = (uint32_t)__shfl_down((int)pair, 6, 8);
#ifdef PGDVS_AB_CHAIN
    if (sel16 != nullptr)
#endif
    if (blockIdx.y == 0 && (tid & 7) == 0 && base < Wd * 32 && (pair | p1 | p2 | p3) != 0u)
      *reinterpret_cast<uint4 *>(sel16 + (((int64_t)src * Wd * 32 + base) >> 4)) = make_uint4(pair, p1, p2, p3);
  }
  if (fa >= f_hi) return;
  if (rec_mine) reinterpret_cast<float4 *>(s_pc)[tid] = rec4;
  const int fb = fa + fpg < f_hi ? fa + fpg : f_hi;
  int n;
  int slot = block_excl_256(__popc(bits), s_wsum, n);
  if (n == 0) return;
  const bool staged = blockIdx.y == 0 && stage.rows != nullptr;
  const int64_t wg_slot = (int64_t)src * stage.gx + blockIdx.x;  // this workgroup's block of staged rows / chunk starts
  if ((tid & 7) == 0) {
    s_cstart[tid >> 3] = slot;
    if (staged) stage.cst[wg_slot * kStepChunks + (tid >> 3)] = (uint16_t)slot;
  }
  for (uint32_t m = bits; m; m &= m - 1) s_list[slot++] = (uint16_t)((tid << 4) | __builtin_ctz(m));
  if (tid == 0) s_qn[0] = s_qn[1] = 0;
  __syncthreads();
#ifdef PGDVS_AB_TBS
  const ByteStampTest stamp{occ_all, (int64_t)H * W, stat};
#else
  const ByteStamp stamp{occ_all, (int64_t)H * W, stat};
#endif
  const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
  // Round 6: a workgroup lists ~120 pixels (60 k per link over 512 workgroups), so with one thread per listed pixel two of its
  // four wavefronts had nothing to screen -- and ran the whole frame loop on NaNs beside the other two, which walked all the
  // later frames one after the other.  Now the (pixel, frame) pairs are dealt over all four: up to 64 listed pixels, every
  // wavefront screens a quarter of the frames; up to 128, a half (`part`: wave-uniform; every part unprojects its pixels
  // itself -- the same depth, the same operations); a wavefront without a listed pixel skips the loop.  Same stamps, same
  // queue decisions (an entry's frame bits are relative to fa whichever part queued it).
  const int F = fb - fa;
  const int parts = (n <= 64 && F >= 4) ? 4 : ((n <= 128 && F >= 2) ? 2 : 1);
  const int R = kStepThreads / parts;  // listed pixels per round
  const int part = __builtin_amdgcn_readfirstlane(tid / R), el = tid - part * R;
  const int per = (F + parts - 1) / parts;
  const int fa_p = fa + part * per;
  const int fb_p = fa_p + per < fb ? fa_p + per : fb;
  for (int e0 = 0; e0 < n; e0 += R) {
    const int e = e0 + el;
    const bool live = e < n;
    float x = 0.f, y = 0.f, z = 0.f, d = 0.f;
    f3 c = {0.f, 0.f, 0.f};
    int64_t slot4 = -1;
    if (live) {
      const int ent = s_list[e];
      const int px = (int)(pixel_base(ent >> 4) + (ent & 15));
      // the depth first, the colour right behind it: the unprojection and the stamps below wait for the depth only, the
      // colour is still on its way while they run and is consumed at the very end of the round -- both are left for
      // agg_rows, which then reads 16 dense bytes per row instead of a sector per scattered depth and another per colour
      d = app.depth[px];
      if (staged && part == 0) {
        c = *reinterpret_cast<const f3 *>(app.rgb + (size_t)px * 3);
        slot4 = (wg_slot * (kStepThreads * kStepPx) + e) * 4;
      }
      const f3 X = append_row(app, cam, px, 0, false, d);
      x = X.x;
      y = X.y;
      z = X.z;
    }
    unsigned dmask = 0;
    if (__ballot(live) != 0ull && fa_p < fb_p)
      dmask = screen_frames([&](const int f) { return s_pc[f - fa]; }, fa_p, fb_p, live, x, y, z, wm1, hm1, W, stamp) << (fa_p - fa);
    queue_doubtful<kQueue, kStepThreads>(s_q, s_qn, dmask, x, y, z, fa, e0 + R >= n, proj, H, W, stamp);
    if (slot4 >= 0) *reinterpret_cast<float4 *>(stage.rows + slot4) = make_float4(d, c.x, c.y, c.z);
  }
}

// After the chain: selected pixels per (frame, tile of 8192 pixels), tile_cnt[f][t] (frames >= 1)
__global__ void __launch_bounds__(kBitTileWords) agg_count_kernel(const uint32_t *__restrict__ sel, int64_t Wd, int tiles,
                                                                  int32_t *__restrict__ tile_cnt) {
  __shared__ int s_wsum[4];
  const int f = 1 + (int)blockIdx.y;
  int total;
  block_excl_256(__popc(sel[(int64_t)f * Wd + (int64_t)blockIdx.x * kBitTileWords + threadIdx.x]), s_wsum, total);
  if (threadIdx.x == 0) tile_cnt[(int64_t)f * tiles + blockIdx.x] = total;
}

// ... and their rows, in the reference's order (frame, then row-major pixel: tmp_pcl[tmp_st_mask] :247-251): tile (t, f)
// starts behind frame 0's rows and everything counted before it.
struct RowsArgs {
  const float *depths;  // [S][P]
  const float *rgbs;    // [S][P][3]
  float *cloud;
  float *xyz;
  int64_t capacity;
  int P, W;
};
__global__ void __launch_bounds__(kBitTileWords)
agg_rows_kernel(const uint32_t *__restrict__ sel, int64_t Wd, int tiles, const int32_t *__restrict__ tile_cnt,
                const int64_t *__restrict__ cnts, const CamBlock *__restrict__ cams, RowsArgs a,
                RowStage stage, int f_staged, uint4 *__restrict__ zero_extra, int zero_extra_n16,
                const int32_t *__restrict__ error, int64_t *__restrict__ count_out) {
  __shared__ uint16_t s_list[kSelTile];
  __shared__ int s_wsum[4];
  __shared__ long long s_before[4];
  __shared__ int s_cstart[kBitTileWords / 4];  // list position of every 128-pixel chunk's first selected pixel
  const int tid = threadIdx.x;
  // (round 6: the per-view call lets this launch clear the counters of the rasteriser that runs behind it -- one memset less
  // on the static branch's chain)
  if (zero_extra != nullptr) {
    const int g0 = ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * kBitTileWords + tid;
    for (int g = g0; g < zero_extra_n16; g += (int)(gridDim.x * gridDim.y) * kBitTileWords) zero_extra[g] = make_uint4(0u, 0u, 0u, 0u);
  }
  const int f = 1 + (int)blockIdx.y, t = (int)blockIdx.x;
  const uint32_t bits = sel[(int64_t)f * Wd + (int64_t)t * kBitTileWords + tid];
  int total;
  int slot = block_excl_256(__popc(bits), s_wsum, total);
  // (the last workgroup stays to leave the cloud's size)
  const bool last_wg = blockIdx.y == gridDim.y - 1 && blockIdx.x == gridDim.x - 1;
  if (total == 0 && !last_wg) return;
  if ((tid & 3) == 0) s_cstart[tid >> 2] = slot;
  for (uint32_t m = bits; m; m &= m - 1) s_list[slot++] = (uint16_t)((tid << 5) | __builtin_ctz(m));
  // Round 6: the selected pixels of the later frames before this (frame, tile), added up HERE from agg_count's per-tile counts
  // (entries tiles .. f * tiles + t - 1 of an L2-resident array: ~23 loads per thread at 1080p x 24 frames) -- until then a
  // one-workgroup scan launch between agg_count and this one wrote the running sums; the workgroups of tiles that selected
  // nothing have left by now.
  long long before = 0;
  for (int64_t u = (int64_t)tiles + tid; u < (int64_t)f * tiles + t; u += kBitTileWords) before += tile_cnt[u];
  for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
  if ((tid & 63) == 0) s_before[tid >> 6] = before;
  __syncthreads();
  before = (s_before[0] + s_before[1]) + (s_before[2] + s_before[3]);
  if (last_wg && tid == 0) {  // the count the caller sees (what agg_finalize_kernel computes for the ordered chain)
    const long long n = cnts[1] + before + total;
    *count_out = *error ? -1 : (n > a.capacity ? a.capacity : n);
  }
  if (total == 0) return;
  const int64_t pos0 = cnts[1] + before;
  // frames whose chain link unprojected its pixels (all but the last one) left their rows packed per 128-pixel chunk
  const bool from_stage = f < f_staged && stage.rows != nullptr;
  AppendSrc app;
  app.depth = a.depths + (size_t)f * (size_t)a.P;
  app.rgb = a.rgbs + (size_t)f * (size_t)a.P * 3;
  app.cloud = a.cloud;
  app.xyz = a.xyz;
  app.P = a.P;
  app.W = a.W;
  const CamBlock &cam = cams[f];
  const int tile_px = t * kSelTile;
#pragma unroll 4
  for (int e = tid; e < total; e += kBitTileWords) {
    const int64_t pos = pos0 + e;
    if (pos >= a.capacity) break;
    const int ent = (int)s_list[e];
    const int ch = ent >> 7;  // 128-pixel chunk of the tile (four threads of 32 pixels)
    if (from_stage) {
      // chunk g of the frame was listed by the link's workgroup `ob` as its chunk `orr` (step_owner_of)
      const int64_t g = (int64_t)t * (kSelTile / kStepChunkPx) + ch;
      int ob, orr;
      step_owner_of(g, stage.gx, ob, orr);
      const int64_t wg_slot = (int64_t)f * stage.gx + ob;
      const int first = stage.cst[wg_slot * kStepChunks + orr];
      const float4 q = *reinterpret_cast<const float4 *>(stage.rows + (wg_slot * (kStepThreads * kStepPx) + first + (e - s_cstart[ch])) * 4);
      const f3 c = {q.y, q.z, q.w};
      append_row(app, cam, tile_px + ent, pos, true, q.x, nullptr, &c);
    } else {
      append_row(app, cam, tile_px + ent, pos, true);
    }
  }
}

#ifdef PGDVS_AB_CHAIN
__global__ void agg_empty_kernel(int *p) {
  if (p != nullptr && threadIdx.x == 12345) *p = 0;
}
#endif
struct CamChunk {
  CamBlock c[12];
};
static_assert(sizeof(CamChunk) + 32 <= 4096, "agg_cams_kernel's arguments");
__global__ void agg_cams_kernel(CamChunk c, CamBlock *__restrict__ dst, int first, int n) {
  const int words = (int)(sizeof(CamBlock) / 4);
  for (int k = threadIdx.x; k < n * words; k += blockDim.x)
    reinterpret_cast<uint32_t *>(dst + first)[k] = reinterpret_cast<const uint32_t *>(&c.c[0])[k];
}

// the count the caller sees: the cloud size, or -1 when a look-back spin gave up in any frame
// (the cloud is then not trustworthy; hosts that read the count raise, see ops.static_aggregate)
// (total: the later frames' count behind frame 0's cnts[1]; null = cnts[S] as the ordered chain left it)
__global__ void agg_finalize_kernel(const int64_t *__restrict__ cnts, const int32_t *__restrict__ error,
                                    const int64_t *__restrict__ total, int S, int64_t capacity,
                                    int64_t *__restrict__ count_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const long long n = total != nullptr ? cnts[1] + *total : cnts[S];
    *count_out = *error ? -1 : (n > capacity ? capacity : n);
  }
}

struct AggWs {
  // state block zeroed once per call: cnts[S+1], ticket[S], error, desc[tiles]
  char *state;
  int64_t state_bytes;
  int64_t *cnts;
  int32_t *ticket, *error;
  unsigned *stat;  // [kAggStatWords] see ByteStamp::stat
  unsigned long long *desc;
  int32_t *tile_cnt;   // [S][tiles] pixels the later frames selected per (frame, tile)
  uint8_t *occ;  // [S][P] one occupancy byte per (frame, pixel), zeroed per call
  uint32_t *sel;  // [S][Wd] selection bits of the later frames (zeroed per call; a link writes the 128-pixel chunks that select)
  float *stage_rows;  // [S][gx][4096][4] (depth, colour) of the later frames' selected pixels as the chain links leave them (see RowStage)
  uint16_t *stage_cst;  // [S][gx][32] list position of every chunk's first selected pixel
  int step_gx;          // workgroups per link
  int64_t Wd;
  int32_t *sel_pix;  // [P] ordered chain: the pixels the current frame selected, in cloud order
  float *xyz;
  ProjF64 *proj;   // [S]
  PushConsts *pc32;  // [S + 2] fp32 screening records (two spare entries: see screen_frames)
  CamBlock *cams;  // [S]
  int64_t total_bytes;
};

static AggWs agg_ws_layout(void *base, int S, int H, int W, int64_t capacity) {
  AggWs w;
  const int64_t P = (int64_t)H * W;
  const int64_t tiles = cdiv(P, kSelTile);
  const int64_t tiles0 = tiles;  // one look-back granule per selection tile
  char *p = reinterpret_cast<char *>(base);
  int64_t off = 0;
  w.tile_cnt = reinterpret_cast<int32_t *>(p + off);
  off += align_up((int64_t)S * tiles * 4, 256);
  // the state block and, right behind it, the occupancy maps of frames 1 .. S-1 (frame 0 has none): ONE fill per call
  // clears both
  w.state = p + off;
  w.cnts = reinterpret_cast<int64_t *>(p + off);
  off += align_up((int64_t)(S + 1) * 8, 16);
  w.ticket = reinterpret_cast<int32_t *>(p + off);
  off += align_up((int64_t)S * 4, 16);
  w.error = reinterpret_cast<int32_t *>(p + off);
  off += 16;
  w.stat = reinterpret_cast<unsigned *>(p + off);
  off += align_up((int64_t)kAggStatWords * 4, 16);
  w.desc = reinterpret_cast<unsigned long long *>(p + off);
  off += align_up(tiles0 * 8, 16);
  off = align_up(off, 256);
  w.state_bytes = (p + off) - w.state;
  // occ[f][q] lives at occ + f * P + q for f >= 1; the pointer itself lies P bytes before the first map and is never
  // dereferenced for frame 0
  w.occ = reinterpret_cast<uint8_t *>(p + off) - P;
  off += align_up((int64_t)(S - 1) * P + 32, 256);
  w.Wd = tiles * kBitTileWords;
  w.sel = reinterpret_cast<uint32_t *>(p + off);
  off += align_up((int64_t)S * w.Wd * 4, 256);
  // (very long videos do without the staging block -- agg_rows then gathers depths and colours itself)
  w.stage_rows = nullptr;
  w.stage_cst = nullptr;
  w.step_gx = (int)align_up(cdiv(w.Wd * 32 / kStepChunkPx, kStepChunks), 8);
  if ((int64_t)S * w.step_gx * (kStepThreads * kStepPx) * 16 <= (4ll << 30)) {
    w.stage_rows = reinterpret_cast<float *>(p + off);
    off += align_up((int64_t)S * w.step_gx * (kStepThreads * kStepPx) * 16, 256);
    w.stage_cst = reinterpret_cast<uint16_t *>(p + off);
    off += align_up((int64_t)S * w.step_gx * kStepChunks * 2, 256);
  }
  w.xyz = reinterpret_cast<float *>(p + off);
  off += align_up((capacity > 0 ? capacity : 1) * 12, 256);
  w.sel_pix = reinterpret_cast<int32_t *>(p + off);
  off += align_up(P * 4, 256);
  w.proj = reinterpret_cast<ProjF64 *>(p + off);
  off += align_up((int64_t)S * (int64_t)sizeof(ProjF64), 256);
  w.pc32 = reinterpret_cast<PushConsts *>(p + off);
  off += align_up((int64_t)(S + 2) * (int64_t)sizeof(PushConsts), 256);
  w.cams = reinterpret_cast<CamBlock *>(p + off);
  off += align_up((int64_t)S * (int64_t)sizeof(CamBlock), 256);
  w.total_bytes = off;
  return w;
}

// the two statistics words the last aggregation on this workspace left behind (see ByteStamp::stat)
const unsigned *agg_stat_words(const void *workspace, int S, int H, int W, int64_t capacity) {
  const AggWs w = agg_ws_layout(const_cast<void *>(workspace), S, H, W, capacity);
  return w.stat;
}

}  // namespace pgdvs

using namespace pgdvs;

PGDVS_API int64_t pgdvs_static_aggregate_workspace_bytes(int S, int H, int W, int64_t capacity) {
  if (S <= 0 || H <= 0 || W <= 0 || capacity <= 0) return -1;
  return agg_ws_layout(nullptr, S, H, W, capacity).total_bytes;
}

static int static_aggregate_impl(const float *rgbs, const float *depths, const uint8_t *dyn_masks,
                                 const double *K3s_host, const double *c2ws_host, int S, int H, int W, float *out,
                                 float *xyz_out, int64_t capacity, int64_t *count_out, void *workspace,
                                 int64_t workspace_bytes, pgdvs_stream_t stream, bool params_cached = false,
                                 void *zero_extra = nullptr, int64_t zero_extra_bytes = 0);

PGDVS_API int pgdvs_static_aggregate(const float *rgbs, const float *depths,
                                     const uint8_t *dyn_masks, const double *K3s_host,
                                     const double *c2ws_host, int S, int H, int W, float *out,
                                     int64_t capacity, int64_t *count_out, void *workspace,
                                     int64_t workspace_bytes, pgdvs_stream_t stream) {
  return static_aggregate_impl(rgbs, depths, dyn_masks, K3s_host, c2ws_host, S, H, W, out, nullptr, capacity, count_out,
                               workspace, workspace_bytes, stream);
}

// The same aggregation; the packed coordinates the push launches work from ([capacity,3], 12 bytes per point)
// are kept in caller storage: the rasteriser's binning passes read them instead of the 24-byte rows.
PGDVS_API int pgdvs_static_aggregate_packed(const float *rgbs, const float *depths,
                                            const uint8_t *dyn_masks, const double *K3s_host,
                                            const double *c2ws_host, int S, int H, int W, float *out, float *xyz_out,
                                            int64_t capacity, int64_t *count_out, void *workspace,
                                            int64_t workspace_bytes, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(xyz_out, "pgdvs_static_aggregate_packed: null pointer");
  return static_aggregate_impl(rgbs, depths, dyn_masks, K3s_host, c2ws_host, S, H, W, out, xyz_out, capacity, count_out,
                               workspace, workspace_bytes, stream);
}

namespace pgdvs {
// the view-level entry point (view_geo.cpp): the same aggregation; params_cached = the workspace still holds the per-frame
// constants of the previous call with the same cameras (five parameter uploads less per view)
int static_aggregate_for_view(const float *rgbs, const float *depths, const uint8_t *dyn_masks, const double *K3s_host,
                              const double *c2ws_host, int S, int H, int W, float *out, float *xyz_out, int64_t capacity,
                              int64_t *count_out, void *workspace, int64_t workspace_bytes, pgdvs_stream_t stream,
                              bool params_cached, void *zero_extra, int64_t zero_extra_bytes) {
  return static_aggregate_impl(rgbs, depths, dyn_masks, K3s_host, c2ws_host, S, H, W, out, xyz_out, capacity, count_out, workspace,
                               workspace_bytes, stream, params_cached, zero_extra, zero_extra_bytes);
}
}  // namespace pgdvs

static int static_aggregate_impl(const float *rgbs, const float *depths, const uint8_t *dyn_masks,
                                 const double *K3s_host, const double *c2ws_host, int S, int H, int W, float *out,
                                 float *xyz_out, int64_t capacity, int64_t *count_out, void *workspace,
                                 int64_t workspace_bytes, pgdvs_stream_t stream, bool params_cached, void *zero_extra,
                                 int64_t zero_extra_bytes) {
  PGDVS_REQUIRE(rgbs && depths && dyn_masks && K3s_host && c2ws_host && out && count_out,
                "pgdvs_static_aggregate: null pointer");
  PGDVS_REQUIRE(S > 0 && S < 65535 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 31) && capacity > 0 &&
                    capacity < (1ll << 45),
                "pgdvs_static_aggregate: bad shape");
  AggWs ws = agg_ws_layout(workspace, S, H, W, capacity);
  if (!workspace || workspace_bytes < ws.total_bytes) {
    set_error("pgdvs_static_aggregate: workspace too small");
    return PGDVS_ERR_WORKSPACE;
  }
  if (xyz_out) ws.xyz = xyz_out;
  hipStream_t st = as_stream(stream);
  const int64_t P = (int64_t)H * W;
  // counts, tickets, error / statistics words, look-back granules and the later frames' maps: one fill
  // (... and the later frames' selection bits, which lie behind the maps: the links only write the chunks that select)
  hipError_t e = fill_async(ws.state, 0, (size_t)(reinterpret_cast<char *>(ws.sel) - ws.state) + (size_t)S * (size_t)ws.Wd * 4, st);
  if (e != hipSuccess) {
    set_error("static_aggregate memset: %s", hipGetErrorString(e));
    return PGDVS_ERR_LAUNCH;
  }
  // ---- per-frame constants: projection (fp64, device-resident) and ray setup (fp32, by value)
  std::vector<CamBlock> cams((size_t)S);
  {
    ProjChunk chunk;
    for (int i = 0; i < S; ++i) {
      const double *K3 = K3s_host + (size_t)i * 9;
      const double *c2w = c2ws_host + (size_t)i * 16;
      ProjF64 &pj = chunk.p[i % kProjChunk];
      for (int k = 0; k < 9; ++k) pj.K3[k] = K3[k];
      if (inv_f64(c2w, pj.w2c, 4) != 0) {
        set_error("pgdvs_static_aggregate: singular c2w for frame %d", i);
        return PGDVS_ERR_INVALID;
      }
      pj.affine = pj.w2c[12] == 0.0 && pj.w2c[13] == 0.0 && pj.w2c[14] == 0.0 && pj.w2c[15] == 1.0;
      pj.ksparse = K3[1] == 0.0 && K3[3] == 0.0 && K3[6] == 0.0 && K3[7] == 0.0 && K3[8] == 1.0;
      {
        double rmax = 0.0;
        bool finite = true;
        for (int r = 0; r < 3; ++r) {
          double rowsum = 0.0;
          for (int c = 0; c < 4; ++c) {
            double m = 0.0, a = 0.0;
            for (int k = 0; k < 3; ++k) {
              m += K3[r * 3 + k] * pj.w2c[k * 4 + c];
              a += fabs(K3[r * 3 + k]) * fabs(pj.w2c[k * 4 + c]);
            }
            pj.M[r * 4 + c] = m;
            rowsum += a;
            finite = finite && std::isfinite(m) && std::isfinite(a);
          }
          rmax = rowsum > rmax ? rowsum : rmax;
        }
        pj.tol_scale = 64.0 * 1.1102230246251565e-16 * rmax;
        pj.screen = pj.affine && finite && rmax > 0.0;
        // fp32 form: entries rounded to fp32 (2^-24 relative each) + three fused multiply-adds whose
        // partial sums are bounded by the row sum x amax (3 x 2^-24): 4 x 2^-24 in total, taken as 4.5;
        // the fp64 form's own distance to the reference order (tol_scale) is folded in, and the
        // float conversions round up
        bool ok32 = pj.screen != 0;
        for (int k = 0; k < 12; ++k) {
          // k = row * 4 + column -> interleaved position of rows 0 / 1, row 2 behind them
          const int pos = k < 8 ? (k & 3) * 2 + (k >> 2) : k;
          pj.M32[pos] = (float)pj.M[k];
          ok32 = ok32 && std::isfinite(pj.M32[pos]);
        }
        for (int r = 0; r < 3; ++r) {
          double rowsum = 0.0;
          for (int c = 0; c < 4; ++c) rowsum += fabs(pj.M[r * 4 + c]);
          const double e = (4.5 * 5.9604644775390625e-08 * rowsum + pj.tol_scale) * 1.000001;
          pj.e32[r] = nextafterf((float)e, INFINITY);
          ok32 = ok32 && std::isfinite(pj.e32[r]);
        }
        pj.screen32 = ok32 ? 1 : 0;
      }
      // rays use K and c2w cast to fp32 (torch.FloatTensor, nvidia_eval.py:841-842)
      float flat[34];
      flat[0] = (float)H;
      flat[1] = (float)W;
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
          flat[2 + r * 4 + c] = (r < 3 && c < 3) ? (float)K3[r * 3 + c] : (r == c ? 1.0f : 0.0f);
      for (int k = 0; k < 16; ++k) flat[18 + k] = (float)c2w[k];
      if (cam_block_from_flat(flat, cams[(size_t)i].v) != 0) {
        set_error("pgdvs_static_aggregate: singular intrinsics for frame %d", i);
        return PGDVS_ERR_INVALID;
      }
      if (!params_cached && (i % kProjChunk == kProjChunk - 1 || i == S - 1)) {
        const int first = i - i % kProjChunk, cnt = i % kProjChunk + 1;
        PGDVS_LAUNCH("agg_params", agg_params_kernel, dim3(1), dim3(256), 0, st, chunk, ws.proj, reinterpret_cast<float *>(ws.pc32), first,
                     cnt, (float)((W > H ? W : H) + 1));
      }
    }
  }
  const int tiles = (int)cdiv(P, kSelTile);
  // option agg_ordered (PGDVS_AGG_ORDERED=1 at load time): round 2's chain for every frame (ordered selection + byte-stamping push that builds the rows),
  // kept as a second implementation that the tests run the same bit-exact cases through
  const bool bit_chain = option_int(options().agg_ordered) == 0 && S > 1;
  auto select = [&](int i) {
    SelArgs a;
    a.dyn_mask = dyn_masks + (size_t)i * P;
    a.occ = ws.occ + (size_t)i * (size_t)P;
    a.depth = depths + (size_t)i * P;
    a.rgb = rgbs + (size_t)i * P * 3;
    a.cloud = out;
    a.xyz = ws.xyz;
    a.cnts = ws.cnts;
    a.desc = ws.desc;
    a.ticket = ws.ticket;
    a.error = ws.error;
    a.sel_pix = i > 0 ? ws.sel_pix : nullptr;
    a.capacity = capacity;
    a.frame = i;
    a.P = (int)P;
    a.W = W;
    a.tiles = tiles;
    PGDVS_LAUNCH("agg_select", agg_select_kernel, dim3(tiles), dim3(kSelThreads), 0, st, a, cams[(size_t)i]);
  };
  auto frame_src = [&](int i) {
    AppendSrc app;
    app.depth = depths + (size_t)i * P;
    app.rgb = rgbs + (size_t)i * P * 3;
    app.cloud = out;
    app.xyz = ws.xyz;
    app.P = (int)P;
    app.W = W;
    return app;
  };
  // cnts[i] = points in the cloud before frame i.  Frame 0 appends every static pixel (~P points x S-1 frames: the
  // chip-filling push launch, 12 frames per workgroup row so that the points are read twice).
  // (round 4, three lanes, 60-view runs on one box: 8 frames per row 1160 / 1168 frames/s in steady state, 6: 1155 / 1152,
  // 12 -- two rows, the coordinates read twice instead of three times --: 1171 / 1184)
  const int fpg = 12;
  auto push = [&](int i) {
    // (ordered chain: later frames append a few per cent of P and their launches are latency chains, shorter with 4
    // frames per row; 256 workgroups walk whatever there is -- the count is device-side)
    const int fpg_i = i == 0 ? fpg : (fpg > 4 ? 4 : fpg);
    const int groups = i + 1 < S ? (int)cdiv(S - 1 - i, fpg_i) : 1;
    const int64_t want = i == 0 ? cdiv(P, kPushThreads) : (cdiv(P, 8 * kPushThreads) < 256 ? cdiv(P, 8 * kPushThreads) : 256);
    const unsigned gx = (unsigned)(want < 1024 ? (want > 8 ? (want + 7) / 8 * 8 : 8) : 1024);  // a multiple of 8: one share per XCD
    if (i == 0) {
      PGDVS_LAUNCH("agg_push0", agg_push_kernel<kPushQueue>, dim3(gx, (unsigned)groups), dim3(kPushThreads), 0, st,
                   (const float *)ws.xyz, (const int64_t *)ws.cnts, i, (const ProjF64 *)ws.proj, (const PushConsts *)ws.pc32, i + 1, S, fpg_i, H, W,
                   ws.occ,
                   (const int32_t *)nullptr, frame_src(i), cams[(size_t)i], ws.stat);
    } else {
      PGDVS_LAUNCH("agg_push", agg_push_kernel<kPushQueueSmall>, dim3(gx, (unsigned)groups), dim3(kPushThreads), 0, st,
                   (const float *)ws.xyz, (const int64_t *)ws.cnts, i, (const ProjF64 *)ws.proj, (const PushConsts *)ws.pc32, i + 1, S, fpg_i, H, W,
                   ws.occ,
                   (const int32_t *)ws.sel_pix, frame_src(i), cams[(size_t)i], ws.stat);
    }
  };
  if (!bit_chain) {
    if (zero_extra != nullptr && zero_extra_bytes > 0) (void)hipMemsetAsync(zero_extra, 0, (size_t)zero_extra_bytes, st);
    for (int i = 0; i < S; ++i) {
      select(i);
      // frame 0 appended its rows itself and only stamps; a later frame's push also builds the rows of the
      // pixels its selection listed -- for the last frame that is all it does
      if (i + 1 < S || i > 0) push(i);
    }
    PGDVS_LAUNCH("agg_finalize", agg_finalize_kernel, dim3(1), dim3(64), 0, st, (const int64_t *)ws.cnts,
                 (const int32_t *)ws.error, (const int64_t *)nullptr, S, capacity, count_out);
    return check_launch("static_aggregate");
  }
  if (!params_cached) {
    CamChunk chunk;
    const int per = (int)(sizeof(chunk.c) / sizeof(chunk.c[0]));
    for (int i = 1; i < S; ++i) {  // (frame 0's block travels by value)
      chunk.c[(i - 1) % per] = cams[(size_t)i];
      if ((i - 1) % per == per - 1 || i == S - 1) {
        const int cnt = (i - 1) % per + 1;
        PGDVS_LAUNCH("agg_cams", agg_cams_kernel, dim3(1), dim3(256), 0, st, chunk, ws.cams, i - cnt + 1, cnt);
      }
    }
  }
  // (round 4: frame 0 as ONE fused launch -- selection, ordered offsets, projections and rows -- was built twice and took
  // 110-114 us against 29 + 61 for this pair)
  select(0);
  push(0);
  // option agg_stage = 0: the links leave nothing behind, agg_rows gathers depth and colour itself -- the path very long
  // videos take anyway (no staging block in their workspace)
  RowStage stage;
  stage.rows = option_int(options().agg_stage) != 0 ? ws.stage_rows : nullptr;
  stage.cst = ws.stage_cst;
  stage.gx = ws.step_gx;
  {
    // 32 chunks of 128 pixels per workgroup, dealt round-robin; a multiple of 8 workgroups per row (see the kernel)
    const unsigned gx = (unsigned)ws.step_gx;
    // frames per workgroup row.  Alone on the chip a link takes 9.9 / 8.4 / 7.9 / 7.6 / 7.6 / 8.2 us with 2 / 3 / 4 / 6 / 8 / 16 frames per
    // row (more rows = more parallel frames), but every row re-reads the chunk and re-gathers the depths, and with seven views
    // in flight the throughput is the other way round: 1037 frames/s with 6, 1048 with 8, 1055 with 16, 1058 with 24-32 -- one
    // row whenever the later frames fit a queue entry's mask
    const int sfpg = kPushMaxFpg;
#ifdef PGDVS_AB_CHAIN
    // tools/r05_chain_cost.sh: what the chain costs the throughput, by ADDING copies of it behind the real one (a link is
    // idempotent: its own map is complete before it runs, so a second run selects, stamps and stages exactly the same)
    const char *dbg = getenv("PGDVS_DBG_CHAIN");
    const int dbg_mode = !dbg ? 0 : !strcmp(dbg, "dup") ? 1 : !strcmp(dbg, "dup_dry") ? 2 : !strcmp(dbg, "dup_empty") ? 3 : !strcmp(dbg, "dup_small") ? 4 : 0;
    for (int pass = 0; pass < (dbg_mode ? 2 : 1); ++pass)
#endif
    for (int i = 1; i < S; ++i) {
      SelArgs a;
      a.dyn_mask = dyn_masks + (size_t)i * P;
      a.occ = ws.occ + (size_t)i * (size_t)P;
      a.frame = i;
      a.P = (int)P;
      a.W = W;
      const unsigned gy = i + 1 < S ? (unsigned)cdiv(S - 1 - i, sfpg) : 1u;
#ifdef PGDVS_AB_CHAIN
      if (pass == 1 && dbg_mode == 3) {
        PGDVS_LAUNCH("agg_empty", agg_empty_kernel, dim3(1), dim3(64), 0, st, (int *)nullptr);
        continue;
      }
      if (pass == 1 && dbg_mode == 4) {  // launch boundaries + the workgroups, which leave at once (frame index beyond the video)
        PGDVS_LAUNCH("agg_empty", agg_empty_kernel, dim3(gx), dim3(kStepThreads), 0, st, (int *)nullptr);
        continue;
      }
      if (pass == 1 && dbg_mode == 2) {
        RowStage none;
        none.rows = nullptr;
        none.cst = stage.cst;
        none.gx = stage.gx;
        PGDVS_LAUNCH("agg_step_dry", agg_step_kernel<kPushQueueSmall>, dim3(gx, gy), dim3(kStepThreads), 0, st, a, (uint8_t *)nullptr,
                     (uint16_t *)nullptr, ws.Wd, i, (const ProjF64 *)ws.proj, (const PushConsts *)ws.pc32, S, sfpg, H, W,
                     frame_src(i), cams[(size_t)i], ws.stat, none);
        continue;
      }
#endif
      PGDVS_LAUNCH("agg_step", agg_step_kernel<kPushQueueSmall>, dim3(gx, gy), dim3(kStepThreads), 0, st, a, ws.occ,
                   reinterpret_cast<uint16_t *>(ws.sel), ws.Wd, i, (const ProjF64 *)ws.proj, (const PushConsts *)ws.pc32, S, sfpg, H, W,
                   frame_src(i), cams[(size_t)i], ws.stat, stage);
    }
  }
  PGDVS_LAUNCH("agg_count", agg_count_kernel, dim3((unsigned)tiles, (unsigned)(S - 1)), dim3(kBitTileWords), 0, st,
               (const uint32_t *)ws.sel, ws.Wd, tiles, ws.tile_cnt);
  {
    RowsArgs ra;
    ra.depths = depths;
    ra.rgbs = rgbs;
    ra.cloud = out;
    ra.xyz = ws.xyz;
    ra.capacity = capacity;
    ra.P = (int)P;
    ra.W = W;
    PGDVS_LAUNCH("agg_rows", agg_rows_kernel, dim3((unsigned)tiles, (unsigned)(S - 1)), dim3(kBitTileWords), 0, st,
                 (const uint32_t *)ws.sel, ws.Wd, tiles, (const int32_t *)ws.tile_cnt, (const int64_t *)ws.cnts,
                 (const CamBlock *)ws.cams, ra, stage, stage.rows != nullptr ? S - 1 : 0, reinterpret_cast<uint4 *>(zero_extra),
                 (int)(zero_extra_bytes / 16), (const int32_t *)ws.error, count_out);
  }
  return check_launch("static_aggregate");
}
