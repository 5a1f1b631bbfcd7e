// A4 (fast path): EXACT k-nearest-neighbour mean distance through a uniform 3-D grid.
//
// pytorch3d.ops.knn_points (the reference's dependency, pgdvs_renderer_dyn.py:405-410) is a
// brute-force O(N^2) scan.  The statistic the renderer consumes -- the mean of the K smallest
// non-self squared distances -- only needs the K+1 smallest distance VALUES of each point,
// which a spatial grid finds exactly: a query scans the cells of a growing cube around its
// own cell and stops as soon as its (K+1)-th best distance is no larger than the distance to
// the cube's nearest open face (minus a rounding margin); nothing outside the cube can then
// change the multiset of the K+1 smallest values.  Queries that do not terminate within
// kRingCap rings (isolated outliers) repeat the search on a second-level grid with
// kCoarseScale-times larger cells; the few still open after that are scanned exhaustively.
// The first-level grid's cell -> points index is sparse (CellIndex: a bit per cell, ranks per 64 cells, starts of
// the occupied cells only): a depth-map cloud occupies ~0.15 % of the cells of its bounding box.  Its cell size
// comes from the cloud's own spacing (grid_params_kernel: the median distance between the points of a thinned copy),
// and the points of a cell are kept in the order of their quarter along x (fine_coord), so that a query can cut
// the x-runs it scans to its ball.
//
// Self-queries first go through a THREAD-per-query pass over the 3x3x3 block of their cell
// (grid_query_tpq_kernel, below); what that leaves open, and cross-set queries, use:
// One WAVEFRONT per query: the 64 lanes evaluate 64 candidates per step with coalesced
// 16-byte loads of the cell-sorted point array; the sorted best-list lives one value per
// lane in a register (lane k holds the k-th smallest), so an insertion is a ballot, one DPP
// shift and one median-of-three -- no LDS, no per-lane divergence.  Values are summed in a
// fixed order shared with the CPU oracle => bit-identical averages.
#include <stdlib.h>

#include "common.h"
#include "fused.h"

namespace pgdvs {

constexpr int kGridMaxCells = 1 << 24;
constexpr int kRingCap = 24;
// After the thread-per-query pass the ring search only sees the sparse remainder, whose long
// searches are no longer hidden behind the bulk: hand them to the coarse grid after a few rings.
// (round 5, noisy-depth scene, 28 k open queries: 4 or 6 rings change nothing -- 591 / 585 frames/s against 591)
constexpr int kRingCapAfterTpq = 3;
// Points per cell the FALLBACK sizing aims at for K + 1 = 51 (scaled with K + 1) -- clouds too small for the sample-based
// sizing below (grid_params_kernel), cells from the bounding box as if the points covered its two largest extents.
// (Rounds 1-4 sized every grid through this target and a trial grid; the record of that tuning -- the benchmark's 311 k-point
// cloud: the thread-per-query pass 223 us at h = 4.6e-3, 195 at 4.1e-3, 185 at 4.0e-3, 180 at 3.5e-3 where the ring search
// grows; round 5, rows cut along x: 158 us at 4.0e-3, 174 / 162 / 169 at targets 15 / 20 / 22 -- is what
// kCellPerSampleDist reproduces.)
constexpr float kTargetPerCellDefault = 18.0f;

struct GridParams {
  float mn[3];
  float h, inv_h;
  int G[3];
  int ncells;
  int n;
  float sample_dist;  // diagnostics: the quantile of the nearest-sample distances behind h, and how many were measured
  int sample_dists;
};

// bbox[0..2] = ~min (ordered-uint, complemented so that an all-zero block is the empty box), bbox[3..5] = max.  Launched with few blocks: six
// atomics per block.
__global__ void __launch_bounds__(256)
grid_bbox_kernel(const float *__restrict__ pts, const int32_t *__restrict__ count,
                 unsigned *__restrict__ bbox) {
  __shared__ float s_mn[4][3], s_mx[4][3];
  const int n = *count;
  float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float v = pts[(size_t)i * 3 + a];
      if (isfinite(v)) {
        mn[a] = fminf(mn[a], v);
        mx[a] = fmaxf(mx[a], v);
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    for (int off = 32; off > 0; off >>= 1) {
      mn[a] = fminf(mn[a], __shfl_down(mn[a], off, 64));
      mx[a] = fmaxf(mx[a], __shfl_down(mx[a], off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      s_mn[threadIdx.x >> 6][a] = mn[a];
      s_mx[threadIdx.x >> 6][a] = mx[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    int a = threadIdx.x;
    float lo = fminf(fminf(s_mn[0][a], s_mn[1][a]), fminf(s_mn[2][a], s_mn[3][a]));
    float hi = fmaxf(fmaxf(s_mx[0][a], s_mx[1][a]), fmaxf(s_mx[2][a], s_mx[3][a]));
    atomicMax(&bbox[a], ~f2ord(lo));
    atomicMax(&bbox[3 + a], f2ord(hi));
  }
}

// ---- cell size from the cloud's own spacing (round 5) -----------------------------------------------------------------
// Rounds 1-4 sized the cells through a TRIAL grid: cells from the bounding box under the assumption that the points
// sample a surface spanning its two largest extents, the occupancy of those cells measured, and the cell size rescaled
// as for a surface (occupancy ~ h^2).  The bounding box of a frame's dynamic content says little about its area (the
// benchmark's trial cells held 1300 points instead of the 72 aimed at), and over that range the scaling law matters: a
// cloud from noisy depth is a slab several cells thick (occupancy ~ h^2.2 at the trial's scale, ~ h^3 at the final one),
// its cells came out with half the points aimed at, and up to a quarter of its queries could not be certified by their
// 3 x 3 x 3 block.  Now: a pseudo-random 64th of the points (hash of the index: a Poisson-like thinning, whatever order
// the cloud arrives in), and for a few hundred of those the distance d1 to the nearest OTHER sample.  The ball of that
// radius around a sample holds ~64 points of the cloud whatever the cloud's local dimension, so the (K+1 = 51)-ball the
// search has to cover has radius ~1.06 x median(d1) for a surface and for a volume alike (2-D: r51 = 0.893 r64, median
// d1 = 0.833 r64; 3-D: 0.927 and 0.885), and the cell is a fixed multiple of it.  One launch fewer than the trial grid,
// no 8 MB of trial counters to clear and read.
constexpr int kSampleShift = 6;           // one point in 64
constexpr int kSampleBlock = 4096;        // points per block of the sampling pass
constexpr int kSampleSlots = 128;         // samples kept per block (64 expected, binomial: sd 8)
constexpr int kSampleMaxQueries = 4096;   // samples whose nearest-sample distance is measured
constexpr int kSampleWindowBlocks = 256;  // ... among the samples of this many blocks around their own (a million points)
// cell size in units of the median nearest-sample distance for K + 1 = 51 (scaled with sqrt((K + 1) / 51) otherwise).
// 1.5 reproduces the cell size the trial grid's tuned target gave on the benchmark's cloud (h = 4.0e-3: the pass takes
// 168 us at 1.4, 159 at 1.5, 170 at 1.65; 400-650 of its 311 k queries go on to the ring search instead of 1900) and gives
// the noisy-depth scene h = 9.0e-3 (12-14 k of 311 k to the ring search; rounds 3-4: 6.3e-3 - 8.2e-3 by view, 27-77 k).
// Nominal distances on the benchmark's cloud: quartiles 2.7e-3 / 3.9e-3, 90 % 4.8e-3 (ratios of a Poisson sample of a
// surface: 1.41, 1.82); noisy: 6.0e-3 / 7.8e-3 / 9.7e-3 (between those of a surface and of a volume: 1.26, 1.49).
constexpr float kCellPerSampleDist = 1.5f;

__device__ __forceinline__ bool is_sample(unsigned i) {
  unsigned x = i * 0x9E3779B1u;  // (a mixing hash of the index; the low bits of a product alone are regular)
  x ^= x >> 15;
  x *= 0x85EBCA77u;
  x ^= x >> 13;
  return (x >> (32 - kSampleShift)) == 0u;
}

// samples of block b (points [b * kSampleBlock, ...)) in index order -> samples[b * kSampleSlots + k] = (x, y, z, index),
// sample_count[b]: deterministic whatever the scheduling (the cell size must not change from run to run)
__global__ void __launch_bounds__(256)
grid_sample_kernel(const float *__restrict__ pts, const int32_t *__restrict__ count, float4 *__restrict__ samples,
                   int32_t *__restrict__ sample_count) {
  __shared__ int s_w[4];
  const int n = *count;
  const int b = blockIdx.x;
  if ((long long)b * kSampleBlock >= n) return;
  constexpr int kPer = kSampleBlock / 256;
  const int i0 = b * kSampleBlock + threadIdx.x * kPer;
  unsigned hit = 0;
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int i = i0 + k;
    if (i < n && is_sample((unsigned)i)) {
      const float x = pts[(size_t)i * 3], y = pts[(size_t)i * 3 + 1], z = pts[(size_t)i * 3 + 2];
      if (isfinite(x) && isfinite(y) && isfinite(z)) hit |= 1u << k;
    }
  }
  const int mine = __popc(hit);
  int incl = mine;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    const int y = __shfl_up(incl, off, 64);
    if (lane >= off) incl += y;
  }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int at = incl - mine;
  for (int w = 0; w < wave; ++w) at += s_w[w];
  for (unsigned m = hit; m; m &= m - 1) {
    const int i = i0 + __builtin_ctz(m);
    if (at < kSampleSlots) samples[(size_t)b * kSampleSlots + at] = make_float4(pts[(size_t)i * 3], pts[(size_t)i * 3 + 1], pts[(size_t)i * 3 + 2], __int_as_float(i));
    ++at;
  }
  const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  // the block's unused slots: a point at infinity (the distance pass reads every slot without looking at the count)
  if ((int)threadIdx.x < kSampleSlots && (int)threadIdx.x >= tot)
    samples[(size_t)b * kSampleSlots + threadIdx.x] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __int_as_float(-1));
  if (threadIdx.x == 255) sample_count[b] = tot < kSampleSlots ? tot : kSampleSlots;
}

// squared distance from query sample (block qb, slot qk) to its nearest other sample; one workgroup per query, every
// thread a share of the slots (unused ones hold a point at infinity).  Queries: the first four slots of every
// `stride`-th block (stride from the device-side count: at most kSampleMaxQueries)
__global__ void __launch_bounds__(256)
grid_sample_nn_kernel(const int32_t *__restrict__ count, const float4 *__restrict__ samples,
                      const int32_t *__restrict__ sample_count, float *__restrict__ d1_out) {
  __shared__ float s_min[4];
  const int n = *count;
  const int nblocks = (int)(((long long)n + kSampleBlock - 1) / kSampleBlock);
  const int stride = (nblocks * 4 + kSampleMaxQueries - 1) / kSampleMaxQueries;  // >= 1 when there is a block
  const int q = blockIdx.x;  // query number
  const int qb = (q >> 2) * (stride > 0 ? stride : 1), qk = q & 3;
  if (qb >= nblocks || qk >= sample_count[qb]) {  // (uniform over the workgroup)
    if (threadIdx.x == 0) d1_out[q] = -1.0f;  // "no measurement"
    return;
  }
  const int self = qb * kSampleSlots + qk;
  const float4 me = samples[self];
  // the samples of at most kSampleWindowBlocks blocks around the query's own (round 6, after the advisor's finding: every query
  // scanned every slot -- 2 G distance evaluations for a 16 M-point cloud, just to pick a cell size).  Clouds of up to a million
  // points are scanned in full as before; beyond that the window holds the query's neighbours whenever the cloud arrives in an
  // order with spatial locality (depth maps in raster order), and otherwise the distance -- and with it the cell size -- comes out
  // larger: slower, never wrong.
  int b_lo = 0, b_hi = nblocks;
  if (nblocks > kSampleWindowBlocks) {
    b_lo = qb - kSampleWindowBlocks / 2;
    b_lo = b_lo < 0 ? 0 : (b_lo > nblocks - kSampleWindowBlocks ? nblocks - kSampleWindowBlocks : b_lo);
    b_hi = b_lo + kSampleWindowBlocks;
  }
  const int s_lo = b_lo * kSampleSlots, nslots = b_hi * kSampleSlots;
  float best = __builtin_inff();
  for (int k0 = s_lo + threadIdx.x; k0 < nslots; k0 += 4 * 256) {
    float4 p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u * 256;
      p[u] = samples[k < nslots ? k : self];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float dx = p[u].x - me.x, dy = p[u].y - me.y, dz = p[u].z - me.z;
      const float d = dx * dx + dy * dy + dz * dz;  // (inf for an unused slot; the query itself is skipped by its slot)
      const int k = k0 + u * 256;
      best = (k != self && k < nslots && d < best) ? d : best;
    }
  }
  for (int off = 32; off > 0; off >>= 1) best = fminf(best, __shfl_xor(best, off, 64));
  if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    best = fminf(fminf(s_min[0], s_min[1]), fminf(s_min[2], s_min[3]));
    d1_out[q] = best < __builtin_inff() ? best : -1.0f;
  }
}

// The grid: origin and extent from the bounding box, cell size h = kCellPerSampleDist x sqrt((K + 1) / 51) x the median
// nearest-sample distance (positive measurements only; fewer than eight of them -- clouds below ~1000 points, or all
// points in one place -- : cells from the bounding box as if the points covered its two largest extents evenly).
// One workgroup of 1024 threads: the median through two 1024-bin histograms of the distances' bit patterns (positive
// floats order like their bits): 11 bits, then 10 more inside the median's bin -- 2^-13 relative.
__global__ void __launch_bounds__(1024)
grid_params_kernel(const unsigned *__restrict__ bbox, const int32_t *__restrict__ count,
                   const float *__restrict__ d1, GridParams *__restrict__ gp, float k_scale, float target, float rank_frac) {
  __shared__ int s_hist[1024];
  __shared__ int s_sel[3];  // chosen bin, entries below it, valid measurements
  __shared__ int s_wtot[16];
  const int tid = threadIdx.x;
  if (tid == 0) s_sel[0] = 0;
  unsigned bits[kSampleMaxQueries / 1024];
  int valid = 0;
#pragma unroll
  for (int k = 0; k < kSampleMaxQueries / 1024; ++k) {
    const float v = d1[tid + k * 1024];
    bits[k] = (v > 0.0f && v < __builtin_inff()) ? __float_as_uint(v) : 0u;
    valid += bits[k] != 0u;
  }
  // (two passes over the same machinery: level 0 bins by bits >> 21, level 1 by (bits >> 11) & 1023 inside the chosen bin)
  unsigned chosen = 0u;
  int below = 0, total = 0;
  for (int level = 0; level < 2; ++level) {
    s_hist[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSampleMaxQueries / 1024; ++k) {
      if (bits[k] == 0u) continue;
      if (level == 0) {
        atomicAdd(&s_hist[bits[k] >> 21], 1);
      } else if ((bits[k] >> 21) == chosen) {
        atomicAdd(&s_hist[(bits[k] >> 11) & 1023u], 1);
      }
    }
    __syncthreads();
    {
      // bin tid: entries below it (scan over the 1024 bins: wavefront scan + the 16 wavefront totals)
      const int mine = s_hist[tid];
      int incl = mine;
      for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(incl, off, 64);
        if ((tid & 63) >= off) incl += y;
      }
      if ((tid & 63) == 63) s_wtot[tid >> 6] = incl;
      __syncthreads();
      int ex = incl - mine, all = 0;
      for (int w = 0; w < 16; ++w) {
        ex += w < (tid >> 6) ? s_wtot[w] : 0;
        all += s_wtot[w];
      }
      if (level == 0 && tid == 0) {
        s_sel[2] = all;
        s_sel[1] = 0;
      }
      __syncthreads();
      const int want = (int)((float)s_sel[2] * rank_frac) - s_sel[1];  // rank of the quantile inside what is being binned
      __syncthreads();
      if (mine > 0 && ex <= want && want < ex + mine) {  // (one bin at most; none when nothing was measured)
        s_sel[0] = tid;
        s_sel[1] += ex;
      }
    }
    __syncthreads();
    if (level == 0) {
      chosen = (unsigned)s_sel[0];
      total = s_sel[2];
    } else {
      chosen = (chosen << 10) | (unsigned)s_sel[0];
    }
    below = s_sel[1];
    __syncthreads();
  }
  (void)below;
  (void)valid;
  if (tid != 0) return;
  const int n = *count;
  float mn[3], ext[3];
  for (int a = 0; a < 3; ++a) {
    float lo = ord2f(~bbox[a]), hi = ord2f(bbox[3 + a]);
    if (!(hi >= lo)) {  // no finite point
      lo = 0.0f;
      hi = 0.0f;
    }
    mn[a] = lo;
    ext[a] = hi - lo;
  }
  float h;
  if (total >= 8) {
    const float d1_med = sqrtf(__uint_as_float((chosen << 11) | 0x400u));  // (the bin's midpoint)
    h = kCellPerSampleDist * k_scale * d1_med;
  } else {
    float e0 = ext[0], e1 = ext[1], e2 = ext[2];
    float big = fmaxf(e0, fmaxf(e1, e2));
    float small = fminf(e0, fminf(e1, e2));
    float mid = e0 + e1 + e2 - big - small;
    float area = big * fmaxf(mid, 1e-3f * big);
    h = sqrtf(area * target / (float)(n > 0 ? n : 1));
  }
  if (!(h > 0.0f) || !isfinite(h)) h = 1.0f;
  gp->sample_dist = total >= 8 ? sqrtf(__uint_as_float((chosen << 11) | 0x400u)) : 0.0f;
  gp->sample_dists = total;
  int G[3];
  for (int it = 0; it < 64; ++it) {
    long long tot = 1;
    bool ok = true;
    for (int a = 0; a < 3; ++a) {
      float g = floorf(ext[a] / h) + 1.0f;
      if (!(g < 1024.0f)) ok = false;
      G[a] = g < 1.0f ? 1 : (g > 1024.0f ? 1024 : (int)g);
      tot *= G[a];
    }
    if (ok && tot <= kGridMaxCells) break;
    h *= 1.3f;
  }
  gp->h = h;
  gp->inv_h = 1.0f / h;
  gp->ncells = G[0] * G[1] * G[2];
  gp->n = n;
  for (int a = 0; a < 3; ++a) {
    gp->mn[a] = mn[a];
    gp->G[a] = G[a];
  }
}

// Second-level grid for the queries the first level gives up on: same bounding box, cells
// `scale` times larger (so the same number of rings reaches `scale` times farther), capped at
// max_cells.
__device__ __forceinline__ GridParams coarse_params(const GridParams *__restrict__ fine, const unsigned *__restrict__ bbox,
                                                    float scale, float scale_many, const int32_t *__restrict__ open_count,
                                                    const int32_t *__restrict__ qcount, int max_cells) {
  GridParams gp;
  // (qcount: the queries of a cross-set search; self-queries: the points)
  if ((long long)*open_count * 100 > (long long)(qcount ? *qcount : fine->n)) scale = scale_many;
  float ext[3];
  for (int a = 0; a < 3; ++a) {
    float lo = ord2f(~bbox[a]), hi = ord2f(bbox[3 + a]);
    if (!(hi >= lo)) {
      lo = 0.0f;
      hi = 0.0f;
    }
    gp.mn[a] = lo;
    ext[a] = hi - lo;
  }
  float h = fine->h * scale;
  if (!(h > 0.0f) || !isfinite(h)) h = 1.0f;
  int G[3];
  for (int it = 0; it < 64; ++it) {
    long long tot = 1;
    bool ok = true;
    for (int a = 0; a < 3; ++a) {
      float g = floorf(ext[a] / h) + 1.0f;
      if (!(g < 1024.0f)) ok = false;
      G[a] = g < 1.0f ? 1 : (g > 1024.0f ? 1024 : (int)g);
      tot *= G[a];
    }
    if (ok && tot <= max_cells) break;
    h *= 1.3f;
  }
  gp.h = h;
  gp.inv_h = 1.0f / h;
  gp.ncells = G[0] * G[1] * G[2];
  gp.n = fine->n;
  for (int a = 0; a < 3; ++a) gp.G[a] = G[a];
  gp.sample_dist = 0.0f;
  gp.sample_dists = 0;
  return gp;
}

__device__ __forceinline__ int cell_coord(float v, float mn, float inv_h, int G) {
  float c = floorf((v - mn) * inv_h);
  c = fminf(fmaxf(c, 0.0f), (float)(G - 1));  // NaN -> 0
  return (int)c;
}

// Sub-cells along x (first-level grid only): the points of a cell are kept in the order of their quarter of the
// cell, so that the x-run of a (dy, dz) row is ordered along x at a granularity of h / 4 and a query can leave out
// the ends of the run that lie outside its ball (grid_query_tpq_kernel).  The fine coordinate is floor(4 u) of the
// same u = (v - mn) * inv_h the cell coordinate is floor(u) of (4 u is exact), so cell == fine >> 2 for every v, and
// both are monotone in v.
constexpr int kSubShift = 2, kSub = 1 << kSubShift;
__device__ __forceinline__ int fine_coord(float v, float mn, float inv_h, int G) {
  float c = floorf((v - mn) * (inv_h * (float)kSub));
  c = fminf(fmaxf(c, 0.0f), (float)(G * kSub - 1));  // NaN -> 0
  return (int)c;
}

// The second-level grid's counting pass with its parameters worked out at the head of every workgroup (round 6; until then a
// one-thread launch for the parameters and a second one to clear the counters in front of the counting pass): a few dozen
// operations on values that are final when the launch starts, workgroup 0 leaves them in *gp for the launches behind it.  The
// counters [0, kCoarseMaxCells] are cleared by the caller (the search's memset, or the per-view call's dyn_warp launch).
__global__ void __launch_bounds__(256)
grid_coarse_count_kernel(const float *__restrict__ pts, const GridParams *__restrict__ fine, const unsigned *__restrict__ bbox,
                         GridParams *__restrict__ gp, float scale, float scale_many, const int32_t *__restrict__ open_count,
                         const int32_t *__restrict__ qcount, int max_cells, int32_t *__restrict__ cell_of,
                         int32_t *__restrict__ cell_count) {
  __shared__ GridParams s_gp;
  if (threadIdx.x == 0) {
    s_gp = coarse_params(fine, bbox, scale, scale_many, open_count, qcount, max_cells);
    if (blockIdx.x == 0) *gp = s_gp;
  }
  __syncthreads();
  if (*open_count == 0) return;  // nothing fell through the first level (the parameters are written either way)
  const GridParams g = s_gp;
  // points arrive in raster order of their source frame: neighbouring lanes mostly share a
  // cell, so the run leaders add whole runs (one atomic instruction per wave step)
  const int n_round = (g.n + 63) / 64 * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += gridDim.x * blockDim.x) {
    int c = -1;
    if (i < g.n) {
      int cx = cell_coord(pts[(size_t)i * 3 + 0], g.mn[0], g.inv_h, g.G[0]);
      int cy = cell_coord(pts[(size_t)i * 3 + 1], g.mn[1], g.inv_h, g.G[1]);
      int cz = cell_coord(pts[(size_t)i * 3 + 2], g.mn[2], g.inv_h, g.G[2]);
      c = (cz * g.G[1] + cy) * g.G[0] + cx;
      cell_of[i] = c;
    }
    wave_tile_count(cell_count, c);
  }
}

// Start offset of a cell's points in the cell-sorted array.
//   dense : start[c], one int per cell (second-level grid, <= 256 K cells).
//   sparse: the first-level grid of a depth-map cloud has ~9 M cells of which 0.15 % hold points, so the dense
//           table cost 36 MB of zeroing and two 36 MB scan passes per call.  Instead: one bit per cell, per 64
//           cells a 16-byte word {bits, rank = occupied cells before the word}, and start[j] for the j-th
//           occupied cell only.  start(c) = start[rank + popcount(bits below c)]: an empty cell gets the start of
//           the next occupied one, which is what the exclusive scan over all cells gave.  Two dependent loads
//           instead of one, 2 MB of table instead of 72.
//           The sparse index keeps kSub starts per occupied cell (the cell's quarters along x, see fine_coord):
//           start(c) = start[kSub * j], start of quarter s of an occupied cell = start[kSub * j + s].
struct CellIndex {
  const int32_t *start;
  const uint4 *tab;  // nullptr = dense
  __device__ __forceinline__ int at(int c) const {
    if (tab == nullptr) return start[c];
    const uint4 w = tab[c >> 6];
    const unsigned long long bits = ((unsigned long long)w.y << 32) | (unsigned long long)w.x;
    const int j = (int)w.z + __popcll(bits & ((1ull << (c & 63)) - 1ull));
    return start[j << kSubShift];
  }
  // sparse only: start of the points at fine x-coordinate >= f of the row whose first cell is `row` (f may be
  // kSub * G: one past the row's last cell, i.e. the start of the next row)
  __device__ __forceinline__ int at_fine(int row, int f) const {
    const int c = row + (f >> kSubShift);
    const uint4 w = tab[c >> 6];
    const unsigned long long bits = ((unsigned long long)w.y << 32) | (unsigned long long)w.x;
    const int j = (int)w.z + __popcll(bits & ((1ull << (c & 63)) - 1ull));
    const int sub = ((bits >> (c & 63)) & 1ull) ? (f & (kSub - 1)) : 0;  // an empty cell: the next occupied cell's start
    return start[(j << kSubShift) + sub];
  }
};

constexpr int kTabWords = (kGridMaxCells >> 6) + 1;  // (+1: cell index ncells, one past the last cell, is looked up too)
constexpr int kRankWordsPerBlock = 4096;
constexpr unsigned kScanSpinLimit = 1u << 22;   // polls of one predecessor word before a look-back gives up (never expected)

// zero the words of the bit table that cover cells [0, ncells] and the per-occupied-cell counters
__global__ void __launch_bounds__(256)
grid_tab_zero_kernel(const GridParams *__restrict__ gp, uint4 *__restrict__ tab, int32_t *__restrict__ occ_count) {
  const int nw = (gp->ncells >> 6) + 1;
  const int nc = gp->n < gp->ncells ? gp->n : gp->ncells;  // occupied cells <= min(points, cells)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += gridDim.x * blockDim.x) tab[i] = make_uint4(0u, 0u, 0u, 0u);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= nc * kSub; i += gridDim.x * blockDim.x) occ_count[i] = 0;
}

// cell of every point; its bit in the table (one atomic per run of lanes that share a cell)
__global__ void __launch_bounds__(256)
grid_mark_kernel(const float *__restrict__ pts, const GridParams *__restrict__ gp, int32_t *__restrict__ cell_of,
                 uint4 *__restrict__ tab) {
  const GridParams g = *gp;
  const int n_round = (g.n + 63) / 64 * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += gridDim.x * blockDim.x) {
    int c = -1;
    if (i < g.n) {
      int fx = fine_coord(pts[(size_t)i * 3 + 0], g.mn[0], g.inv_h, g.G[0]);
      int cy = cell_coord(pts[(size_t)i * 3 + 1], g.mn[1], g.inv_h, g.G[1]);
      int cz = cell_coord(pts[(size_t)i * 3 + 2], g.mn[2], g.inv_h, g.G[2]);
      c = (cz * g.G[1] + cy) * g.G[0] + (fx >> kSubShift);
      cell_of[i] = (c << kSubShift) | (fx & (kSub - 1));  // (<= 2^24 cells: 26 bits)
    }
    const RunInfo r = wave_runs(c);
    if (r.is_leader && c >= 0)
      atomicOr(reinterpret_cast<unsigned long long *>(&tab[c >> 6]), 1ull << (c & 63));
  }
}

// rank of every table word = occupied cells before it.  (Until round 6 two launches: bits per workgroup of kRankWordsPerBlock
// words, then every workgroup summed the (<= 65) totals before its own and scanned its words.  One launch in which a
// workgroup counted the bits of all words before its own itself took 25 us: the last workgroup's 1 MB walk.)
__device__ __forceinline__ int tab_word_bits(const uint4 *__restrict__ tab, int w) {
  const uint2 b = *reinterpret_cast<const uint2 *>(&tab[w]);
  return __popc(b.x) + __popc(b.y);
}

// (round 6: ONE launch.  A workgroup takes its block of kRankWordsPerBlock words by ticket, counts its bits, publishes the count
// as a tagged 64-bit word and adds up the words of ALL blocks before it -- at most 64, one or two per lane of its first
// wavefront, polled until they carry the tag; every predecessor is held by a workgroup that already runs (tickets), the spin is
// bounded like the scan's.  `state`: kRankBlocksMax words + the ticket behind them, zeroed with the search's state block.)
constexpr int kRankBlocksMax = (kTabWords + kRankWordsPerBlock - 1) / kRankWordsPerBlock;  // 65
__global__ void __launch_bounds__(1024)
grid_rank_kernel(const GridParams *__restrict__ gp, uint4 *__restrict__ tab, unsigned long long *__restrict__ state,
                 int32_t *__restrict__ ticket, int32_t *__restrict__ error, int32_t *__restrict__ nocc_out) {
  __shared__ int ws[16];
  __shared__ int s_blk, s_pre;
  const int nw = (gp->ncells >> 6) + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_blk = atomicAdd(ticket, 1);
  __syncthreads();
  const int blk = s_blk;
  const int w0 = blk * kRankWordsPerBlock;
  if (w0 >= nw) return;  // (nobody waits for a block without words: the blocks behind it have none either)
  int c[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int w = w0 + tid * 4 + k;
    c[k] = w < nw ? tab_word_bits(tab, w) : 0;
    s += c[k];
  }
  int x = s;
  for (int off = 1; off < 64; off <<= 1) {
    const int y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) ws[wave] = x;
  __syncthreads();
  if (wave == 0) {
    int total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) total += ws[w];
    if (lane == 0) __hip_atomic_store(&state[blk], (1ull << 32) | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int pre = 0;
    for (int b = lane; b < blk; b += 64) {
      unsigned spins = 0;
      unsigned long long v;
      for (;;) {
        v = __hip_atomic_load(&state[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 32) != 0ull) break;
        if (++spins > kScanSpinLimit) {  // keeps a protocol bug from hanging the GPU (results poisoned: grid_fallback_finish_kernel)
          atomicExch(error, 1);
          v = 0ull;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      pre += (int)(unsigned)v;
    }
    for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off, 64);
    if (lane == 0) s_pre = pre;
  }
  __syncthreads();
  int run = s_pre + x - s;
  for (int w = 0; w < wave; ++w) run += ws[w];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int w = w0 + tid * 4 + k;
    if (w < nw) tab[w].z = (unsigned)run;
    run += c[k];
    if (w == nw - 1) {  // occupied cells in total, and the entries of their counter array
      nocc_out[0] = run;
      nocc_out[1] = run * kSub;
    }
  }
}

// points per quarter of an occupied cell: cell_of[i] becomes kSub * (the cell's index among the occupied ones) + quarter
__global__ void __launch_bounds__(256)
grid_occ_count_kernel(const GridParams *__restrict__ gp, int32_t *__restrict__ cell_of, const uint4 *__restrict__ tab,
                      int32_t *__restrict__ occ_count) {
  const int n = gp->n;
  const int n_round = (n + 63) / 64 * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += gridDim.x * blockDim.x) {
    int j = -1;
    if (i < n) {
      const int cf = cell_of[i], c = cf >> kSubShift;
      const uint4 w = tab[c >> 6];
      const unsigned long long bits = ((unsigned long long)w.y << 32) | (unsigned long long)w.x;
      j = (((int)w.z + __popcll(bits & ((1ull << (c & 63)) - 1ull))) << kSubShift) | (cf & (kSub - 1));
      cell_of[i] = j;
    }
    wave_tile_count(occ_count, j);
  }
}

// exclusive scan of in[0 .. n) -> out[0 .. n]; n = *n_ptr is a device value (the cells of a dense grid, or the
// quarter-cell counters of the sparse one's occupied cells).  (Rounds 1-2: three launches -- block sums, their scan, apply;
// rounds 3-4: one workgroup walking the tiles with a carry -- 2 us for the 17 K occupied cells of the benchmark's cloud, but
// bound by ONE compute unit's load / store rate: 8 us for the 70 K quarter counters, 24 us with the chip busy.)
// One launch, one workgroup per tile of 8 K entries, tiles chained by look-back: a workgroup publishes its tile's sum
// (flag 1) in a 64-bit word of `state`, adds up the words of the tiles before it back to the first one that carries a whole
// prefix (flag 2; tile 0 does from the start), then publishes its own.  Flag and value travel in one atomic word, so no
// ordering between words is needed; a tile is a TICKET (see below), so a workgroup never waits for a tile nobody holds.  `state`
// and the ticket word are zeroed before the launch (the call's memset).  Inside a tile thread t owns the 16-byte groups (k * 1024 + t): every load
// and store instruction of a wavefront covers 1 KB of consecutive memory.  `in` must be readable up to the next multiple
// of four entries behind n (the workspace's arrays are).
constexpr int kScanVec = 2;
constexpr int kScanTile = 1024 * 4 * kScanVec;
constexpr int kScanMaxBlocks = 256;             // workgroups of a scan launch: they take tiles by ticket until none is left
// (round 6, after the advisor's finding: until then the tile was blockIdx.x -- dispatch order is NOT index order on eight XCDs
// with other views' kernels on the CUs: beside the tile pass one scan of 9 live tiles among 1013 launched took 102 us instead
// of 6 --, the spin was unbounded, and the launch was sized from the capacity.  Now: a ticket per tile from a word of the
// zeroed state block, so that a tile's predecessors are always held by workgroups that already run; at most kScanMaxBlocks
// workgroups that loop; a bounded spin that sets *error and goes on with a zero carry -- offsets stay inside the arrays, the
// caller poisons the results.)
__global__ void __launch_bounds__(1024)
grid_scan_kernel(const int32_t *__restrict__ in, const int32_t *__restrict__ n_ptr, int32_t *__restrict__ out,
                 unsigned long long *__restrict__ state, int32_t *__restrict__ ticket, int32_t *__restrict__ error) {
  __shared__ int s_tot[kScanVec * 16];  // [sub-tile][wavefront]
  __shared__ int s_carry;
  __shared__ int s_tile;
  const int n = *n_ptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (;;) {
  if (tid == 0) s_tile = atomicAdd(ticket, 1);
  __syncthreads();
  const int tile = s_tile;
  if ((long long)tile * kScanTile > n) return;  // (<= n: the tile that holds out[n]; uniform over the workgroup)
  int4 v[kScanVec];
  int x[kScanVec];
#pragma unroll
  for (int k = 0; k < kScanVec; ++k) {
    const int idx = tile * kScanTile + (k * 1024 + tid) * 4;
    int4 q = make_int4(0, 0, 0, 0);
    if (idx < n) q = *reinterpret_cast<const int4 *>(in + idx);
    q.y = idx + 1 < n ? q.y : 0;
    q.z = idx + 2 < n ? q.z : 0;
    q.w = idx + 3 < n ? q.w : 0;
    v[k] = q;
  }
#pragma unroll
  for (int k = 0; k < kScanVec; ++k) {
    int t = v[k].x + v[k].y + v[k].z + v[k].w;
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(t, off, 64);
      if (lane >= off) t += y;
    }
    x[k] = t;  // inclusive over the wavefront's groups of sub-tile k
    if (lane == 63) s_tot[k * 16 + wave] = t;
  }
  __syncthreads();
  if (wave == 0) {
    // exclusive scan of the kScanVec * 16 totals in (sub-tile, wavefront) order, one per lane
    const int own = lane < kScanVec * 16 ? s_tot[lane] : 0;
    int t = own;
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(t, off, 64);
      if (lane >= off) t += y;
    }
    if (lane < kScanVec * 16) s_tot[lane] = t - own;
    const int sum = __shfl(t, 63, 64);  // the tile's sum
    // look-back: 64 earlier tiles at a time
    int carry = 0;
    if (tile > 0) {
      if (lane == 0)
        __hip_atomic_store(&state[tile], (1ull << 32) | (unsigned)sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int back = tile - 1; back >= 0; back -= 64) {
        const int i = back - lane;
        unsigned long long w = 3ull << 32;  // (lanes before tile 0: nothing to add, nothing to wait for)
        if (i >= 0) {
          unsigned spins = 0;
          for (;;) {
            w = __hip_atomic_load(&state[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((w >> 32) != 0ull) break;
            if (++spins > kScanSpinLimit) {  // keeps a protocol bug from hanging the GPU
              atomicExch(error, 1);
              w = 2ull << 32;  // "whole prefix 0": the walk stops here, the offsets stay inside the arrays
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        const unsigned long long whole = __ballot((w >> 32) == 2ull);
        const int stop = whole ? __builtin_ctzll(whole) : 63;  // nearest tile with a whole prefix
        int c = (lane <= stop && i >= 0) ? (int)(unsigned)w : 0;
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
        carry += c;
        if (whole) break;
      }
    }
    if (lane == 0) {
      __hip_atomic_store(&state[tile], (2ull << 32) | (unsigned)(carry + sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_carry = carry;
    }
  }
  __syncthreads();
  const int carry = s_carry;
#pragma unroll
  for (int k = 0; k < kScanVec; ++k) {
    const int idx = tile * kScanTile + (k * 1024 + tid) * 4;
    int4 q;
    q.x = carry + s_tot[k * 16 + wave] + x[k] - (v[k].x + v[k].y + v[k].z + v[k].w);
    q.y = q.x + v[k].x;
    q.z = q.y + v[k].y;
    q.w = q.z + v[k].z;
    if (idx + 3 <= n) {
      *reinterpret_cast<int4 *>(out + idx) = q;
    } else if (idx <= n) {  // the group that holds out[n] = the total
      out[idx] = q.x;
      if (idx + 1 <= n) out[idx + 1] = q.y;
      if (idx + 2 <= n) out[idx + 2] = q.z;
    }
  }
  __syncthreads();  // (s_tot / s_carry / s_tile are rewritten by the next tile)
  }
}

// slot of this lane inside its cell, counting the cell's (already scanned) count DOWN: no second
// counter array to zero.  The order inside a cell is arbitrary either way (the results do not
// depend on it: only sorted distance values are consumed).
__device__ __forceinline__ int wave_cell_take(int32_t *__restrict__ remaining, int c) {
  const int lane = threadIdx.x & 63;
  RunInfo r = wave_runs(c);
  int base = 0;
  if (r.is_leader && c >= 0) base = atomicSub(&remaining[c], r.length) - r.length;
  base = __shfl(base, r.leader, 64);
  return base + (lane - r.leader);
}

__global__ void __launch_bounds__(256)
grid_fill_kernel(const float *__restrict__ pts, const GridParams *__restrict__ gp,
                 const int32_t *__restrict__ cell_of, const int32_t *__restrict__ cell_start,
                 int32_t *__restrict__ cell_count, float4 *__restrict__ sorted,
                 const int32_t *__restrict__ gate) {
  // (cell_of holds whatever indexes cell_start / cell_count: the cell id of a dense grid, the occupied-cell
  // index of the sparse one)
  if (gate && *gate == 0) return;
  const int n = gp->n;
  const int n_round = (n + 63) / 64 * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += gridDim.x * blockDim.x) {
    const int c = i < n ? cell_of[i] : -1;
    const int slot = wave_cell_take(cell_count, c);
    if (i < n)
      sorted[cell_start[c] + slot] = make_float4(pts[(size_t)i * 3], pts[(size_t)i * 3 + 1], pts[(size_t)i * 3 + 2],
                                                 __int_as_float(i));
  }
}

// ---- the query kernel --------------------------------------------------------
// Sorted best-list, one value per lane (lane k = k-th smallest so far, +inf = empty).
struct BestList {
  float best;
  float mx;  // wave-uniform copy of lane KK-1
};

__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// lane i <- lane i-1, lane 0 <- 0: DPP wave_shr:1 with bound_ctrl (no LDS traffic)
__device__ __forceinline__ float wave_shift_up1_zero(float v) {
  int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true);
  return __int_as_float(r);
}

// Insert the candidates of one batch that beat the current K+1-th best.  The list is
// sorted ascending, so inserting v is: every lane holding a value > v takes
// max(left neighbour, v) -- the first such lane receives v (its left neighbour is <= v, or
// the zero fill for lane 0; squared distances are >= 0), the others shift up by one --
// which is the median of (own value, left neighbour, v) on every lane.
__device__ __forceinline__ void best_insert_batch(BestList &b, float d, bool valid, int KK, int lane) {
  (void)lane;
  unsigned long long mask = __ballot(valid && d < b.mx);
  if (!mask) return;
  // The filter used the threshold from before this batch; a candidate that no longer beats
  // the (tighter) current threshold lands at a position >= K+1, which is never read.
  while (mask) {
    int l = __builtin_ctzll(mask);
    mask &= mask - 1;
    float v = readlane_f(d, l);
    float up = wave_shift_up1_zero(b.best);
    // lanes holding a value <= v keep it (up <= best <= v: the median is best); lanes holding a
    // larger one take max(up, v) (best is the largest of the three): one v_med3_f32
    b.best = __builtin_amdgcn_fmed3f(b.best, up, v);
  }
  b.mx = readlane_f(b.best, KK - 1);
}

// value of lane (lane ^ J): register-to-register lane exchanges (DPP quad / row patterns, the
// gfx950 row and half-wave swap instructions) instead of ds_bpermute, whose ~100-cycle round
// trips through the LDS crossbar serialise the 21 dependent stages of the sort
template <int J>
__device__ __forceinline__ float lane_xor(float v, int lane) {
  const int iv = __float_as_int(v);
  if (J == 1) return __int_as_float(__builtin_amdgcn_update_dpp(iv, iv, 0xB1, 0xf, 0xf, false));  // quad_perm [1,0,3,2]
  if (J == 2) return __int_as_float(__builtin_amdgcn_update_dpp(iv, iv, 0x4E, 0xf, 0xf, false));  // quad_perm [2,3,0,1]
  if (J == 4) {
    // banks 0,2 read lane+4 (row_shl:4), banks 1,3 read lane-4 (row_shr:4)
    int t = __builtin_amdgcn_update_dpp(iv, iv, 0x104, 0xf, 0x5, false);
    t = __builtin_amdgcn_update_dpp(t, iv, 0x114, 0xf, 0xa, false);
    return __int_as_float(t);
  }
  if (J == 8) return __int_as_float(__builtin_amdgcn_update_dpp(iv, iv, 0x128, 0xf, 0xf, false));  // row_ror:8
  if (J == 16) {
    // {rows 0,0,2,2}, {rows 1,1,3,3}
    auto r = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);
    return __int_as_float((lane & 16) ? r[0] : r[1]);
  }
  // J == 32: {low half twice}, {high half twice}
  auto r = __builtin_amdgcn_permlane32_swap(iv, iv, false, false);
  return __int_as_float((lane & 32) ? r[0] : r[1]);
}

template <int K, int J>
__device__ __forceinline__ float sort_step(float v, int lane) {
  const float o = lane_xor<J>(v, lane);
  const bool up = (lane & K) == 0;     // ascending block
  const bool lower = (lane & J) == 0;  // this lane keeps the smaller of the pair
  // NaN never reaches here (distances of finite points); fminf/fmaxf keep the multiset
  const float mn = fminf(v, o), mx = fmaxf(v, o);
  return (up == lower) ? mn : mx;
}

// ascending bitonic sort of one value per lane across the wavefront
__device__ __forceinline__ float wave_sort_asc(float v, int lane) {
  v = sort_step<2, 1>(v, lane);
  v = sort_step<4, 2>(v, lane);
  v = sort_step<4, 1>(v, lane);
  v = sort_step<8, 4>(v, lane);
  v = sort_step<8, 2>(v, lane);
  v = sort_step<8, 1>(v, lane);
  v = sort_step<16, 8>(v, lane);
  v = sort_step<16, 4>(v, lane);
  v = sort_step<16, 2>(v, lane);
  v = sort_step<16, 1>(v, lane);
  v = sort_step<32, 16>(v, lane);
  v = sort_step<32, 8>(v, lane);
  v = sort_step<32, 4>(v, lane);
  v = sort_step<32, 2>(v, lane);
  v = sort_step<32, 1>(v, lane);
  v = sort_step<64, 32>(v, lane);
  v = sort_step<64, 16>(v, lane);
  v = sort_step<64, 8>(v, lane);
  v = sort_step<64, 4>(v, lane);
  v = sort_step<64, 2>(v, lane);
  v = sort_step<64, 1>(v, lane);
  return v;
}

__device__ __forceinline__ float dist2(float qx, float qy, float qz, float4 p) {
  float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
  float d = dx * dx;
  d = d + dy * dy;
  d = d + dz * dz;
  return d;
}

// scan sorted[start, end): 64 candidates per lane-step, four steps of loads in flight.
// `bound`: candidates farther than this can be ignored (+inf = no bound).
__device__ __forceinline__ void scan_range(BestList &b, bool &first, const float4 *__restrict__ sorted,
                                           int start, int end, float qx, float qy, float qz, int KK,
                                           int lane, float bound = __builtin_inff()) {
  for (int j0 = start; j0 < end; j0 += 256) {
    float d[4];
    bool v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int j = j0 + u * 64 + lane;
      v[u] = j < end;
      float4 p = v[u] ? sorted[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      d[u] = dist2(qx, qy, qz, p);
      v[u] = v[u] && d[u] <= bound;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (j0 + u * 64 >= end) break;
      if (first) {
        // empty list: sort the first 64 candidates instead of 64 serial insertions
        b.best = wave_sort_asc(v[u] ? d[u] : __builtin_inff(), lane);
        b.mx = readlane_f(b.best, KK - 1);
        first = false;
      } else {
        best_insert_batch(b, d[u], v[u], KK, lane);
      }
    }
  }
}

// mean over columns 1..K of the sorted list (column 0 = the point itself is dropped;
// columns beyond the number of points count as 0, as pytorch3d pads).  Summation order is
// a fixed butterfly over 64 slots -- s[i] += s[i+32], s[i] += s[i+16], ... s[0] += s[1] --
// which the CPU oracle and the brute-force kernel reproduce, so results stay bit-identical.
//
// Cross-set mode (queries from a second array, A17 track-to-base filter,
// pgdvs_renderer_dyn_track.py:299-312): there is no self column, all KK columns count.
struct QuerySrc {
  const float *qpts;      // null: the queries are the points themselves (cell-sorted order)
  const int32_t *qcount;
  int first_col;          // 1: drop column 0 (self), 0: keep it
  // second-level pass: only the queries listed here (ids into qpts / qsorted) are processed, and
  // self-queries take their position from `qsorted` (the first-level sorted array) while the
  // grid being searched is another one
  const int32_t *list;
  const int32_t *list_count;
  const float4 *qsorted;
};

__device__ __forceinline__ int query_count(const QuerySrc &qs, int n_points) {
  return qs.list ? *qs.list_count : (qs.qpts ? *qs.qcount : n_points);
}

// q: base query id (index into qpts, or into the sorted array of self-queries)
__device__ __forceinline__ float4 load_query(const QuerySrc &qs, const float4 *__restrict__ sorted, int q) {
  if (qs.qpts == nullptr) return (qs.qsorted ? qs.qsorted : sorted)[q];
  return make_float4(qs.qpts[(size_t)q * 3], qs.qpts[(size_t)q * 3 + 1], qs.qpts[(size_t)q * 3 + 2],
                     __int_as_float(q));
}

__device__ __forceinline__ void knn_finish(float best, int KK, int first_col, int n, int lane, int orig,
                                           float *__restrict__ avg_out) {
  float s = (lane >= first_col && lane < KK && lane < n) ? best : 0.0f;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s = s + __shfl_down(s, off, 64);
  if (lane == 0) avg_out[orig] = s / (float)(KK - first_col);
}

// Squared distance (lower bound) from coordinate q to the slab of cells [c0, c1] along one
// axis.  The slab is widened by a rounding margin, and is unbounded on a side that touches
// the grid boundary (boundary cells also hold the points clamped into them).
__device__ __forceinline__ float box_axis_dist2(float q, float mn, float h, int c0, int c1, int G) {
  float lo = c0 <= 0 ? -__builtin_inff() : mn + (float)c0 * h - 0.02f * h;
  float hi = c1 >= G - 1 ? __builtin_inff() : mn + (float)(c1 + 1) * h + 0.02f * h;
  float d = fmaxf(fmaxf(lo - q, q - hi), 0.0f);
  return d * d;
}

__device__ __forceinline__ void
grid_query_body(const GridParams *__restrict__ gp, const float4 *__restrict__ sorted,
                const CellIndex cell_start, int KK, QuerySrc qs, float *__restrict__ avg_out,
                int32_t *__restrict__ stats, int ring_cap, int32_t *__restrict__ fb_count,
                int32_t *__restrict__ fb_list, float *__restrict__ fb_bound) {
  const GridParams g = *gp;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nq = query_count(qs, g.n);
  // persistent waves: the grid is sized for the machine, not for the (device-side) count
  for (int qi = blockIdx.x * 4 + wave; qi < nq; qi += gridDim.x * 4) {
  const int q = qs.list ? qs.list[qi] : qi;
  const float4 qp = load_query(qs, sorted, q);
  const float qx = qp.x, qy = qp.y, qz = qp.z;
  const int orig = __float_as_int(qp.w);
  if (qs.qpts && !(isfinite(qx) && isfinite(qy) && isfinite(qz))) {
    // a non-finite query has NaN/inf distances to everything upstream: its mean fails every
    // `< threshold` test
    if (lane == 0) avg_out[orig] = __builtin_nanf("");
    continue;
  }
  const int cx = cell_coord(qx, g.mn[0], g.inv_h, g.G[0]);
  const int cy = cell_coord(qy, g.mn[1], g.inv_h, g.G[1]);
  const int cz = cell_coord(qz, g.mn[2], g.inv_h, g.G[2]);
  BestList b;
  b.best = __builtin_inff();
  b.mx = __builtin_inff();
  bool first = true;
  bool done = false;
  int rings = 0;
  for (int r = 1; r <= ring_cap && !done; ++r) {
    rings = r;
    // Shell of Chebyshev radius r (r == 1: the whole 3x3x3 cube) as x-runs: the lanes look
    // up the cell ranges of 64 (dy,dz) rows at a time, then the non-empty runs are scanned.
    const int side = 2 * r + 1;
    const int nrows = side * side;
    for (int base = 0; base < nrows; base += 64) {
      int row_id = base + lane;
      // r == 1: visit the query's own row first so the threshold is tight before the
      // neighbouring rows are (mostly) pruned
      // r == 1: visit the rows nearest first -- own row, the two rows of the same z layer, the
      // two rows above / below, then the four diagonal ones -- so that the threshold is tight
      // before the farther rows are (mostly) pruned and fewer candidates are inserted only to
      // be evicted later
      if (r == 1) row_id = row_id < 9 ? (int)((0x862071534ull >> (row_id * 4)) & 15) : row_id;
      int s0 = 0, e0 = 0, s1 = 0, e1 = 0;
      float bd0 = 0.0f, bd1 = 0.0f;  // lower bound of the squared distance to any point of the run
      if (row_id < nrows) {
        int rz = (int)(((float)row_id + 0.5f) / (float)side);
        int dz = rz - r, dy = row_id - rz * side - r;
        int z = cz + dz, y = cy + dy;
        if (z >= 0 && z < g.G[2] && y >= 0 && y < g.G[1]) {
          int row = (z * g.G[1] + y) * g.G[0];
          float byz = box_axis_dist2(qy, g.mn[1], g.h, y, y, g.G[1]) + box_axis_dist2(qz, g.mn[2], g.h, z, z, g.G[2]);
          bool outer = r == 1 || dz == -r || dz == r || dy == -r || dy == r;
          if (outer) {
            int x0 = cx - r < 0 ? 0 : cx - r;
            int x1 = cx + r >= g.G[0] ? g.G[0] - 1 : cx + r;
            s0 = cell_start.at(row + x0);
            e0 = cell_start.at(row + x1 + 1);
            bd0 = byz + box_axis_dist2(qx, g.mn[0], g.h, x0, x1, g.G[0]);
          } else {
            if (cx - r >= 0) {
              s0 = cell_start.at(row + cx - r);
              e0 = cell_start.at(row + cx - r + 1);
              bd0 = byz + box_axis_dist2(qx, g.mn[0], g.h, cx - r, cx - r, g.G[0]);
            }
            if (cx + r < g.G[0]) {
              s1 = cell_start.at(row + cx + r);
              e1 = cell_start.at(row + cx + r + 1);
              bd1 = byz + box_axis_dist2(qx, g.mn[0], g.h, cx + r, cx + r, g.G[0]);
            }
          }
        }
      }
      unsigned long long m0 = __ballot(e0 > s0);
      while (m0) {
        int l = __builtin_ctzll(m0);
        m0 &= m0 - 1;
        if (readlane_f(bd0, l) >= b.mx) continue;  // no point of this run can enter the list
        scan_range(b, first, sorted, __builtin_amdgcn_readlane(s0, l), __builtin_amdgcn_readlane(e0, l), qx,
                   qy, qz, KK, lane);
      }
      unsigned long long m1 = __ballot(e1 > s1);
      while (m1) {
        int l = __builtin_ctzll(m1);
        m1 &= m1 - 1;
        if (readlane_f(bd1, l) >= b.mx) continue;
        scan_range(b, first, sorted, __builtin_amdgcn_readlane(s1, l), __builtin_amdgcn_readlane(e1, l), qx,
                   qy, qz, KK, lane);
      }
    }
    // distance from the query to the nearest face of the scanned cube that still has
    // cells behind it; everything unseen is at least that far away
    float db = __builtin_inff();
    bool open = false;
    const int c[3] = {cx, cy, cz};
    const float qv[3] = {qx, qy, qz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (c[a] - r > 0) {
        open = true;
        db = fminf(db, qv[a] - (g.mn[a] + (float)(c[a] - r) * g.h));
      }
      if (c[a] + r < g.G[a] - 1) {
        open = true;
        db = fminf(db, (g.mn[a] + (float)(c[a] + r + 1) * g.h) - qv[a]);
      }
    }
    if (!open) {
      done = true;  // the cube covered the whole grid
    } else {
      float safe = db - 0.02f * g.h;
      if (safe > 0.0f && b.mx <= safe * safe) done = true;  // mx finite => list is full
    }
  }
  if (stats && lane == 0) atomicAdd(&stats[done ? (rings < 15 ? rings : 14) : 15], 1);
  if (!done) {
    // isolated point: handed to the cooperative exact scan (grid_fallback_kernel)
    if (lane == 0) {
      int slot = atomicAdd(fb_count, 1);
      fb_list[slot] = q;
      fb_bound[slot] = b.mx;  // the K+1-th best seen so far bounds the true one from above
    }
    continue;
  }
  knn_finish(b.best, KK, qs.first_col, g.n, lane, orig, avg_out);
  }
}

// ---- first pass of self-queries: one THREAD per query, ring 1 only ----------------------
// With ~24 points per occupied cell, 99.6 % of the queries of a depth-map cloud are settled by
// the 3x3x3 block around their own cell (ring histogram, PGDVS_KNN_STATS=1).  For those the
// wavefront-per-query search above spends most of its ~1050 vector instructions per query on
// serial insertions and on a 64-wide sort; here a lane owns a query and keeps its K+1 best in
// registers as a sorted list, inserting with one median-of-three per slot.  Lanes are
// consecutive in cell-sorted order, so a wavefront's lanes walk (nearly) the same nine x-runs
// of the sorted point array: the per-lane candidate loads coalesce into broadcasts.  A
// candidate that beats the lane's (K+1)-th best is parked in a 4-deep per-lane queue; the
// (K+1)-instruction insertion chain runs only when some lane's queue is full, i.e. about
// once per four *accepted* candidates of the busiest lane instead of once per candidate.
// About 350 vector instructions per query.  Queries whose list is not provably complete after
// ring 1 (the same criterion as above) are handed to the wavefront-per-query search.
// The average is summed in the same 64-slot butterfly order as knn_finish.
// (Round 5, measured and dropped: lanes that stop requesting candidates at 3 / 4 / 5 queued ones, a flush of as many chains
// as the fullest queue holds once 8 or 16 lanes wait -- an independent-lanes model promised -17 % vector instructions;
// bit-exact, and 192-213 us against 160: the lanes of a wavefront are neighbours and accept together, waiting only
// lengthens the walk.  Queues of 5 / 6 / 8 without waiting: 155-158 us.)
constexpr int kTpqQueue = 4;
// second attempt of a wavefront (below): four times the first threshold, at most what the block can certify.  (Until round 5:
// that bound itself -- every list of the wavefront then takes whatever the block holds, 535 us for the noisy scene's pass
// against 470 with 4 x and 420 with 2.5 x, where 13 k lists come out short a second time and the ring search grows by 50 us.)
constexpr float kTpqRetryScale = 4.0f;

// insert c into the ascending list a (the largest element drops out): slot i keeps its value
// if that is <= c, takes c if its left neighbour is <= c < a[i], and takes the left neighbour
// otherwise -- the median of (a[i-1], a[i], c), one instruction per slot, evaluated from the top
// down so that every slot still sees its neighbour's old value
template <int KK>
__device__ __forceinline__ void tpq_insert(float (&a)[KK], float c) {
#pragma unroll
  for (int i = KK - 1; i > 0; --i) a[i] = __builtin_amdgcn_fmed3f(a[i - 1], a[i], c);
  // (one instruction: fminf canonicalises both operands first; neither is NaN here -- a NaN distance never passes
  // the `d < mx` test that feeds the queue)
  asm("v_min_f32 %0, %1, %2" : "=v"(a[0]) : "v"(a[0]), "v"(c));
}

template <int KK>
__global__ void __launch_bounds__(256, 5)
grid_query_tpq_kernel(const GridParams *__restrict__ gp, const float4 *__restrict__ sorted,
                      const CellIndex cell_start, int first_col, float *__restrict__ avg_out,
                      int32_t *__restrict__ open_count, int32_t *__restrict__ open_list, float thr_mult) {
  __shared__ int2 s_run[9][256];
  __shared__ float s_bd[9][256];
  __shared__ float s_thr[256];  // the lane's starting threshold (see below)
  __shared__ float s_safe[256];  // ... and the squared distance up to which its block can certify a list
  __shared__ unsigned long long s_order[256];  // ... and its row numbers in visiting order, four bits each
  const GridParams g = *gp;
  const int n = g.n;
  const int tid = threadIdx.x;
  // persistent workgroups (no barrier is used: every wavefront only touches its own LDS columns)
  for (int q0 = blockIdx.x * 256; q0 < n; q0 += gridDim.x * 256) {
  const int q = q0 + tid;
  const bool live = q < n;
  const float4 qp = sorted[live ? q : n - 1];
  const float qx = qp.x, qy = qp.y, qz = qp.z;
  // points in the block (the nine (dy,dz) rows as x-runs): the density estimate behind the starting threshold below
  float t0;
  {
  int ncand = 0;
  const int cx = cell_coord(qx, g.mn[0], g.inv_h, g.G[0]);
  const int cy = cell_coord(qy, g.mn[1], g.inv_h, g.G[1]);
  const int cz = cell_coord(qz, g.mn[2], g.inv_h, g.G[2]);
  const int x0 = cx - 1 < 0 ? 0 : cx - 1;
  const int x1 = cx + 1 >= g.G[0] ? g.G[0] - 1 : cx + 1;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int z = cz + k / 3 - 1, y = cy + k % 3 - 1;
    if (live && z >= 0 && z < g.G[2] && y >= 0 && y < g.G[1]) {
      const int row = (z * g.G[1] + y) * g.G[0];
      ncand += cell_start.at(row + x1 + 1) - cell_start.at(row + x0);
    }
  }
  // Acceptance threshold from the start.  The points sample a surface: the squared radius that holds K+1 of them is
  // about (K+1) / (pi * density), and the block's own candidate count measures the density -- on the benchmark cloud
  // r^2 / ((K+1) h^2 / candidates) is 3.3 in the median, 4.0 at the 99th percentile.  Starting from 4.5 x that instead
  // of +inf, a lane accepts ~70 candidates instead of ~100 (every accepted candidate costs all 64 lanes a (K+1)-long
  // insertion chain) and skips rows farther than the threshold.  The list simply starts out filled with the threshold
  // (no register beside it -- the kernel sits at the 96-register edge of five waves per SIMD -- and a copy in LDS for
  // the check at the end): accepted candidates are strictly smaller and push it out.  A lane whose last slot still holds it at the end (0.5 %: the estimate was too
  // small for it) proves nothing and goes to the wavefront-per-query search like any other open query, so the results
  // do not depend on the estimate.
  t0 = (KK >= 17 && ncand >= KK) ? (KK >= 40 ? thr_mult : 6.0f) * (float)KK * g.h * g.h / (float)ncand : __builtin_inff();
  // What the block can certify at all: the distance to its nearest face that still has cells behind it (minus the
  // rounding margin), squared -- a list whose last entry lies beyond it is not complete whatever it holds.  Kept in LDS:
  // it is the threshold of the second attempt below, and the completeness test at the end.
  {
    float db = __builtin_inff();
    bool open = false;
    const int c[3] = {cx, cy, cz};
    const float qv[3] = {qx, qy, qz};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      if (c[ax] - 1 > 0) {
        open = true;
        db = fminf(db, qv[ax] - (g.mn[ax] + (float)(c[ax] - 1) * g.h));
      }
      if (c[ax] + 1 < g.G[ax] - 1) {
        open = true;
        db = fminf(db, (g.mn[ax] + (float)(c[ax] + 2) * g.h) - qv[ax]);
      }
    }
    const float safe = db - 0.02f * g.h;
    // +inf: nothing behind any face (everything is in the block); 0: the query sits on the margin (never complete)
    s_safe[tid] = !open ? __builtin_inff() : (safe > 0.0f ? safe * safe : 0.0f);
  }
  }
  float a[KK];
  float mx;
  bool cut;
  // Second attempt, per wavefront: the density estimate above assumes points on a SURFACE.  In a cloud from noisy depth
  // (a slab several cells thick) it is too tight for every query -- round 4's noisy scene sent 303 k of 311 k queries to the
  // ring search, 2.5 ms per view instead of 0.9 -- although the block holds their neighbours.  When a quarter of a
  // wavefront's lists come out short, the wavefront runs the block again with kTpqRetryScale times the threshold (at most
  // the largest one that can still certify a list: the block's own bound above) for those lanes; the others repeat their search unchanged.  Isolated
  // short lists (0.5 % on the benchmark's cloud) keep going to the ring search: a repeat would cost their whole wavefront.
  for (int attempt = 0;; ++attempt) {
  // The nine rows as x-runs, each cut down to the part of it the ball of the starting threshold can reach: the points
  // of a row are ordered along x in quarters of a cell (fine_coord), and a candidate closer than t0 lies within
  // sqrt(t0 - (the row's distance in y and z)) of the query along x.  On the benchmark's cloud this leaves out ~40 % of
  // the block's candidates before a single load.  (The margins of the box distances, plus one for the square root.)
  // (The cell of the query is worked out again here, from copies of its coordinates the compiler cannot see through:
  // kept from the first time, cx .. bx would stay live across the candidate loop for the sake of the second attempt,
  // eight registers the kernel does not have -- 54 spilled.)
  {
  float lx = qx, ly = qy, lz = qz;
  asm volatile("" : "+v"(lx), "+v"(ly), "+v"(lz));
  const int cx = cell_coord(lx, g.mn[0], g.inv_h, g.G[0]);
  const int cy = cell_coord(ly, g.mn[1], g.inv_h, g.G[1]);
  const int cz = cell_coord(lz, g.mn[2], g.inv_h, g.G[2]);
  const int x0 = cx - 1 < 0 ? 0 : cx - 1;
  const int x1 = cx + 1 >= g.G[0] ? g.G[0] - 1 : cx + 1;
  const float bx = box_axis_dist2(lx, g.mn[0], g.h, x0, x1, g.G[0]);
  unsigned key[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int row_id = (int)((0x862071534ull >> (k * 4)) & 15);
    const int dz = row_id / 3 - 1, dy = row_id % 3 - 1;
    const int z = cz + dz, y = cy + dy;
    int2 se = make_int2(0, 0);
    float bd = 0.0f;
    if (live && z >= 0 && z < g.G[2] && y >= 0 && y < g.G[1]) {
      const int row = (z * g.G[1] + y) * g.G[0];
      const float byz = box_axis_dist2(ly, g.mn[1], g.h, y, y, g.G[1]) + box_axis_dist2(lz, g.mn[2], g.h, z, z, g.G[2]);
      bd = bx + byz;
      const float hw2 = t0 - byz;
      if (hw2 > 0.0f) {
        const float hw = __builtin_amdgcn_sqrtf(hw2) * 1.0001f + 0.02f * g.h;
        int f0 = fine_coord(lx - hw, g.mn[0], g.inv_h, g.G[0]);
        int f1 = fine_coord(lx + hw, g.mn[0], g.inv_h, g.G[0]);
        f0 = f0 < x0 * kSub ? x0 * kSub : f0;
        f1 = f1 > x1 * kSub + (kSub - 1) ? x1 * kSub + (kSub - 1) : f1;
        if (f0 <= f1) {
          se.x = cell_start.at_fine(row, f0);
          se.y = cell_start.at_fine(row, f1 + 1);
        }
      }
    }
    s_run[k][tid] = se;
    s_bd[k][tid] = bd;
    // visiting key: the row's box distance with the row number in its low four bits (non-negative floats order like
    // their bit patterns; rows without points last)
    key[k] = se.y > se.x ? ((__float_as_uint(bd) & ~15u) | (unsigned)k) : (0xfffffff0u | (unsigned)k);
  }
  // Every lane visits ITS rows nearest first (sorting network on the nine keys, 25 min / max pairs): the fixed row
  // order above is the nearest-first order of a query in the middle of its cell, but for a query near a cell corner
  // it is close to farthest-first -- nearly every candidate then enters the list only to be evicted (~280 accepted
  // candidates on the busiest lane of a wavefront against ~115 on average), and the insertion chains run for all 64
  // lanes whenever ONE lane's queue is full.  The result does not depend on the order.
  {
#define PGDVS_CE(i, j)                        \
  {                                           \
    const unsigned lo = key[i] < key[j] ? key[i] : key[j]; \
    key[j] = key[i] < key[j] ? key[j] : key[i];            \
    key[i] = lo;                              \
  }
    PGDVS_CE(0, 3) PGDVS_CE(1, 7) PGDVS_CE(2, 5) PGDVS_CE(4, 8)
    PGDVS_CE(0, 7) PGDVS_CE(2, 4) PGDVS_CE(3, 8) PGDVS_CE(5, 6)
    PGDVS_CE(0, 2) PGDVS_CE(1, 3) PGDVS_CE(4, 5) PGDVS_CE(7, 8)
    PGDVS_CE(1, 4) PGDVS_CE(3, 6) PGDVS_CE(5, 7)
    PGDVS_CE(0, 1) PGDVS_CE(2, 4) PGDVS_CE(3, 5) PGDVS_CE(6, 8)
    PGDVS_CE(2, 3) PGDVS_CE(4, 5) PGDVS_CE(6, 7)
    PGDVS_CE(1, 2) PGDVS_CE(3, 4) PGDVS_CE(5, 6)
#undef PGDVS_CE
  }
  {
    unsigned long long order = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) order |= (unsigned long long)(key[k] & 15u) << (4 * k);
    s_order[tid] = order;  // (read back per row switch: two registers fewer across the candidate loop)
  }
  }
  __builtin_amdgcn_sched_barrier(0);  // (the 51 copies must not become live while the run set-up above still is)
  {
    s_thr[tid] = t0;
#pragma unroll
    for (int i = 0; i < KK; ++i) a[i] = t0;
  }
  mx = a[KK - 1];
  float qd[kTpqQueue];
#pragma unroll
  for (int u = 0; u < kTpqQueue; ++u) qd[u] = __builtin_inff();
  int qc = 0;
#ifdef PGDVS_AB_TPQ_STATS
  int dbg_acc = 0, dbg_cand = 0, dbg_flush = 0, dbg_steps = 0;
#endif
  int k = -1;
  // the lane's current run as BYTE offsets into `sorted` (16 bytes per point; the host takes this pass only below
  // 2^27 points): the load then takes scalar base + 32-bit vector offset as it stands -- no sign extension, no
  // 64-bit shift-add per candidate
  unsigned j = 0u, e = 0u;
  // Two candidate buffers: each half-step requests a candidate into one buffer -- for every lane,
  // unconditionally, from a clamped index, so that no exec-masked move forces a wait right
  // behind the load -- and evaluates the candidate the previous half-step requested into the
  // other.  The load of a candidate is thus always a full half-step (and the other wavefronts'
  // turns) ahead of its use.
  float4 ca = make_float4(0.f, 0.f, 0.f, 0.f), cb = ca;
  bool va = false, vb = false;
  auto half_step = [&](float4 &c_issue, bool &v_issue, const float4 &c_use, const bool v_use) -> bool {
    if (j >= e && k < 9) {  // this lane's run is exhausted: next row (one per half-step)
      ++k;
      if (k < 9) {
        const int row = (int)(s_order[tid] >> (4 * k)) & 15;
        const int2 se = s_run[row][tid];
        j = (unsigned)se.x * 16u;
        e = s_bd[row][tid] < mx ? (unsigned)se.y * 16u : j;  // no point of the run can enter the list: skip it
      }
    }
    v_issue = j < e;
    c_issue = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sorted) + (size_t)(v_issue ? j : 0u));
    j += v_issue ? 16u : 0u;
    if (v_use) {
      const float d = dist2(qx, qy, qz, c_use);
      if (d < mx) {  // (stale threshold: a candidate that no longer qualifies lands beyond column K)
#pragma unroll
        for (int u = kTpqQueue - 1; u > 0; --u) qd[u] = qd[u - 1];
        qd[0] = d;
        ++qc;
#ifdef PGDVS_AB_TPQ_STATS
        ++dbg_acc;
#endif
      }
#ifdef PGDVS_AB_TPQ_STATS
      ++dbg_cand;
#endif
    }
    if (__builtin_amdgcn_ballot_w64(qc == kTpqQueue) != 0) {
#pragma unroll
      for (int u = 0; u < kTpqQueue; ++u) {
        tpq_insert<KK>(a, qd[u]);
        qd[u] = __builtin_inff();
        __builtin_amdgcn_sched_barrier(0);  // one chain at a time: interleaved chains double the live registers
      }
      qc = 0;
      mx = a[KK - 1];
#ifdef PGDVS_AB_TPQ_STATS
      ++dbg_flush;
#endif
    }
#ifdef PGDVS_AB_TPQ_STATS
    ++dbg_steps;
#endif
    return __builtin_amdgcn_ballot_w64(v_issue || k < 9) != 0;  // anything requested or left to visit?
  };
  for (;;) {
    if (!half_step(cb, vb, ca, va)) break;
    if (!half_step(ca, va, cb, vb)) break;
  }
#pragma unroll
  for (int u = 0; u < kTpqQueue; ++u) {
    tpq_insert<KK>(a, qd[u]);
    __builtin_amdgcn_sched_barrier(0);
  }
  mx = a[KK - 1];
#ifdef PGDVS_AB_TPQ_STATS
  {  // [1] accepted candidates (all lanes), [2] candidates evaluated, [3] flushes x 64, [4] half-steps x 64, [5] max accepted per wave, summed
    int32_t *st = open_count - 18;
    int acc = dbg_acc, cand = dbg_cand, mxa = dbg_acc;
    for (int off = 32; off > 0; off >>= 1) {
      acc += __shfl_xor(acc, off, 64);
      cand += __shfl_xor(cand, off, 64);
      const int o = __shfl_xor(mxa, off, 64);
      mxa = o > mxa ? o : mxa;
    }
    if ((tid & 63) == 0) {
      atomicAdd(&st[1], acc);
      atomicAdd(&st[2], cand);
      atomicAdd(&st[3], dbg_flush);
      atomicAdd(&st[4], dbg_steps);
      atomicAdd(&st[5], mxa);
    }
  }
#endif
  cut = mx < __builtin_inff() && mx == s_thr[tid];  // the threshold left the list short
  const float safe2 = s_safe[tid];
  const bool retry = cut && live && mx < safe2;  // (a list cut at the block's own bound is not complete anyway)
  if (attempt == 1 || __builtin_popcountll(__builtin_amdgcn_ballot_w64(retry)) < 16) break;
  t0 = retry ? fminf(kTpqRetryScale * t0, safe2) : t0;
  }
  if (!live) continue;
  // complete?  (the block's bound: see above)
  const bool done = mx <= s_safe[tid];
  if (!done || cut || n < KK || first_col > 1) {
    open_list[atomicAdd(open_count, 1)] = q;
    continue;
  }
  // mean over columns first_col..K in the 64-slot butterfly order of knn_finish (zeros skipped: x + 0 = x)
  // (clouds with fewer than K+1 points and first_col > 1 took the other search above: one wave-uniform condition here
  // instead of one per column -- 51 hoisted mask pairs spilled scalar registers into vector lanes)
  a[0] = first_col == 1 ? 0.0f : a[0];
  int len = KK;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int i = 0; i < off; ++i)
      if (i + off < len) a[i] = a[i] + a[i + off];
    len = len < off ? len : off;
  }
  avg_out[__float_as_int(qp.w)] = a[0] / (float)(KK - first_col);
  }
}

// first level (all queries) and second level (the listed open queries on the coarse grid): the
// same search, two kernel symbols so that profiles keep them apart
__global__ void __launch_bounds__(256)
grid_query_kernel(const GridParams *__restrict__ gp, const float4 *__restrict__ sorted,
                  const CellIndex cell_start, int KK, QuerySrc qs, float *__restrict__ avg_out,
                  int32_t *__restrict__ stats, int ring_cap, int32_t *__restrict__ fb_count,
                  int32_t *__restrict__ fb_list, float *__restrict__ fb_bound) {
  grid_query_body(gp, sorted, cell_start, KK, qs, avg_out, stats, ring_cap, fb_count, fb_list, fb_bound);
}

__global__ void __launch_bounds__(256)
grid2_query_kernel(const GridParams *__restrict__ gp, const float4 *__restrict__ sorted,
                   const CellIndex cell_start, int KK, QuerySrc qs, float *__restrict__ avg_out,
                   int32_t *__restrict__ stats, int ring_cap, int32_t *__restrict__ fb_count,
                   int32_t *__restrict__ fb_list, float *__restrict__ fb_bound) {
  grid_query_body(gp, sorted, cell_start, KK, qs, avg_out, stats, ring_cap, fb_count, fb_list, fb_bound);
}

// Exact scan of ALL points for the queries the ring search gave up on, split over many
// workgroups: work item (query f, slice s) -- a 256-thread workgroup scans slice s of the
// point array (each of its 4 waves a strided share) and leaves the slice's 64 smallest
// distances in global memory; grid_fallback_merge_kernel folds the kFbSlices lists.
constexpr int kFbSlices = 16;
constexpr int kFbMaxSliced = 16384;  // queries beyond this many use the one-workgroup scan

__global__ void __launch_bounds__(256)
grid_fallback_kernel(const GridParams *__restrict__ gp, const float4 *__restrict__ sorted, int KK,
                     QuerySrc qs, const int32_t *__restrict__ fb_count, const int32_t *__restrict__ fb_list,
                     const float *__restrict__ fb_bound, float *__restrict__ fb_partial) {
  __shared__ float s_best[4][64];
  const int n = gp->n;
  const int nfb = *fb_count;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int slice_len = ((n + kFbSlices - 1) / kFbSlices + 255) / 256 * 256;
  const int nsl = nfb < kFbMaxSliced ? nfb : kFbMaxSliced;
  for (int item = blockIdx.x; item < nsl * kFbSlices; item += gridDim.x) {
    const int f = item / kFbSlices, sl = item - f * kFbSlices;
    const float4 qp = load_query(qs, sorted, fb_list[f]);
    const float bound = fb_bound[f];
    BestList b;
    b.best = __builtin_inff();
    b.mx = __builtin_inff();
    bool first = true;
    const int s_beg = sl * slice_len;
    const int s_end = s_beg + slice_len < n ? s_beg + slice_len : n;
    for (int j0 = s_beg + wave * 256; j0 < s_end; j0 += 4 * 256) {
      int e = j0 + 256 < s_end ? j0 + 256 : s_end;
      scan_range(b, first, sorted, j0, e, qp.x, qp.y, qp.z, KK, lane, bound);
    }
    __syncthreads();
    s_best[wave][lane] = b.best;
    __syncthreads();
    if (wave == 0) {
      for (int w = 1; w < 4; ++w) best_insert_batch(b, s_best[w][lane], true, KK, lane);
      fb_partial[(size_t)item * 64 + lane] = b.best;
    }
  }
}

// ... the merge of the slices' lists, and -- the overflow path: more open queries than kFbMaxSliced (degenerate clouds) -- one
// 1024-thread workgroup per query scanning everything, in ONE launch (round 6; two until then): a wavefront per sliced query
// first, whole workgroups for the queries beyond kFbMaxSliced after that
__global__ void __launch_bounds__(1024)
grid_fallback_finish_kernel(const GridParams *__restrict__ gp, const float4 *__restrict__ sorted, int KK,
                            QuerySrc qs, const int32_t *__restrict__ fb_count, const int32_t *__restrict__ fb_list,
                            const float *__restrict__ fb_bound, const float *__restrict__ fb_partial, float *__restrict__ avg_out,
                            const int32_t *__restrict__ scan_error) {
  __shared__ float s_best[16][64];
  const int n = gp->n;
  if (*scan_error != 0) {
    // a look-back scan of this search gave up (grid_scan_kernel): its cell starts are not trustworthy -- every average becomes
    // NaN, and with it the statistics and the threshold behind them (callers that read the threshold see NaN)
    const int nq = query_count(qs, n);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x) avg_out[i] = __builtin_nanf("");
    return;
  }
  const int nfb = *fb_count;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nsl = nfb < kFbMaxSliced ? nfb : kFbMaxSliced;
  for (int f = blockIdx.x * 16 + wave; f < nsl; f += gridDim.x * 16) {
    BestList b;
    b.best = fb_partial[((size_t)f * kFbSlices) * 64 + lane];  // slice 0 is already sorted
    b.mx = readlane_f(b.best, KK - 1);
    for (int sl = 1; sl < kFbSlices; ++sl)
      best_insert_batch(b, fb_partial[((size_t)f * kFbSlices + sl) * 64 + lane], true, KK, lane);
    knn_finish(b.best, KK, qs.first_col, n, lane, __float_as_int(load_query(qs, sorted, fb_list[f]).w), avg_out);
  }
  for (int f = kFbMaxSliced + blockIdx.x; f < nfb; f += gridDim.x) {
    const float4 qp = load_query(qs, sorted, fb_list[f]);
    const float bound = fb_bound[f];
    BestList b;
    b.best = __builtin_inff();
    b.mx = __builtin_inff();
    bool first = true;
    for (int j0 = wave * 256; j0 < n; j0 += 16 * 256) {
      int e = j0 + 256 < n ? j0 + 256 : n;
      scan_range(b, first, sorted, j0, e, qp.x, qp.y, qp.z, KK, lane, bound);
    }
    __syncthreads();
    s_best[wave][lane] = b.best;
    __syncthreads();
    if (wave == 0) {
      for (int w = 1; w < 16; ++w) best_insert_batch(b, s_best[w][lane], true, KK, lane);
      knn_finish(b.best, KK, qs.first_col, n, lane, __float_as_int(qp.w), avg_out);
    }
  }
}

struct GridWs {
  unsigned *bbox;
  GridParams *gp;
  int32_t *block_sums, *cell_of;
  float4 *samples;         // [sample_blocks][kSampleSlots]: the thinned cloud behind the cell size (grid_sample_kernel)
  int32_t *sample_count;   // [sample_blocks]
  float *sample_d1;        // [kSampleMaxQueries]
  int sample_blocks;
  uint4 *tab;                                   // first-level grid: bit table with ranks (CellIndex)
  int32_t *occ_count, *occ_start, *nocc;        // points per occupied cell, their starts, the number of occupied cells
  float4 *sorted;
  int32_t *fb_count, *fb_list;
  int32_t *open_count, *open_list;  // queries the thread-per-query pass left to the ring search
  float *fb_bound;
  float *fb_partial;  // [kFbMaxSliced][kFbSlices][64]
  unsigned long long *scan_state, *scan_state2, *rank_state;
  int32_t *scan_ticket, *scan_error;
  int scan_tiles;
  int64_t state_bytes;  // the block the call's memset clears
  int32_t *stats;  // [16] ring histogram, filled only when PGDVS_KNN_STATS=1 (diagnostics)
  // second level
  GridParams *gp2;
  int32_t *cell_count2, *cell_start2, *block_sums2;
  float4 *sorted2;
  int32_t *fb2_count, *fb2_list;
  float *fb2_bound;
  int64_t total_bytes;
};

constexpr int kCoarseMaxCells = 1 << 18;
// Cells of the second-level grid in units of the first level's.  8: the few isolated points of a depth-map cloud (0.4 % of the
// benchmark's queries) reach their neighbours within a ring or two.  When MANY queries are still open after the ring search
// (more than 1 % -- a cloud from noisy depth: a slab with flying pixels all around it) each of them would scan 27 cells of
// thousands of points: cells half as wide then (round 5, noisy scene: count + fill + search 0.40 -> 0.23 ms; with the
// benchmark's 1247 open queries the finer cells would cost 17 -> 50 us).  Decided on the device from the open-query count.
constexpr float kCoarseScale = 8.0f, kCoarseScaleMany = 4.0f;

static GridWs grid_ws_layout(void *base, int64_t capacity, int64_t qcapacity) {
  GridWs w;
  char *p = reinterpret_cast<char *>(base);
  int64_t off = 0;
  // one 256-byte state block, zeroed by one memset per call: complemented bbox min, bbox max,
  // the ring histogram and the open-query counters of both levels
  w.bbox = reinterpret_cast<unsigned *>(p + off);
  w.stats = reinterpret_cast<int32_t *>(p + off + 128);
  w.fb_count = reinterpret_cast<int32_t *>(p + off + 192);
  w.fb2_count = reinterpret_cast<int32_t *>(p + off + 196);
  w.open_count = reinterpret_cast<int32_t *>(p + off + 200);
  w.nocc = reinterpret_cast<int32_t *>(p + off + 204);  // [2]: occupied cells, kSub x that
  w.scan_ticket = reinterpret_cast<int32_t *>(p + off + 216);  // [2]: tile tickets of the two scans (grid_scan_kernel)
  w.scan_error = reinterpret_cast<int32_t *>(p + off + 224);   // a look-back spin gave up: results are poisoned (NaN)
  off += 256;
  // ... and the look-back words of the two scans (grid_scan_kernel), one per tile that can occur
  w.scan_state = reinterpret_cast<unsigned long long *>(p + off);
  w.scan_tiles = (int)((((capacity < kGridMaxCells ? (capacity > 0 ? capacity : 1) : (int64_t)kGridMaxCells) + 2) * kSub) / kScanTile + 1);
  off += (int64_t)w.scan_tiles * 8;
  w.scan_state2 = reinterpret_cast<unsigned long long *>(p + off);
  off += (int64_t)(kCoarseMaxCells / kScanTile + 2) * 8;
  w.rank_state = reinterpret_cast<unsigned long long *>(p + off);  // [kRankBlocksMax] tagged bit counts + the ticket word
  off += (int64_t)(kRankBlocksMax + 1) * 8;
  off = align_up(off, 256);
  w.state_bytes = off;
  w.gp = reinterpret_cast<GridParams *>(p + off);
  off += 256;
  w.sample_blocks = (int)(((capacity > 0 ? capacity : 1) + kSampleBlock - 1) / kSampleBlock);
  w.samples = reinterpret_cast<float4 *>(p + off);
  off += align_up((int64_t)w.sample_blocks * kSampleSlots * 16, 256);
  w.sample_count = reinterpret_cast<int32_t *>(p + off);
  off += align_up((int64_t)w.sample_blocks * 4, 256);
  w.sample_d1 = reinterpret_cast<float *>(p + off);
  off += align_up((int64_t)kSampleMaxQueries * 4, 256);
  w.tab = reinterpret_cast<uint4 *>(p + off);
  off += align_up((int64_t)kTabWords * 16, 256);
  {  // occupied cells <= min(points, cells)
    const int64_t occ_cap = ((capacity < kGridMaxCells ? (capacity > 0 ? capacity : 1) : (int64_t)kGridMaxCells) + 2) * kSub;
    w.occ_count = reinterpret_cast<int32_t *>(p + off);
    off += align_up(occ_cap * 4, 256);
    w.occ_start = reinterpret_cast<int32_t *>(p + off);
    off += align_up(occ_cap * 4, 256);
  }
  w.block_sums = reinterpret_cast<int32_t *>(p + off);
  off += align_up((int64_t)(kGridMaxCells / kScanTile) * 4, 256);
  w.cell_of = reinterpret_cast<int32_t *>(p + off);
  off += align_up((capacity > 0 ? capacity : 1) * 4, 256);
  w.sorted = reinterpret_cast<float4 *>(p + off);
  off += align_up((capacity > 0 ? capacity : 1) * 16, 256);
  const int64_t qcap = qcapacity > capacity ? qcapacity : capacity;  // queries that may fall back
  w.fb_list = reinterpret_cast<int32_t *>(p + off);
  off += align_up((qcap > 0 ? qcap : 1) * 4, 256);
  w.fb_bound = reinterpret_cast<float *>(p + off);
  off += align_up((qcap > 0 ? qcap : 1) * 4, 256);
  w.open_list = reinterpret_cast<int32_t *>(p + off);
  off += align_up((capacity > 0 ? capacity : 1) * 4, 256);
  w.fb_partial = reinterpret_cast<float *>(p + off);
  off += align_up((int64_t)kFbMaxSliced * kFbSlices * 64 * 4, 256);
  w.gp2 = reinterpret_cast<GridParams *>(p + off);
  off += 256;
  w.cell_count2 = reinterpret_cast<int32_t *>(p + off);
  off += align_up(((int64_t)kCoarseMaxCells + 1) * 4, 256);
  w.cell_start2 = reinterpret_cast<int32_t *>(p + off);
  off += align_up(((int64_t)kCoarseMaxCells + 1) * 4, 256);
  w.block_sums2 = reinterpret_cast<int32_t *>(p + off);
  off += align_up((int64_t)(kGridMaxCells / kScanTile) * 4, 256);
  w.sorted2 = reinterpret_cast<float4 *>(p + off);
  off += align_up((capacity > 0 ? capacity : 1) * 16, 256);
  w.fb2_list = reinterpret_cast<int32_t *>(p + off);
  off += align_up((qcap > 0 ? qcap : 1) * 4, 256);
  w.fb2_bound = reinterpret_cast<float *>(p + off);
  off += align_up((qcap > 0 ? qcap : 1) * 4, 256);
  w.total_bytes = off;
  return w;
}

// the device-side exit counters the last search on this workspace left behind: queries the thread-per-query pass handed to
// the ring search, queries the ring search handed to the coarse grid, queries scanned exhaustively
void knn_grid_counter_words(const void *workspace, int64_t capacity, int64_t qcapacity, const int32_t **to_ring,
                            const int32_t **to_coarse, const int32_t **to_exhaustive) {
  const GridWs w = grid_ws_layout(const_cast<void *>(workspace), capacity, qcapacity);
  *to_ring = w.open_count;
  *to_coarse = w.fb_count;
  *to_exhaustive = w.fb2_count;
}

int64_t knn_grid_workspace_bytes(int64_t capacity, int64_t qcapacity) {
  return grid_ws_layout(nullptr, capacity, qcapacity).total_bytes;
}

// qpts == nullptr: self mode (mean of columns 1..K).  Otherwise cross mode: mean of the K+1
// smallest distances from each of the *qcount queries to the points.
static int knn_grid_search(const float *pts, const int32_t *count, int64_t capacity, int K, float *avg_out,
                           void *workspace, int64_t workspace_bytes, hipStream_t st, const float *qpts,
                           const int32_t *qcount, int64_t qcapacity, bool prepared);

int knn_grid_mean_dist(const float *pts, const int32_t *count, int64_t capacity, int K, float *avg_out,
                       void *workspace, int64_t workspace_bytes, hipStream_t st, const float *qpts,
                       const int32_t *qcount, int64_t qcapacity) {
  return knn_grid_search(pts, count, capacity, K, avg_out, workspace, workspace_bytes, st, qpts, qcount, qcapacity, false);
}

// fused.h: the per-view call clears the state block and computes the bounding box inside launches it runs anyway
void knn_grid_state_block(void *workspace, int64_t capacity, void **block, int64_t *bytes, unsigned **bbox, void **tab,
                          int64_t *tab_bytes, int32_t **occ_count, int *occ_mult, void **coarse, int64_t *coarse_bytes) {
  const GridWs ws = grid_ws_layout(workspace, capacity, 0);
  *block = ws.bbox;
  *bytes = ws.state_bytes;
  *bbox = ws.bbox;
  *tab = ws.tab;
  *tab_bytes = (int64_t)kTabWords * 16;
  *occ_count = ws.occ_count;
  *occ_mult = kSub;
  *coarse = ws.cell_count2;
  *coarse_bytes = align_up(((int64_t)kCoarseMaxCells + 1) * 4, 16);
}
int knn_grid_mean_dist_prepared(const float *pts, const int32_t *count, int64_t capacity, int K, float *avg_out, void *workspace,
                                int64_t workspace_bytes, hipStream_t st) {
  return knn_grid_search(pts, count, capacity, K, avg_out, workspace, workspace_bytes, st, nullptr, nullptr, 0, true);
}

// prepared: the state block is cleared and holds the points' bounding box already, the bit table and the quarter-cell counters
// are cleared, and so are the second-level grid's counters (no memsets, no grid_bbox launch, no grid_tab_zero launch)
static int knn_grid_search(const float *pts, const int32_t *count, int64_t capacity, int K, float *avg_out,
                           void *workspace, int64_t workspace_bytes, hipStream_t st, const float *qpts,
                           const int32_t *qcount, int64_t qcapacity, bool prepared) {
  GridWs ws = grid_ws_layout(workspace, capacity, qcapacity);
  const int KK = K + 1;
  QuerySrc qs;
  qs.qpts = qpts;
  qs.qcount = qcount;
  qs.first_col = qpts ? 0 : 1;
  qs.list = nullptr;
  qs.list_count = nullptr;
  qs.qsorted = nullptr;
  const int64_t nq_cap = qpts ? qcapacity : capacity;
  if (!workspace || workspace_bytes < ws.total_bytes) {
    set_error("knn_grid: workspace too small");
    return PGDVS_ERR_WORKSPACE;
  }
  // empty box: ~min = 0 and max = 0 in the order-preserving uint encoding
  unsigned gpts = (unsigned)(cdiv(capacity, 256) < 2048 ? cdiv(capacity, 256) : 2048);
  unsigned gbb = gpts < 128 ? gpts : 128;
  const float target = kTargetPerCellDefault * (float)(K + 1) / 51.0f;
  if (!prepared) {
    hipError_t e = hipMemsetAsync(ws.bbox, 0x00, (size_t)ws.state_bytes, st);  // ~min, max, stats, counters, scan words
    if (e != hipSuccess) {
      set_error("knn_grid memset: %s", hipGetErrorString(e));
      return PGDVS_ERR_LAUNCH;
    }
    e = hipMemsetAsync(ws.cell_count2, 0x00, ((size_t)kCoarseMaxCells + 1) * 4, st);  // the second-level grid's counters
    if (e != hipSuccess) {
      set_error("knn_grid memset: %s", hipGetErrorString(e));
      return PGDVS_ERR_LAUNCH;
    }
    PGDVS_LAUNCH("grid_bbox", grid_bbox_kernel, dim3(gbb), dim3(256), 0, st, pts, count, ws.bbox);
  }
  // the cell size from the spacing of a thinned copy of the cloud (see grid_params_kernel)
  PGDVS_LAUNCH("grid_sample", grid_sample_kernel, dim3((unsigned)ws.sample_blocks), dim3(256), 0, st, pts, count, ws.samples,
               ws.sample_count);
  PGDVS_LAUNCH("grid_sample_nn", grid_sample_nn_kernel, dim3(kSampleMaxQueries), dim3(256), 0, st, count,
               (const float4 *)ws.samples, (const int32_t *)ws.sample_count, ws.sample_d1);
  PGDVS_LAUNCH("grid_params", grid_params_kernel, dim3(1), dim3(1024), 0, st, ws.bbox, count, (const float *)ws.sample_d1, ws.gp,
               sqrtf((float)(K + 1) / 51.0f), target, 0.5f);
  // the final grid as a sparse cell index (CellIndex): bit table, ranks, points per occupied cell, their starts
  if (!prepared) PGDVS_LAUNCH("grid_tab_zero", grid_tab_zero_kernel, dim3(512), dim3(256), 0, st, ws.gp, ws.tab, ws.occ_count);
  PGDVS_LAUNCH("grid_mark", grid_mark_kernel, dim3(gpts), dim3(256), 0, st, pts, ws.gp, ws.cell_of, ws.tab);
  const unsigned nrb = (unsigned)cdiv(kTabWords, kRankWordsPerBlock);
  PGDVS_LAUNCH("grid_rank", grid_rank_kernel, dim3(nrb), dim3(1024), 0, st, ws.gp, ws.tab, ws.rank_state,
               reinterpret_cast<int32_t *>(ws.rank_state + kRankBlocksMax), ws.scan_error, ws.nocc);
  PGDVS_LAUNCH("grid_occ_count", grid_occ_count_kernel, dim3(gpts), dim3(256), 0, st, ws.gp, ws.cell_of,
               (const uint4 *)ws.tab, ws.occ_count);
  const int64_t occ_cap = capacity < kGridMaxCells ? capacity : (int64_t)kGridMaxCells;
  const int nb = (int)cdiv(occ_cap + 1, kScanTile);  // (<= kGridMaxCells / kScanTile = 1024: one pass over the block sums)
  (void)nb;
  PGDVS_LAUNCH("grid_scan", grid_scan_kernel, dim3((unsigned)(ws.scan_tiles < kScanMaxBlocks ? ws.scan_tiles : kScanMaxBlocks)), dim3(1024), 0, st,
               ws.occ_count, (const int32_t *)(ws.nocc + 1), ws.occ_start, ws.scan_state, ws.scan_ticket, ws.scan_error);
  PGDVS_LAUNCH("grid_fill", grid_fill_kernel, dim3(gpts), dim3(256), 0, st, pts, ws.gp, ws.cell_of,
               ws.occ_start, ws.occ_count, ws.sorted, (const int32_t *)nullptr);
  CellIndex ci;
  ci.start = ws.occ_start;
  ci.tab = ws.tab;
  const bool want_stats = option_int(options().knn_stats) != 0;
  int32_t *stats = want_stats ? ws.stats : nullptr;
#ifdef PGDVS_AB_TPQ_STATS
  stats = nullptr;
#endif
  const unsigned gq = (unsigned)(cdiv(nq_cap, 4) < 256 * 8 ? (cdiv(nq_cap, 4) > 0 ? cdiv(nq_cap, 4) : 1) : 256 * 8);
  // (option knn_no_tpq: diagnostics and tests, the wavefront-per-query search for every query)
  bool tpq = qpts == nullptr && option_int(options().knn_no_tpq) == 0 && capacity < (1ll << 27);  // (32-bit byte offsets)
  // starting threshold of the thread-per-query pass in units of the block's estimate (any value is exact).  Round 5, rows
  // cut to this threshold's ball: 5.2 -> 166 us, 4.5 -> 158, 4.0 -> 152 (+1.5 us of ring search), 3.7 -> 210, 3.4 -> 303
  // (a quarter of a wavefront's lists short = the whole wavefront searches again): 4.5 keeps its distance from that edge
  const float thr_mult = 4.5f;
  if (tpq) {
    const unsigned gt = (unsigned)(cdiv(capacity, 256) < 2560 ? (cdiv(capacity, 256) > 0 ? cdiv(capacity, 256) : 1) : 2560);
    switch (KK) {
#define PGDVS_TPQ_CASE(N)                                                                                        \
  case N:                                                                                                        \
    PGDVS_LAUNCH("grid_query_tpq", grid_query_tpq_kernel<N>, dim3(gt), dim3(256), 0, st, ws.gp, ws.sorted,      \
                 ci, qs.first_col, avg_out, ws.open_count, ws.open_list, thr_mult);                              \
    break;
      PGDVS_TPQ_CASE(5)
      PGDVS_TPQ_CASE(9)
      PGDVS_TPQ_CASE(17)
      PGDVS_TPQ_CASE(21)
      PGDVS_TPQ_CASE(51)
#undef PGDVS_TPQ_CASE
      default:
        tpq = false;
    }
  }
  const int ring_cap_after_tpq = kRingCapAfterTpq;
  QuerySrc qs1 = qs;
  if (tpq) {  // the ring search only sees what the first pass left open
    qs1.list = ws.open_list;
    qs1.list_count = ws.open_count;
  }
  PGDVS_LAUNCH("grid_query", grid_query_kernel, dim3(gq), dim3(256), 0, st, ws.gp,
               ws.sorted, ci, KK, qs1, avg_out, stats, tpq ? ring_cap_after_tpq : kRingCap, ws.fb_count, ws.fb_list,
               ws.fb_bound);
  // Second level: the queries still open after kRingCap rings (isolated points, far from
  // everything in units of the cell size) repeat the ring search on a grid with
  // kCoarseScale-times larger cells before anything is scanned exhaustively.  All of it is
  // gated on the device-side count of open queries.
  const int nb2 = kCoarseMaxCells / kScanTile;
  PGDVS_LAUNCH("grid2_count", grid_coarse_count_kernel, dim3(gpts), dim3(256), 0, st, pts, ws.gp, ws.bbox, ws.gp2, kCoarseScale,
               kCoarseScaleMany, (const int32_t *)ws.fb_count, qpts ? qcount : (const int32_t *)nullptr, kCoarseMaxCells, ws.cell_of,
               ws.cell_count2);
  (void)nb2;
  PGDVS_LAUNCH("grid2_scan", grid_scan_kernel, dim3(kCoarseMaxCells / kScanTile + 1), dim3(1024), 0, st, ws.cell_count2,
               (const int32_t *)&ws.gp2->ncells, ws.cell_start2, ws.scan_state2, ws.scan_ticket + 1, ws.scan_error);
  PGDVS_LAUNCH("grid2_fill", grid_fill_kernel, dim3(gpts), dim3(256), 0, st, pts, ws.gp2, ws.cell_of,
               ws.cell_start2, ws.cell_count2, ws.sorted2, (const int32_t *)ws.fb_count);
  CellIndex ci2;  // the coarse grid stays dense (<= 256 K cells)
  ci2.start = ws.cell_start2;
  ci2.tab = nullptr;
  QuerySrc qs2 = qs;
  qs2.list = ws.fb_list;
  qs2.list_count = ws.fb_count;
  qs2.qsorted = ws.sorted;
  PGDVS_LAUNCH("grid2_query", grid2_query_kernel, dim3(gq < 1024 ? gq : 1024), dim3(256), 0, st, ws.gp2, ws.sorted2,
               ci2, KK, qs2, avg_out, (int32_t *)nullptr, kRingCap, ws.fb2_count, ws.fb2_list,
               ws.fb2_bound);
  // exhaustive scan for what is left (rare)
  PGDVS_LAUNCH("grid_fallback", grid_fallback_kernel, dim3(4096), dim3(256), 0, st, ws.gp, ws.sorted, KK,
               qs, ws.fb2_count, ws.fb2_list, ws.fb2_bound, ws.fb_partial);
  PGDVS_LAUNCH("grid_fallback_finish", grid_fallback_finish_kernel, dim3(256), dim3(1024), 0, st, ws.gp, ws.sorted, KK, qs,
               ws.fb2_count, ws.fb2_list, ws.fb2_bound, ws.fb_partial, avg_out, (const int32_t *)ws.scan_error);
#ifdef PGDVS_AB_TPQ_STATS
  if (want_stats) stats = ws.stats;
#endif
  if (stats) {  // diagnostics only (PGDVS_KNN_STATS=1): synchronises and prints the ring histogram
    int32_t hst[16], nfb = 0, nfb2 = 0;
    GridParams g1, g2;
    if (hipStreamSynchronize(st) == hipSuccess &&
        hipMemcpy(hst, ws.stats, sizeof(hst), hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(&nfb, ws.fb_count, 4, hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(&nfb2, ws.fb2_count, 4, hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(&g1, ws.gp, sizeof(g1), hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(&g2, ws.gp2, sizeof(g2), hipMemcpyDeviceToHost) == hipSuccess) {
      fprintf(stderr, "[knn_grid] n=%d d1=%g(%d) h=%g G=%dx%dx%d (%d cells); level2 h=%g %dx%dx%d; rings:", g1.n, g1.sample_dist, g1.sample_dists, g1.h, g1.G[0],
              g1.G[1], g1.G[2], g1.ncells, g2.h, g2.G[0], g2.G[1], g2.G[2]);
      for (int i = 1; i < 16; ++i) fprintf(stderr, " %d", hst[i]);
      fprintf(stderr, "; to level 2: %d, exhaustive: %d\n", nfb, nfb2);
    }
  }
  return check_launch("knn_grid");
}

}  // namespace pgdvs
