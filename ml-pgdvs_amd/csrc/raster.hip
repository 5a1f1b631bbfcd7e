// A9: point z-buffer rasteriser + norm-weighted compositor.
// Semantics: pytorch3d 0.7.4 PointsRasterizer(bin_size=0) -> PointsRenderer ->
// NormWeightedCompositor as driven by StaticGeoPointRenderer.forward
// (pgdvs/renderers/st_geo_renderer.py:77-120) and render_dyn_pcl
// (pgdvs/renderers/pgdvs_renderer_dyn.py:671-724); camera convention from
// pgdvs/utils/pytorch3d_utils.py:5-47.
//
// The reference path is pytorch3d's NAIVE rasteriser: every pixel scans every point,
// O(H*W*N).  Here points are binned to 16x16-pixel tiles first (count -> scan -> fill; a list
// entry is 16 bytes: NDC x, y, point id, view z, so that the tile pass never gathers), then one
// 256-thread workgroup per tile.  Selection uses the total order (z, id), which is also
// pytorch3d's priority-queue order, so idx/zbuf/dist2 do not depend on list order.
//   sorted path (a tile's list fits the LDS capacity -- the normal case): the workgroup sorts its
//     list by (z, id) in LDS -- one distribution pass into 1024 monotone z-buckets, exact ranks
//     inside a bucket by counting -- and a point's RANK becomes its 32-bit key.  Every wavefront
//     (one 8x8 quadrant, pixel per lane) then walks the sorted list front to back: cull by the
//     distance of the disc centre to the quadrant's rectangle of pixel centres (ballot), per-pixel
//     disc test, and a branch-free insertion of the rank (carried as a float) into the pixel's K
//     smallest (K v_med3_f32: 8 vector instructions per tested point for K=3, no divergence),
//     leaving as soon as every pixel of the quadrant holds K points: nothing behind can matter.
//     2048 entries x 16 bytes + the strips = 37 KB of LDS: four workgroups per CU.  (Measured before this form, per wavefront at 1080p x 3.5 M points: 292 tested
//     points of which 167 diverged into a 13-instruction 64-bit insertion, 1130 staged entries; in
//     sorted order 228 are tested and 541 staged.)
//   general path (longer lists, or > 64 equal-depth points in one bucket): lists staged through
//     LDS in chunks, hierarchical-z cull, 64-bit (z,id) keys kept sorted per pixel.
// Tiles are drawn in launch order (until round 5: one contiguous band of tiles per XCD -- see raster_tile_kernel).
#include <cstdlib>

#include "common.h"
#include "raster_cam.h"

namespace pgdvs {

constexpr int kTile = 16;

struct TileBox {
  int tx0, tx1, ty0, ty1;  // inclusive; empty if tx0 > tx1
};

// conservative tile bounding box of the disc of NDC radius `radius` around (x,y)
__device__ __forceinline__ TileBox tile_box(const RasterCam &rc, float3 p, float radius, int H,
                                            int W, int ntx, int nty) {
  TileBox b;
  b.tx0 = 1;
  b.tx1 = 0;
  b.ty0 = 1;
  b.ty1 = 0;
  if (!(p.z >= 0.0f) || !isfinite(p.x) || !isfinite(p.y)) return b;
  // invert PixToNonSquareNdc (pixel index reversed: xidx = W-1-xi)
  float offx = rc.range_x / 2.0f, offy = rc.range_y / 2.0f;
  float xc = (float)(W - 1) - ((p.x + offx) * (float)W - offx) / rc.range_x;
  float yc = (float)(H - 1) - ((p.y + offy) * (float)H - offy) / rc.range_y;
  // xc, yc are in pixel-index units (pixel i is centred at i): the disc touches the pixels with
  // |i - xc| < radius in pixels, i.e. the integers of [ceil(xc - r), floor(xc + r)].  The margin only has to cover the
  // rounding of this inversion against the forward pix_to_ndc used by the tile kernel (~1e-3 px at 4k).  (A margin of
  // 1.5 px put 24 % more entries into the tile lists; until round 5 the interval was [floor(xc - r), ceil(xc + r)], one
  // pixel more on either side: PGDVS_AB_BOX_WIDE builds.)
  float rpx = radius * (float)W / rc.range_x + 0.0625f;
  float rpy = radius * (float)H / rc.range_y + 0.0625f;
#ifdef PGDVS_AB_BOX_WIDE
  float x0 = floorf(xc - rpx), x1 = ceilf(xc + rpx);
  float y0 = floorf(yc - rpy), y1 = ceilf(yc + rpy);
#else
  float x0 = ceilf(xc - rpx), x1 = floorf(xc + rpx);
  float y0 = ceilf(yc - rpy), y1 = floorf(yc + rpy);
  if (x1 < x0 || y1 < y0) return b;  // (a disc smaller than a pixel between two pixel centres)
#endif
  if (x1 < 0.0f || y1 < 0.0f || x0 > (float)(W - 1) || y0 > (float)(H - 1)) return b;
  int ix0 = x0 < 0.0f ? 0 : (int)x0, iy0 = y0 < 0.0f ? 0 : (int)y0;
  int ix1 = x1 > (float)(W - 1) ? W - 1 : (int)x1, iy1 = y1 > (float)(H - 1) ? H - 1 : (int)y1;
  b.tx0 = ix0 / kTile;
  b.tx1 = ix1 / kTile;
  b.ty0 = iy0 / kTile;
  b.ty1 = iy1 / kTile;
  (void)ntx;
  (void)nty;
  return b;
}


// maximum over the wavefront as a scalar: DPP butterflies inside each row of 16 lanes, row
// broadcasts across rows (the total lands in lane 63), no LDS round trips
__device__ __forceinline__ int wave_max_i32_scalar(int v) {
  int t;
  t = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
  v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false);  // row_half_mirror
  v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false);  // row_mirror
  v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v = t > v ? t : v;
  t = __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  v = t > v ? t : v;
  return __builtin_amdgcn_readlane(v, 63);
}

// the tile box of every point spans at most 2 x 2 tiles: disc diameter + the margins of tile_box within one tile side
__device__ __forceinline__ bool tile_box_within_2x2(const RasterCam &rc, float radius, int H, int W) {
  const float rpx = radius * (float)W / rc.range_x + 0.0625f, rpy = radius * (float)H / rc.range_y + 0.0625f;
  return 2.0f * rpx + 2.0f < (float)kTile && 2.0f * rpy + 2.0f < (float)kTile;
}



constexpr int kRasterMaxK = 8;



// ---- a conservative depth bound per tile, BEFORE the binning (round 4) ---------------------------------------------
// Behind K points whose discs certainly cover every pixel of a tile nothing can enter any of the tile's per-pixel lists:
// the binning passes drop such (point, tile) pairs, and the tile pass neither loads nor sorts them.
//   pass 0a: zmin[pixel] = smallest view depth among the points whose CENTRE is nearest to that pixel (atomic minimum on the
//            depth's bit pattern: depths are >= 0, so the patterns order like the values);
//   pass 0b: per b x b block of pixels (b = 4, or 2 for small radii) the K-th smallest of its per-pixel minima -- K
//            different points, each within (b - 1/2) sqrt(2) pixels of every pixel centre of the block, which is inside
//            every disc when the radius exceeds that by the margin the host checks -- and per tile the maximum over its
//            blocks (+inf when a block holds fewer than K such points).
// A pair is dropped only for z > bound (strictly): the K covering points then precede it in the (z, id) order at every
// pixel of the tile, whatever the ids.  On a smooth surface seen obliquely this removes a quarter of the entries (the far
// side of every tile's list: tools/raster_prune_sim.py, 0.76 kept on the benchmark's cloud); on clouds from noisy depth,
// where the depth order inside a disc is random, five sixths (0.17 kept) -- the statistics under which lists otherwise
// grow with every source frame.
// The bound pays for itself only where lists are long: it costs a pass over the cloud with one atomic per point (25-40 us
// at 3.5 M points), and on the benchmark's nominal cloud (1.7 points per pixel, depth order set by the surface's slope) the
// quarter of the entries it removes is worth 12 us of the tile pass.  `gate_rows` (a density: rows >= gate x pixels, the count
// is device-side) switches the three passes and the tests in the binning on together; below it the passes leave at once.
__device__ __forceinline__ int64_t raster_rows(int64_t n_host, const int64_t *__restrict__ n_dev) {
  int64_t n = n_dev ? *n_dev : n_host;
  return n > n_host ? n_host : (n < 0 ? 0 : n);
}

__global__ void __launch_bounds__(256)
raster_zmin_init_kernel(int64_t n_host, const int64_t *__restrict__ n_dev, int64_t gate_rows, uint4 *__restrict__ zmin16, int64_t n16) {
  if (raster_rows(n_host, n_dev) < gate_rows) return;
  const uint4 v = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) zmin16[i] = v;
}

__global__ void __launch_bounds__(256)
raster_zmin_kernel(const float *__restrict__ pts, int64_t pts_stride, int64_t n_host, const int64_t *__restrict__ n_dev,
                   int64_t gate_rows, const float *__restrict__ cam, int H, int W, unsigned *__restrict__ zmin) {
  const int64_t n = raster_rows(n_host, n_dev);
  if (n < gate_rows) return;
  const RasterCam rc = make_raster_cam(cam, H, W);
  const float offx = rc.range_x / 2.0f, offy = rc.range_y / 2.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float *X = pts + i * pts_stride;
    const float3 p = point_to_ndc(rc, X[0], X[1], X[2]);
    if (!(p.z >= 0.0f) || !isfinite(p.x) || !isfinite(p.y)) continue;
    // the centre in pixel-index units, exactly as tile_box inverts PixToNonSquareNdc
    const float xc = (float)(W - 1) - ((p.x + offx) * (float)W - offx) / rc.range_x;
    const float yc = (float)(H - 1) - ((p.y + offy) * (float)H - offy) / rc.range_y;
    const float xr = rintf(xc), yr = rintf(yc);
    if (!(xr >= 0.0f && xr <= (float)(W - 1) && yr >= 0.0f && yr <= (float)(H - 1))) continue;
    // (round 6: look before the atomic -- a minimum only falls, so a value read earlier, however stale, that is already <= this
    // depth makes the atomic a no-op; with three or more points per pixel most of them are)
    unsigned *cell = &zmin[(size_t)(int)yr * W + (int)xr];
    const unsigned zb = __float_as_uint(p.z + 0.0f);
    if (zb < *cell) atomicMin(cell, zb);  // (107 -> 84 us on the noisy-depth scene's 6.5 M points; an atomic load instead: the same)
  }
}

// one workgroup per tile, one thread per pixel; kB x kB pixels per block = kB * kB adjacent lanes
template <int kB>
__global__ void __launch_bounds__(256)
raster_bound_kernel(const unsigned *__restrict__ zmin, int64_t n_host, const int64_t *__restrict__ n_dev, int64_t gate_rows, int H,
                    int W, int ntx, int K, float *__restrict__ tile_bound) {
  if (raster_rows(n_host, n_dev) < gate_rows) return;  // (the binning passes apply the same gate: the bounds are not read)
  constexpr int kG = kB * kB;            // lanes per block
  constexpr int kPerRow = kTile / kB;    // blocks per tile row
  __shared__ unsigned s_max;
  const int tile = blockIdx.x, ty = tile / ntx, tx = tile - ty * ntx;
  const int tid = threadIdx.x, lane = tid & 63;
  const int blk = tid / kG, j = tid % kG;
  const int x = tx * kTile + (blk % kPerRow) * kB + (j % kB), y = ty * kTile + (blk / kPerRow) * kB + (j / kB);
  const bool inside = x < W && y < H;
  const unsigned v = inside ? zmin[(size_t)y * W + x] : 0xffffffffu;
  if (tid == 0) s_max = 0u;
  __syncthreads();
  // rank of this lane's minimum among its block's (ties by lane), and whether the block holds a pixel of the image
  const int base = lane & ~(kG - 1);
  int rank = 0;
  bool any_inside = inside;
#pragma unroll
  for (int o = 1; o < kG; ++o) {
    const int other = base | ((lane + o) & (kG - 1));
    const unsigned w = (unsigned)__shfl((int)v, other, 64);
    any_inside |= (bool)__shfl((int)inside, other, 64);
    rank += (w < v) | ((w == v) & (other < lane)) ? 1 : 0;
  }
  if (any_inside) {
    if (K > kG) {
      if (j == 0) atomicMax(&s_max, 0x7f800000u);
    } else if (rank == K - 1) {
      atomicMax(&s_max, v < 0x7f800000u ? v : 0x7f800000u);  // (no point: +inf -- nothing may be dropped from this tile)
    }
  }
  __syncthreads();
  if (tid == 0) tile_bound[tile] = __uint_as_float(s_max);
}

// Binning, pass 1: entries per tile.  A workgroup takes one contiguous chunk of the cloud (a few rows of
// one source frame: a few hundred tiles) and counts into an LDS table first, so that a tile costs the
// chunk one global atomic instead of one per run of lanes.  Images with more tiles than the table holds
// count straight into global memory.
constexpr int kBinSlots = 16384;  // 64 KB of LDS counters: up to 2048 x 2048 pixels
constexpr int kBinThreads = 1024;

__device__ __forceinline__ void wave_tile_count_lds(int *s_tab, int t) {
  RunInfo r = wave_runs(t);
  if (r.is_leader && t >= 0) atomicAdd(&s_tab[t], r.length);
}

// (kSlots: 8192 covers 1080p and gives five workgroups of the fill pass per CU; 16384 for larger images)
template <int kSlots>
__global__ void __launch_bounds__(kBinThreads)
raster_project_count_kernel(const float *__restrict__ pts, int64_t pts_stride, int64_t n_host,
                            const int64_t *__restrict__ n_dev, const float *__restrict__ cam,
                            float radius, int H, int W, int ntx, int nty,
                            int32_t *__restrict__ tile_count, int32_t *__restrict__ status,
                            const float *__restrict__ tile_bound, int64_t gate_rows, const int32_t *__restrict__ run_flag) {
  __shared__ int s_tab[kSlots];  // one counter per tile (unused when the image has more tiles)
  // (exact path behind the direct binning: only when a tile's segment overflowed -- see raster_fill_kernel)
  if (run_flag != nullptr && *run_flag == 0) return;
  // a device-side count never exceeds the rows the caller sized the workspace for (and a
  // negative one -- the aggregation's error status -- renders nothing)
  int64_t n = n_dev ? *n_dev : n_host;
  // the status word of pgdvs_points_raster_bounded: 1 = the device count exceeds the row bound the workspace was sized
  // for (the rows beyond it are NOT drawn), 2 = the count is negative (the producer's own error status), 0 = fine
  if (status != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *status = n > n_host ? 1 : (n < 0 ? 2 : 0);
  n = n > n_host ? n_host : (n < 0 ? 0 : n);
  if (n < gate_rows) tile_bound = nullptr;
  RasterCam rc = make_raster_cam(cam, H, W);
  const int ntiles = ntx * nty;
  const bool local = ntiles <= kSlots;
  const bool box2 = tile_box_within_2x2(rc, radius, H, W);
  if (local) {
    for (int t = threadIdx.x; t < ntiles; t += kBinThreads) s_tab[t] = 0;
    __syncthreads();
  }
  const int64_t chunk = ((n + gridDim.x - 1) / gridDim.x + 63) / 64 * 64;  // whole wavefronts stay in the loop
  const int64_t lo = (int64_t)blockIdx.x * chunk;
  const int64_t hi = lo + chunk;
  // the next point's coordinates are requested before this one's counters are touched
  float nxt[3] = {0.0f, 0.0f, 0.0f};
  if (lo + threadIdx.x < n) {
    const float *X = pts + (lo + threadIdx.x) * pts_stride;
    nxt[0] = X[0]; nxt[1] = X[1]; nxt[2] = X[2];
  }
  for (int64_t i = lo + threadIdx.x; i < hi; i += kBinThreads) {
    const float cur[3] = {nxt[0], nxt[1], nxt[2]};
    if (i + kBinThreads < hi && i + kBinThreads < n) {
      const float *X = pts + (i + kBinThreads) * pts_stride;
      nxt[0] = X[0]; nxt[1] = X[1]; nxt[2] = X[2];
    }
    TileBox b;
    b.tx0 = 1; b.tx1 = 0; b.ty0 = 1; b.ty1 = 0;
    float pz = 0.0f;
    if (i < n) {
      float3 p = point_to_ndc(rc, cur[0], cur[1], cur[2]);
      b = tile_box(rc, p, radius, H, W, ntx, nty);
      pz = p.z + 0.0f;
    }
    // (a disc narrower than a tile side touches at most 2 x 2 tiles: four fixed rounds instead of two wave-wide
    // maxima -- 2 x 26 instructions, as many as the projection -- to find the round count)
    int nx = 2, ny = 2;
    if (!box2) {
      nx = wave_max_i32_scalar(b.tx1 - b.tx0 + 1);
      ny = wave_max_i32_scalar(b.ty1 - b.ty0 + 1);
    }
    for (int jy = 0; jy < ny; ++jy)
      for (int jx = 0; jx < nx; ++jx) {
        int tx = b.tx0 + jx, ty = b.ty0 + jy;
        int t = (tx <= b.tx1 && ty <= b.ty1) ? ty * ntx + tx : -1;
        if (tile_bound != nullptr && t >= 0 && pz > tile_bound[t]) t = -1;  // behind K covering points: see raster_bound_kernel
        if (local)
          wave_tile_count_lds(s_tab, t);
        else
          wave_tile_count(tile_count, t);
      }
  }
  if (local) {
    __syncthreads();
    for (int t = threadIdx.x; t < ntiles; t += kBinThreads) {
      const int v = s_tab[t];
      if (v) __hip_atomic_fetch_add(&tile_count[t], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// single-block exclusive scan: offsets[i] = sum_{j<i} counts[j], offsets[n] = total.  Eight consecutive counts
// per thread (1080p: 8160 tiles = one round, one barrier pair)
__global__ void __launch_bounds__(1024)
raster_scan_kernel(const int32_t *__restrict__ counts, int n, int32_t *__restrict__ offsets, const int32_t *__restrict__ run_flag) {
  __shared__ int wave_sums[1024 / kWave];
  if (run_flag != nullptr && *run_flag == 0) return;
  constexpr int kPer = 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int carry = 0;  // (every thread keeps its own copy)
  for (int start = 0; start < n; start += 1024 * kPer) {
    const int i0 = start + (int)threadIdx.x * kPer;
    int v[kPer], s = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      v[k] = i0 + k < n ? counts[i0 + k] : 0;
      s += v[k];
    }
    int x = s;
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_sums[wave] = x;
    __syncthreads();
    int run = carry + x - s, total = 0;
#pragma unroll
    for (int w = 0; w < 1024 / kWave; ++w) {
      if (w < wave) run += wave_sums[w];
      total += wave_sums[w];
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      if (i0 + k < n) offsets[i0 + k] = run;
      run += v[k];
    }
    carry += total;
    __syncthreads();  // wave_sums is rewritten by the next round
  }
  if (threadIdx.x == 0) offsets[n] = carry;
}

// Binning, pass 2: the list entries.  Per chunk of kFillThreads * kFillPer points a workgroup hands every
// (point, tile) pair its rank among the chunk's entries of that tile from the LDS table (ranks kept in
// registers), reserves one contiguous range per tile it touched with a single global atomic, and writes the
// entries at range start + rank: a chunk's entries of one tile are adjacent in the list (whole-line writes
// instead of 16-byte appends interleaved between workgroups).
// (round 5: 4 rows per thread instead of 8 -- the pass is bound by the latency of a chunk's dependent phases (cold loads,
// LDS ranks, one returning global atomic per touched tile, stores), not by its bytes: 72 -> 57 us alone; with the list of
// touched tiles below 50 us; 2 rows: 52, 16 rows: 114.  The throughput of the benchmark's loop does not move: other lanes'
// kernels were already running in this pass's shadow.)
constexpr int kFillPer = 4;
constexpr int kFillThreads = 256;
constexpr int kFillMaxSpan = 2;  // tile box sides kept in registers; wider boxes (large radii) take the slow path

// Tiles a chunk touches, in the order of their first reservation (kFillTouchCap of them; a chunk of 1024 consecutive rows
// of a depth-map cloud touches 50-200 of the 8160 tiles of a 1080p frame): the passes over the table -- one global
// reservation per tile, clearing it for the next chunk -- then walk this list instead of every slot.
constexpr int kFillTouchCap = 1024;

__device__ __forceinline__ int wave_tile_reserve_lds(int *s_tab, int t, unsigned short *s_touch, int *s_ntouch) {
  const int lane = threadIdx.x & 63;
  RunInfo r = wave_runs(t);
  int base = 0;
  if (r.is_leader && t >= 0) {
    base = atomicAdd(&s_tab[t], r.length);
    if (base == 0) {  // the chunk's first entry of this tile
      const int at = atomicAdd(s_ntouch, 1);
      if (at < kFillTouchCap) s_touch[at] = (unsigned short)t;
    }
  }
  base = __shfl(base, r.leader, 64);
  return base + (lane - r.leader);
}

template <int kSlots>
__global__ void __launch_bounds__(kFillThreads)
raster_fill_kernel(const float *__restrict__ pts, int64_t pts_stride, int64_t n_host, const int64_t *__restrict__ n_dev,
                   const float *__restrict__ cam, float radius, int H, int W, int ntx, int nty,
                   const int32_t *__restrict__ offsets,
                   int32_t *__restrict__ cursor, float4 *__restrict__ lists,
                   int64_t list_capacity, const float *__restrict__ tile_bound, int64_t gate_rows,
                   int seg, int32_t *__restrict__ overflow, int32_t *__restrict__ status, const int32_t *__restrict__ run_flag) {
  __shared__ int s_tab[kSlots];  // one counter per tile (unused on the slow path)
  __shared__ unsigned short s_touch[kFillTouchCap];
  __shared__ int s_ntouch;
  static_assert(kSlots <= 65536, "tile numbers in the touched list are 16 bits wide");
  int n_touched = kFillTouchCap + 1;  // of the previous chunk (> cap: the whole table is cleared)
  // Round 5, DIRECT binning (seg > 0): no counting pass and no scan in front of this one -- every tile owns a fixed
  // segment of `seg` entries (lists + t * seg), a chunk reserves its range with the same global atomic, and a range
  // that would leave the segment raises `overflow` (its entries are dropped): the exact count / scan / fill passes are
  // enqueued behind this launch either way and return at once unless the flag is up (run_flag), the tile pass reads
  // whichever lists are valid.  At 1080p x 24 frames the longest list holds ~1900 entries of 4096; what the direct
  // pass saves is the counting pass's second projection of every point (24 us + the 5 us scan per view).
  if (run_flag != nullptr && *run_flag == 0) return;
  // a device-side count never exceeds the rows the caller sized the workspace for (and a
  // negative one -- the aggregation's error status -- renders nothing)
  int64_t n = n_dev ? *n_dev : n_host;
  // (direct mode: the status word of pgdvs_points_raster_bounded, written by the counting pass otherwise)
  if (status != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *status = n > n_host ? 1 : (n < 0 ? 2 : 0);
  if (seg > 0 && blockIdx.x == 0 && threadIdx.x == 0) overflow[1] = 1;  // (stats[2]: this call's lists may be the segments)
  n = n > n_host ? n_host : (n < 0 ? 0 : n);
  if (n < gate_rows) tile_bound = nullptr;
  RasterCam rc = make_raster_cam(cam, H, W);
  const int ntiles = ntx * nty;
  // tile boxes of at most 2 x 2 tiles (disc diameter + margins within one tile side) and a table that holds
  // every tile: otherwise the entries are appended one run of lanes at a time
  const bool local = ntiles <= kSlots && tile_box_within_2x2(rc, radius, H, W);
  constexpr int64_t kChunk = (int64_t)kFillThreads * kFillPer;
  for (int64_t c0 = (int64_t)blockIdx.x * kChunk; c0 < n; c0 += (int64_t)gridDim.x * kChunk) {
    int t0[kFillPer], span[kFillPer];
    float ex[kFillPer], ey[kFillPer], ez[kFillPer];
    // all loads of the chunk first (the LDS atomics below would otherwise fence them one behind the other);
    // projected again rather than read back: the same instructions on the same inputs as in the counting
    // pass give the same tile box, and 32 bytes per point of intermediate traffic disappear
#pragma unroll
    for (int u = 0; u < kFillPer; ++u) {
      const int64_t i = c0 + (int64_t)u * kFillThreads + threadIdx.x;
      const float *X = pts + (i < n ? i : n - 1) * pts_stride;
      ex[u] = X[0];
      ey[u] = X[1];
      ez[u] = X[2];
    }
    if (local) {
      if (n_touched <= kFillTouchCap) {
        for (int k = threadIdx.x; k < n_touched; k += kFillThreads) s_tab[s_touch[k]] = 0;
      } else {
        for (int t = threadIdx.x; t < ntiles; t += kFillThreads) s_tab[t] = 0;
      }
      if (threadIdx.x == 0) s_ntouch = 0;
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < kFillPer; ++u) {
      const int64_t i = c0 + (int64_t)u * kFillThreads + threadIdx.x;
      TileBox b;
      b.tx0 = 1; b.tx1 = 0; b.ty0 = 1; b.ty1 = 0;
      if (i < n) {
        const float3 q = point_to_ndc(rc, ex[u], ey[u], ez[u]);
        b = tile_box(rc, q, radius, H, W, ntx, nty);
        ex[u] = q.x;
        ey[u] = q.y;
        ez[u] = q.z + 0.0f;  // -0.0 -> +0.0 once, so that the z bits order like the values
      }
      t0[u] = b.ty0 * ntx + b.tx0;
      span[u] = (b.tx1 - b.tx0 + 1) | ((b.ty1 - b.ty0 + 1) << 16);  // (0, 0) for a point that touches no tile
      // tiles of the 2 x 2 box (bits 8-11: jy * 2 + jx) whose depth bound this point lies behind -- the same test on the
      // same values as in the counting pass
      if (tile_bound != nullptr && local) {
        unsigned behind = 0;
#pragma unroll
        for (int jy = 0; jy < kFillMaxSpan; ++jy)
#pragma unroll
          for (int jx = 0; jx < kFillMaxSpan; ++jx)
            if (jx < (span[u] & 0xff) && jy < (span[u] >> 16) && ez[u] > tile_bound[t0[u] + jy * ntx + jx]) behind |= 1u << (8 + jy * 2 + jx);
        span[u] |= (int)behind;
      }
    }
    if (local) {
      unsigned rank[kFillPer][kFillMaxSpan];  // two 16-bit ranks per word: [u][jy] holds jx = 0, 1
#pragma unroll
      for (int u = 0; u < kFillPer; ++u) {
#pragma unroll
        for (int jy = 0; jy < kFillMaxSpan; ++jy) {
          rank[u][jy] = 0;
#pragma unroll
          for (int jx = 0; jx < kFillMaxSpan; ++jx) {
            const int t = (jx < (span[u] & 0xff) && jy < (span[u] >> 16) && !((span[u] >> (8 + jy * 2 + jx)) & 1))
                              ? t0[u] + jy * ntx + jx : -1;
            rank[u][jy] |= ((unsigned)wave_tile_reserve_lds(s_tab, t, s_touch, &s_ntouch) & 0xffffu) << (16 * jx);  // < kChunk <= 65536
          }
        }
      }
      static_assert(kChunk <= 65536, "ranks are packed in 16 bits");
      __syncthreads();
      n_touched = s_ntouch;
      // one global atomic per tile the chunk touched; the requests of a thread go out back to back
      if (n_touched <= kFillTouchCap) {
        for (int kb = threadIdx.x; kb < n_touched; kb += kFillThreads * 4) {
          int t[4], v[4], r[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int idx = kb + k * kFillThreads;
            t[k] = idx < n_touched ? (int)s_touch[idx] : -1;
            v[k] = t[k] >= 0 ? s_tab[t[k]] : 0;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            r[k] = 0;
            if (t[k] >= 0) {
              if (seg > 0) {
                const int at = atomicAdd(&cursor[t[k]], v[k]);
                if (at + v[k] > seg) atomicOr(overflow, 1);
                r[k] = t[k] * seg + at;
              } else {
                r[k] = offsets[t[k]] + atomicAdd(&cursor[t[k]], v[k]);
              }
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (t[k] >= 0) s_tab[t[k]] = r[k];  // list position of this chunk's first entry of the tile
        }
      } else
      for (int tb = threadIdx.x; tb < ntiles; tb += kFillThreads * 4) {
        int v[4], r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int t = tb + k * kFillThreads;
          v[k] = t < ntiles ? s_tab[t] : 0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int t = tb + k * kFillThreads;
          if (seg > 0) {
            const int at = v[k] ? atomicAdd(&cursor[t], v[k]) : 0;
            if (at + v[k] > seg) atomicOr(overflow, 1);
            r[k] = t * seg + at;
          } else {
            r[k] = v[k] ? offsets[t] + atomicAdd(&cursor[t], v[k]) : 0;
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (v[k]) s_tab[tb + k * kFillThreads] = r[k];  // list position of this chunk's first entry of the tile
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < kFillPer; ++u) {
        const int64_t i = c0 + (int64_t)u * kFillThreads + threadIdx.x;
        const float4 ent = make_float4(ex[u], ey[u], __int_as_float((int)i), ez[u]);
#pragma unroll
        for (int jy = 0; jy < kFillMaxSpan; ++jy)
#pragma unroll
          for (int jx = 0; jx < kFillMaxSpan; ++jx)
            if (jx < (span[u] & 0xff) && jy < (span[u] >> 16) && !((span[u] >> (8 + jy * 2 + jx)) & 1)) {
              const int t = t0[u] + jy * ntx + jx;
              const int64_t pos = (int64_t)s_tab[t] + (int)((rank[u][jy] >> (16 * jx)) & 0xffffu);
              if (pos < (seg > 0 ? (int64_t)(t + 1) * seg : list_capacity)) lists[pos] = ent;
            }
      }
      __syncthreads();  // the table is zeroed again for the next chunk
    } else {
#pragma unroll
      for (int u = 0; u < kFillPer; ++u) {
        const int64_t i = c0 + (int64_t)u * kFillThreads + threadIdx.x;
        const float4 ent = make_float4(ex[u], ey[u], __int_as_float((int)i), ez[u]);
        const int nx = wave_max_i32_scalar(span[u] & 0xffff), ny = wave_max_i32_scalar(span[u] >> 16);
        for (int jy = 0; jy < ny; ++jy)
          for (int jx = 0; jx < nx; ++jx) {
            int t = (jx < (span[u] & 0xffff) && jy < (span[u] >> 16)) ? t0[u] + jy * ntx + jx : -1;
            if (tile_bound != nullptr && t >= 0 && ez[u] > tile_bound[t]) t = -1;
            const int slot = wave_tile_reserve(cursor, t);
            if (t >= 0) {
              if (seg > 0) {
                if (slot < seg)
                  lists[(int64_t)t * seg + slot] = ent;
                else
                  atomicOr(overflow, 1);
              } else {
                const int64_t pos = (int64_t)offsets[t] + slot;
                if (pos < list_capacity) lists[pos] = ent;
              }
            }
          }
      }
    }
  }
}

// The K nearest of a pixel under pytorch3d's total order (z, id): one 64-bit key
// (z bits << 32 | id) per entry -- z >= 0 here, so the IEEE bit pattern is monotone and a single
// unsigned compare replaces the (z <, z ==, id <) triple.
template <int K>
struct TopK {
  unsigned long long key[K];
  static constexpr unsigned long long kEmpty = ~0ull;
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int k = 0; k < K; ++k) key[k] = kEmpty;
  }
  __device__ __forceinline__ bool has(int k) const { return key[k] != kEmpty; }
  __device__ __forceinline__ int id(int k) const { return (int)(unsigned)(key[k] & 0xffffffffull); }
  __device__ __forceinline__ float z(int k) const { return __uint_as_float((unsigned)(key[k] >> 32)); }
  // keep the K smallest keys, sorted; the caller has checked kk < key[K-1].  The list is
  // sorted, so "kk < key[k]" is monotone in k and every slot is a two-level select: take the
  // left neighbour if kk goes in front of it, else kk if it goes in front of this one.
  // (The squared distance is not carried along: the epilogue recomputes it from the id with the
  // same two subtractions, two products and one sum.)
  __device__ __forceinline__ void insert_below_last(unsigned long long kk) {
    bool c[K];
#pragma unroll
    for (int k = 0; k < K - 1; ++k) c[k] = kk < key[k];
    c[K - 1] = true;
#pragma unroll
    for (int k = K - 1; k > 0; --k) key[k] = c[k - 1] ? key[k - 1] : (c[k] ? kk : key[k]);
    key[0] = c[0] ? kk : key[0];
  }
};

constexpr int kSortCap = 2048;       // list entries the sorted path holds in LDS (<= 40 KB with the rest: 4 workgroups per CU, see the static_assert)
// Round 4: a second instantiation for the tiles of DENSE clouds (1080p x 48 frames: 1800 entries per tile on average, 40 % of
// the tiles beyond 2048 -- the general path took them at 2.5 x the time per tile and the tile pass grew from 0.2 to 1.1 ms):
// 4096 entries, 69 KB of LDS, two workgroups per CU.  Launched only from 2.2 rows per pixel (a host-side bound on the rows:
// the headline's clouds never pay for it); the first launch then leaves the longer lists to it.
constexpr int kSortCapLong = 4096;
constexpr int kSortBuckets = 1024;   // monotone z-buckets of the distribution pass
constexpr int kSortMaxBucket = 64;   // more entries than this in one bucket (equal depths): general path
// ranks are carried as floats (exact below 2^24; +inf = empty slot) so that v_med3_f32 applies

// insert rank r into the ascending list key[0..K) (the largest drops out): slot k takes the median of
// (left neighbour, itself, r), evaluated from the top down so that every slot sees old values
template <int K>
__device__ __forceinline__ void rank_insert(float (&key)[K], float r) {
#pragma unroll
  for (int k = K - 1; k > 0; --k) key[k] = __builtin_amdgcn_fmed3f(key[k - 1], key[k], r);
  // min(key[0], r) as ONE instruction: fminf -- and the median with -inf, which LLVM folds into it -- first
  // canonicalises both operands (v_max_f32 x, x); neither is ever NaN here (ranks or +inf)
  asm("v_min_f32 %0, %1, %2" : "=v"(key[0]) : "v"(key[0]), "v"(r));
}

__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// kCap: capacity of the sorted path (kSortCap / kSortCapLong).  pass: 0 = every tile (lists beyond kCap take the general
// path), 1 = the first of two launches (tiles with longer lists are left to the second), 2 = the second (only those tiles).
// kFrag: the fragments' depths are wanted (zbuf): the sorted path then keeps them in LDS by rank as well.  Without them (the
// renderer's path) a workgroup needs 28 KB instead of 37 -- the keys of the sort and the sorted coordinates share one
// array, ids are 4 bytes -- and FIVE workgroups fit a CU instead of four.
template <int K, int kCap, bool kFrag>
__global__ void __launch_bounds__(256, kCap <= kSortCap ? (kFrag ? 4 : 5) : 2)
raster_tile_kernel(const float4 *__restrict__ lists, const int32_t *__restrict__ offsets, int64_t list_capacity,
                   const float4 *__restrict__ seg_lists, const int32_t *__restrict__ seg_count, int seg,
                   const int32_t *__restrict__ seg_overflow, int pass,
                   const float *__restrict__ feat, int64_t feat_stride, float radius, int H, int W,
                   int ntx, int nty, int tiles_per_xcd, int64_t *__restrict__ idx_out,
                   float *__restrict__ zbuf_out, float *__restrict__ dist_out,
                   float *__restrict__ rgb_out, int rgb_planar, float *__restrict__ mask_out, int32_t *__restrict__ stats) {
  constexpr int kSortPerThread = kCap / 256;
  // one array, two lives: during the sort the keys (z bits, id) in bucket order; from step 6 on the NDC (x, y) in rank order
  // (every key has been read by then: barrier).  General path: staging of 256 entries in the first 4 KB.
  __shared__ uint2 s_buf[kCap];
  uint2 *s_kz = s_buf;
  float2 *s_xy = reinterpret_cast<float2 *>(s_buf);
  __shared__ unsigned s_id[kCap];                // rank order: point id
  __shared__ unsigned s_zr[kFrag ? kCap : 1];    // rank order: z bits (fragments only)
  __shared__ float4 s_wave[4][68];          // per-wave strip of culled points (+ padding)
  // bucket counts, then bucket starts: dead before the walk begins, so they share the strips' storage
  unsigned *s_cnt = reinterpret_cast<unsigned *>(&s_wave[0][0]);
  static_assert(kSortBuckets * 4 <= 4 * 68 * 16, "the counters alias the strips");
  static_assert(kCap * (kFrag ? 16 : 12) + 4 * 68 * 16 + 128 <= (kCap <= kSortCap ? (kFrag ? 40960 : 32768) : 81920),
                "four / five (two) workgroups per CU");
  static_assert(kCap % 256 == 0 && kCap <= 4096, "every thread owns kCap / 256 entries; positions are packed in 12 bits");
  __shared__ unsigned s_red[12];
  __shared__ int s_flag;
  // XCD-aware mapping: blocks b, b+8, ... share an XCD -> give them a contiguous tile band
  const int ntiles = ntx * nty;
  // Tiles in launch order: neighbouring tiles land on different XCDs (workgroup b runs on XCD b mod 8).  Until round 5 every
  // XCD drew one contiguous band of the image (tiles_per_xcd: b -> (b & 7) * band + (b >> 3)) for the locality of the
  // epilogue's feature gathers: 177 -> 167 us with the plain order, +1.9 % throughput (A/B, three pairs).
  (void)tiles_per_xcd;
  // (scattered -- b * 2731 mod tiles -- or reversed instead of the plain order: 171 / 168 us against 170, no tail to trim;
  // tile ROWS round-robin over the XCDs -- every XCD an eighth of the rows, horizontal neighbours on one L2 --: 176 us, as
  // slow as the bands: it is neighbours sharing an XCD that costs, not the bands' imbalance; the plain order pays with
  // 87 MB more counter traffic per view, the neighbours' shared rows now fetched into several L2s)
  const int tile = blockIdx.x;
  if (tile >= ntiles) return;
  const int ty = tile / ntx, tx = tile - ty * ntx;
  // a wavefront owns one 8x8 quadrant of the tile so that it can cull the list against its own
  // (radius-expanded) bounds before the per-pixel tests
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = (wave & 1) * 8 + (lane & 7), ly = (wave >> 1) * 8 + (lane >> 3);
  const int xi = tx * kTile + lx, yi = ty * kTile + ly;
  const bool inside = xi < W && yi < H;
  const float range_x = W > H ? 2.0f * (float)W / (float)H : 2.0f;
  const float range_y = H > W ? 2.0f * (float)H / (float)W : 2.0f;
  // pixels of a partial tile outside the image: NaN centre, every `d2 < r2` test fails
  const float xf = inside ? pix_to_ndc(W - 1 - xi, W, range_x) : __builtin_nanf("");
  const float yf = pix_to_ndc(H - 1 - yi, H, range_y);
  const float r2 = radius * radius;
  // quadrant bounds in NDC (pixel index is reversed: larger xi = smaller x), widened by the
  // radius and a rounding margin
  const int qx0 = tx * kTile + (wave & 1) * 8, qy0 = ty * kTile + (wave >> 1) * 8;
  const float margin = radius * 1.001f + 1e-6f;
  const float bx_hi = pix_to_ndc(W - 1 - qx0, W, range_x) + margin;
  const float bx_lo = pix_to_ndc(W - 1 - (qx0 + 7), W, range_x) - margin;
  const float by_hi = pix_to_ndc(H - 1 - qy0, H, range_y) + margin;
  const float by_lo = pix_to_ndc(H - 1 - (qy0 + 7), H, range_y) - margin;
  // the tile's list: its segment of the direct binning pass, or -- no direct pass, or one of its segments overflowed -- the
  // exact passes' range (uniform over the launch: every tile reads the same flag)
  const bool direct = seg > 0 && *seg_overflow == 0;
  int64_t beg, end;
  if (direct) {
    beg = (int64_t)tile * seg;
    end = beg + seg_count[tile];
  } else {
    beg = offsets[tile];
    end = offsets[tile + 1];
    if (end > list_capacity) end = list_capacity;
  }
  const int64_t n64 = end > beg ? end - beg : 0;
  const float4 *__restrict__ L = (direct ? seg_lists : lists) + beg;

  // per-pixel results, common to both paths
  bool has[K];
  int rid[K];
  float rz[K], rd2[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    has[k] = false;
    rid[k] = -1;
    rz[k] = -1.0f;
    rd2[k] = -1.0f;
  }

  // (two launches: every tile is drawn by exactly one of them)
  if ((pass == 1 && n64 > kSortCap) || (pass == 2 && n64 <= kSortCap)) return;
  bool sorted_path = n64 > 0 && n64 <= kCap;  // (uniform over the workgroup)
  if (sorted_path) {
    const int n = (int)n64;
    // ---- 1. the list, once, into registers (all loads in flight together); its depth range; clear the counters
    float4 ent[kSortPerThread];
#pragma unroll
    for (int k = 0; k < kSortPerThread; ++k) {
      const int e = tid + k * 256;
      ent[k] = L[e < n ? e : n - 1];
    }
    unsigned zmn = 0xffffffffu, zmx = 0u;
#pragma unroll
    for (int k = 0; k < kSortPerThread; ++k) {
      const unsigned zb = __float_as_uint(ent[k].w);  // (clamped duplicates of the last entry change neither bound)
      zmn = zb < zmn ? zb : zmn;
      zmx = zb > zmx ? zb : zmx;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned a = __shfl_xor(zmn, off, 64), b = __shfl_xor(zmx, off, 64);
      zmn = a < zmn ? a : zmn;
      zmx = b > zmx ? b : zmx;
    }
    if (lane == 0) {
      s_red[wave] = zmn;
      s_red[4 + wave] = zmx;
    }
#pragma unroll
    for (int j = 0; j < kSortBuckets / 256; ++j) s_cnt[tid + j * 256] = 0u;
    if (tid == 0) s_flag = 0;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      zmn = s_red[w] < zmn ? s_red[w] : zmn;
      zmx = s_red[4 + w] > zmx ? s_red[4 + w] : zmx;
    }
    // bucket = floor((z bits - min) * scale): conversions, product and truncation are monotone, so
    // a smaller depth never lands in a later bucket (z >= 0: the bit patterns order like the values)
    const float scale = (float)kSortBuckets / ((float)(zmx - zmn) + 1.0f);
    // ---- 2. bucket of every entry + its arrival number inside the bucket
    unsigned pos[kSortPerThread];
#pragma unroll
    for (int k = 0; k < kSortPerThread; ++k) {
      pos[k] = 0u;
      if (k * 256 < n) {
        const int e = tid + k * 256;
        if (e < n) {
          const unsigned zb = __float_as_uint(ent[k].w);
          unsigned b = (unsigned)((float)(zb - zmn) * scale);
          b = b < (unsigned)kSortBuckets - 1u ? b : (unsigned)kSortBuckets - 1u;
          pos[k] = (b << 12) | atomicAdd(&s_cnt[b], 1u);
        }
      }
    }
    __syncthreads();
    // ---- 3. exclusive scan of the kSortBuckets (1024) counters, four per thread
    {
      unsigned c[kSortBuckets / 256], tot = 0u, mx = 0u;
#pragma unroll
      for (int j = 0; j < kSortBuckets / 256; ++j) {
        c[j] = s_cnt[tid * (kSortBuckets / 256) + j];
        mx = c[j] > mx ? c[j] : mx;
        tot += c[j];
      }
      if (mx > (unsigned)kSortMaxBucket) s_flag = 1;
      unsigned incl = tot;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned y = __shfl_up(incl, off, 64);
        if (lane >= off) incl += y;
      }
      if (lane == 63) s_red[8 + wave] = incl;
      __syncthreads();
      unsigned base = incl - tot;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (w < wave) base += s_red[8 + w];
#pragma unroll
      for (int j = 0; j < kSortBuckets / 256; ++j) {
        s_cnt[tid * (kSortBuckets / 256) + j] = base;
        base += c[j];
      }
    }
    __syncthreads();
    sorted_path = s_flag == 0;
    if (!sorted_path && tid == 0) atomicAdd(&stats[0], 1);  // (> kSortMaxBucket equal depths in one bucket: rare)
    if (sorted_path) {
      // ---- 4. keys into bucket order
#pragma unroll
      for (int k = 0; k < kSortPerThread; ++k) {
        if (k * 256 < n) {
          const int e = tid + k * 256;
          if (e < n) {
            const float4 q = ent[k];
            const unsigned b = pos[k] >> 12;
            const unsigned p = s_cnt[b] + (pos[k] & 0xfffu);
            s_kz[p] = make_uint2(__float_as_uint(q.w), __float_as_uint(q.z));
            pos[k] = (b << 12) | p;
          }
        }
      }
      __syncthreads();
      // ---- 5. exact rank: bucket start + the number of smaller (z, id) keys in the same bucket
#pragma unroll
      for (int k = 0; k < kSortPerThread; ++k) {
        if (k * 256 < n) {
          const int e = tid + k * 256;
          if (e < n) {
            const unsigned b = pos[k] >> 12, p = pos[k] & 0xfffu;
            const unsigned bs = s_cnt[b], be = b + 1u < (unsigned)kSortBuckets ? s_cnt[b + 1u] : (unsigned)n;
            const uint2 me = s_kz[p];
            unsigned r = bs;
            // buckets hold 0-3 entries: the first four are read together (independent LDS reads), a
            // longer bucket finishes in a loop
            uint2 o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = s_kz[bs + j < (unsigned)n ? bs + j : (unsigned)n - 1u];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              r += (bs + j < be) & ((o[j].x < me.x) | ((o[j].x == me.x) & (o[j].y < me.y))) ? 1u : 0u;
            for (unsigned j = bs + 4u; j < be; ++j) {
              const uint2 oj = s_kz[j];
              r += (oj.x < me.x) | ((oj.x == me.x) & (oj.y < me.y)) ? 1u : 0u;
            }
            pos[k] = r;
          }
        }
      }
      __syncthreads();
      // ---- 6. entries into rank order
#pragma unroll
      for (int k = 0; k < kSortPerThread; ++k) {
        if (k * 256 < n) {
          const int e = tid + k * 256;
          if (e < n) {
            const float4 q = ent[k];
            s_xy[pos[k]] = make_float2(q.x, q.y);
            s_id[pos[k]] = __float_as_uint(q.z);
            if (kFrag) s_zr[pos[k]] = __float_as_uint(q.w);
          }
        }
      }
      __syncthreads();
      // ---- 7. front-to-back walk, one quadrant per wavefront.  Per 64 sorted entries: box cull with a
      // ballot, survivors (x, y, rank) compacted into the wave's LDS strip; per survivor one broadcast read,
      // the disc test and a branch-free insertion of its rank (+inf where the disc misses the pixel).
      // (Tried: ranks rebuilt from the ballot mask on the scalar unit with 8-byte strip entries and an
      // exec-masked insertion -- fewer vector instructions, but ~9 scalar ones per test, and the CU's one
      // scalar unit serves four SIMDs: 340 us against 318 us.)
      float key[K];
#pragma unroll
      for (int k = 0; k < K; ++k) key[k] = __builtin_inff();
      float4 *strip = s_wave[wave];
      for (int base = 0; base < n; base += 64) {
        // every pixel of the quadrant holds K points: whatever follows lies behind all of them
        const unsigned long long open = __ballot(inside && key[K - 1] == __builtin_inff());
        if (open == 0ull) break;
        // Only the pixels still OPEN can take a point (the walk is front to back: a full pixel's list is final), and
        // they huddle -- along a depth edge, in a corner the nearer surface does not cover -- so the cull below uses
        // the rectangle of the open pixels' centres, not the quadrant's: -25 % point tests on the benchmark scene.
        // Rows / columns of the open set from the ballot on the scalar unit (lane = 8 * row + column), their NDC
        // coordinates read from the lanes that own them (x and y decrease with the pixel index).
        const int r_lo = __builtin_ctzll(open) >> 3, r_hi = (63 - __builtin_clzll(open)) >> 3;
        unsigned cols = (unsigned)(open | (open >> 32));
        cols |= cols >> 16;
        cols = (cols | (cols >> 8)) & 0xffu;
        const int c_lo = __builtin_ctz(cols), c_hi = 31 - __builtin_clz(cols);
        // (x from an OPEN lane of the column: lanes outside the image carry NaN there; y is valid on every lane)
        const float ox_hi = readlane_f(xf, __builtin_ctzll(open & (0x0101010101010101ull << c_lo)));
        const float ox_lo = readlane_f(xf, __builtin_ctzll(open & (0x0101010101010101ull << c_hi)));
        const float oy_hi = readlane_f(yf, r_lo * 8), oy_lo = readlane_f(yf, r_hi * 8);
        const int e = base + lane;
        const float2 c = s_xy[e < n ? e : 0];
        // the disc must reach the rectangle of those pixel centres (distance of its centre to the rectangle, with the
        // same rounding margin as the box): drops the corners of the expanded box
        const float ex = fmaxf(fmaxf(ox_lo - c.x, c.x - ox_hi), 0.0f), ey = fmaxf(fmaxf(oy_lo - c.y, c.y - oy_hi), 0.0f);
        const bool in = e < n && ex * ex + ey * ey <= margin * margin;
        const unsigned long long mask = __ballot(in);
        if (!mask) continue;
        const int cnt = (int)__popcll(mask);
        if (in) strip[__popcll(mask & ((1ull << lane) - 1ull))] = make_float4(c.x, c.y, (float)e, 0.f);
        if (lane < 3) strip[cnt + lane] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), 0.f);
        // same wave writes and reads: LDS ops of one wave complete in order; keep the compiler
        // from moving the reads above the writes
        __builtin_amdgcn_wave_barrier();
        float4 p[4], pn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = strip[u];
        for (int j = 0; j < cnt; j += 4) {
          // the next four entries are requested before these four are evaluated
#pragma unroll
          for (int u = 0; u < 4; ++u) pn[u] = strip[(j + 4 + u) & 63];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float dx = p[u].x - xf, dy = p[u].y - yf;
            const float d2 = dx * dx + dy * dy;
            rank_insert<K>(key, d2 < r2 ? p[u].z : __builtin_inff());
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) p[u] = pn[u];
        }
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (key[k] != __builtin_inff()) {
          const int r = (int)key[k];
          const float2 c = s_xy[r];
          const float dx = c.x - xf, dy = c.y - yf;
          has[k] = true;
          rid[k] = (int)s_id[r];
          rz[k] = kFrag ? __uint_as_float(s_zr[kFrag ? r : 0]) : 0.0f;  // (only written out with kFrag)
          rd2[k] = dx * dx + dy * dy;  // the loop's own arithmetic on the same operands
        }
      }
    }
  }
  if (!sorted_path && n64 > 0) {
    // ---- general path: any list length, any order; 64-bit (z, id) keys kept sorted per pixel
    __syncthreads();
    float4 *s_pt = reinterpret_cast<float4 *>(s_xy);  // 256 staged entries
    TopK<K> q;
    q.init();
    for (int64_t base = 0; base < n64; base += 256) {
      const int64_t e = base + tid;
      __syncthreads();
      if (e < n64) s_pt[tid] = L[e];  // (x, y, id, z): (id, z) in this order is the key's (low, high) register pair
      __syncthreads();
      const int m = (int)((n64 - base) < 256 ? (n64 - base) : 256);
      for (int sub = 0; sub < m; sub += 64) {
        // hierarchical z: once every pixel of the quadrant holds K entries, a point strictly behind the
        // farthest of their K-th depths cannot enter any list (an empty slot counts as +inf, pixels
        // outside the image as 0)
        const int zcull = wave_max_i32_scalar(inside ? (q.has(K - 1) ? (int)(q.key[K - 1] >> 32) : 0x7fffffff) : 0);
        const float4 c = s_pt[(sub + lane) & 255];
        const bool hit = (sub + lane) < m && c.x >= bx_lo && c.x <= bx_hi && c.y >= by_lo && c.y <= by_hi &&
                         __float_as_int(c.w) <= zcull;
        const unsigned long long mask = __ballot(hit);
        if (!mask) continue;
        const int cnt = (int)__popcll(mask);
        float4 *strip = s_wave[wave];
        if (hit) strip[__popcll(mask & ((1ull << lane) - 1ull))] = c;
        if (lane < 3) strip[cnt + lane] = make_float4(__builtin_inff(), __builtin_inff(), 0.f, 0.f);
        __builtin_amdgcn_wave_barrier();
        for (int j = 0; j < cnt; j += 4) {
          float4 p[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) p[u] = strip[j + u];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float dx = p[u].x - xf, dy = p[u].y - yf;
            const float d2 = dx * dx + dy * dy;
            const unsigned long long kk =
                ((unsigned long long)__float_as_uint(p[u].w) << 32) | __float_as_uint(p[u].z);
            // one branch for both tests (bitwise &: no short-circuit)
            if ((d2 < r2) & (kk < q.key[K - 1])) q.insert_below_last(kk);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    // squared distances of the kept points: one pass over the list (ids are unique)
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (q.has(k)) {
        has[k] = true;
        rid[k] = q.id(k);
        rz[k] = q.z(k);
      }
    }
    for (int64_t base = 0; base < n64; base += 256) {
      const int64_t e = base + tid;
      __syncthreads();
      if (e < n64) s_pt[tid] = L[e];
      __syncthreads();
      const int m = (int)((n64 - base) < 256 ? (n64 - base) : 256);
      for (int j = 0; j < m; ++j) {
        const float4 c = s_pt[j];
        const int id = __float_as_int(c.z);
#pragma unroll
        for (int k = 0; k < K; ++k)
          if (has[k] && rid[k] == id) {
            const float dx = c.x - xf, dy = c.y - yf;
            rd2[k] = dx * dx + dy * dy;
          }
      }
    }
  }
  if (!inside) return;
  const size_t pix = (size_t)yi * W + xi;
  // NormWeightedCompositor: w = 1 - d2/r2, t = max(sum w, 1e-4), out = sum w*f/t
  float t = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k)
    if (has[k]) t = t + (1.0f - rd2[k] / r2);
  t = t > 1e-4f ? t : 1e-4f;
  float acc[3] = {0.f, 0.f, 0.f}, ones = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (idx_out) idx_out[pix * K + k] = has[k] ? (int64_t)rid[k] : (int64_t)-1;
    if (zbuf_out) zbuf_out[pix * K + k] = has[k] ? rz[k] : -1.0f;
    if (dist_out) dist_out[pix * K + k] = rd2[k];
    if (has[k]) {
      float w = 1.0f - rd2[k] / r2;
      ones = ones + w * 1.0f / t;
      if (rgb_out) {
        const float *f = feat + (int64_t)rid[k] * feat_stride;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = acc[c] + w * f[c] / t;
      }
    }
  }
  if (rgb_out) {
    if (rgb_planar) {
      const size_t P = (size_t)H * W;
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb_out[c * P + pix] = acc[c];
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb_out[pix * 3 + c] = acc[c];
    }
  }
  if (mask_out) mask_out[pix] = ones > 0.0f ? 1.0f : 0.0f;
}

__global__ void raster_status_kernel(const int64_t *__restrict__ n_dev, int32_t *__restrict__ status) {
  if (threadIdx.x == 0) *status = (n_dev && *n_dev > 0) ? 1 : ((n_dev && *n_dev < 0) ? 2 : 0);
}

static int64_t max_tiles_per_point(float radius, int H, int W) {
  float range_x = W > H ? 2.0f * (float)W / (float)H : 2.0f;
  float range_y = H > W ? 2.0f * (float)H / (float)W : 2.0f;
  double wx = 2.0 * ((double)radius * W / range_x + 1.5) + 2.0;
  double wy = 2.0 * ((double)radius * H / range_y + 1.5) + 2.0;
  int64_t nx = (int64_t)(wx / kTile) + 2, ny = (int64_t)(wy / kTile) + 2;
  int64_t ntx = cdiv(W, kTile), nty = cdiv(H, kTile);
  if (nx > ntx) nx = ntx;
  if (ny > nty) ny = nty;
  return nx * ny;
}

constexpr double kSegLayoutDensity = 2.2;  // rows per pixel below which a workspace carries the direct pass's segments
constexpr int kSegEntries = 4096;  // most entries per tile segment of the direct binning pass (the longest list of the 1080p x 24 benchmark: ~1900)
struct RasterWs {
  int32_t *tile_count, *cursor, *offsets;
  int32_t *seg_cursor;  // [ntiles] entries per tile as the direct binning pass counted them (its per-tile cursor)
  float4 *seg_lists;    // [ntiles][seg_entries] (null: the workspace was sized without them)
  int seg_entries;      // twice the average list the row bound allows, at least 256 and at most kSegEntries per tile
  int32_t *stats;  // [64] zeroed per call with the counters: [0] tiles the sorted path handed to the general path for equal depths,
                   // [1] a segment of the direct binning pass overflowed (the exact passes ran, the tile pass read their lists)
  unsigned *zmin;      // [H*W] per-pixel minimum depth of the point centres (bit patterns), filled per call
  float *tile_bound;   // [ntiles] depth behind which nothing can enter the tile's lists
  float4 *lists;  // 16-byte entries (x_ndc, y_ndc, id, z)
  int64_t list_capacity;
  int64_t total_bytes;
};

static RasterWs raster_ws_layout(void *base, int64_t n, int H, int W, float radius) {
  RasterWs w;
  int64_t ntiles = cdiv(W, kTile) * cdiv(H, kTile);
  char *p = reinterpret_cast<char *>(base);
  int64_t off = 0;
  w.tile_count = reinterpret_cast<int32_t *>(p + off);
  off += align_up(ntiles * 4, 256);
  w.cursor = reinterpret_cast<int32_t *>(p + off);
  off += align_up(ntiles * 4, 256);
  w.seg_cursor = reinterpret_cast<int32_t *>(p + off);
  off += align_up(ntiles * 4, 256);
  w.stats = reinterpret_cast<int32_t *>(p + off);
  off += 256;
  w.offsets = reinterpret_cast<int32_t *>(p + off);
  off += align_up((ntiles + 1) * 4, 256);
  w.tile_bound = reinterpret_cast<float *>(p + off);
  off += align_up(ntiles * 4, 256);
  w.zmin = reinterpret_cast<unsigned *>(p + off);
  off += align_up((int64_t)H * W * 4, 256);
  w.list_capacity = (n > 0 ? n : 1) * max_tiles_per_point(radius, H, W);
  w.lists = reinterpret_cast<float4 *>(p + off);
  off += align_up(w.list_capacity * 16, 256);
  // the direct pass's segments: only where a segment is worth more than the average list can need (images of at least a
  // few tiles; tiny test images keep the exact passes alone) and the block stays below 2 GB
  // ... and clouds the direct pass can take at all: below kSegLayoutDensity rows per pixel (the default of the runtime gate,
  // option raster_bound_density; a workspace sized for a denser cloud -- 1080p x 48 frames: 0.53 GB of segments per lane -- goes
  // straight to the exact passes and carries no segment block.  A caller that raises the runtime gate beyond this keeps
  // correct results: without segments the exact passes run.)
  w.seg_lists = nullptr;
  w.seg_entries = 0;
  if (ntiles >= 16 && n > 0 && (double)n < kSegLayoutDensity * (double)H * (double)W) {
    int64_t se = align_up(2 * cdiv(w.list_capacity, ntiles), 256);
    se = se < 256 ? 256 : (se > kSegEntries ? kSegEntries : se);
    if (ntiles * se * 16 <= (2ll << 30)) {
      w.seg_entries = (int)se;
      w.seg_lists = reinterpret_cast<float4 *>(p + off);
      off += align_up(ntiles * se * 16, 256);
    }
  }
  w.total_bytes = off;
  return w;
}

// what the last rasterisation on this workspace did: out[0] list entries, out[1] longest tile list, out[2] tiles whose list was
// too long for the sorted path (general path), out[3] tiles the sorted path gave up on for equal depths.  One workgroup.
__global__ void __launch_bounds__(1024) raster_counters_kernel(const int32_t *__restrict__ offsets, int ntiles,
                                                                const int32_t *__restrict__ stats, int64_t *__restrict__ out,
                                                                const int32_t *__restrict__ seg_cursor, int seg) {
  __shared__ int s_max[16], s_long[16];
  __shared__ long long s_tot[16];
  // (the direct binning pass's counts when its lists were the ones drawn)
  const bool direct = seg > 0 && seg_cursor != nullptr && stats[2] != 0 && stats[1] == 0;
  int mx = 0, nlong = 0;
  long long tot = 0;
  for (int t = threadIdx.x; t < ntiles; t += 1024) {
    const int n = direct ? seg_cursor[t] : offsets[t + 1] - offsets[t];
    tot += n;
    mx = n > mx ? n : mx;
    nlong += n > kSortCap ? 1 : 0;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const int a = __shfl_xor(mx, off, 64);
    mx = a > mx ? a : mx;
    nlong += __shfl_xor(nlong, off, 64);
    tot += __shfl_xor(tot, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_max[threadIdx.x >> 6] = mx;
    s_long[threadIdx.x >> 6] = nlong;
    s_tot[threadIdx.x >> 6] = tot;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int m = 0, l = 0;
    long long tt = 0;
    for (int w = 0; w < 16; ++w) {
      m = s_max[w] > m ? s_max[w] : m;
      l += s_long[w];
      tt += s_tot[w];
    }
    out[0] = tt;
    out[1] = m;
    out[2] = l;
    out[3] = stats[0];
  }
}

void raster_counters(const void *workspace, int64_t n_rows, int H, int W, float radius, int64_t *out_dev, hipStream_t st) {
  const RasterWs ws = raster_ws_layout(const_cast<void *>(workspace), n_rows, H, W, radius);
  const int ntiles = (int)(cdiv(W, kTile) * cdiv(H, kTile));
  PGDVS_LAUNCH("raster_counters", raster_counters_kernel, dim3(1), dim3(1024), 0, st, (const int32_t *)ws.offsets, ntiles,
               (const int32_t *)ws.stats, out_dev, (const int32_t *)ws.seg_cursor, ws.seg_entries);
}

}  // namespace pgdvs

using namespace pgdvs;

namespace pgdvs {
// fused.h: the block of counters a rasterisation starts from (cleared per call), and the bounded rasterisation with that block
// already cleared by the caller
void raster_counter_block(void *workspace, int64_t n_rows, int H, int W, float radius, void **block, int64_t *bytes) {
  const RasterWs ws = raster_ws_layout(workspace, n_rows, H, W, radius);
  *block = ws.tile_count;
  *bytes = (int64_t)((char *)ws.offsets - (char *)ws.tile_count);
}
}  // namespace pgdvs
static int points_raster_impl(const float *pts, int64_t pts_stride, const float *feat, int64_t feat_stride, int64_t n_points,
                              const int64_t *n_points_dev, int32_t *status_dev, const float *cam_tgt, float radius, int K,
                              int H, int W, int64_t *idx, float *zbuf, float *dist2, float *rgb, int rgb_planar, float *mask,
                              void *workspace, int64_t workspace_bytes, pgdvs_stream_t stream, bool counters_cleared);
namespace pgdvs {
int points_raster_bounded_cleared(const float *pts, int64_t pts_stride, const float *feat, int64_t feat_stride, int64_t n_points,
                                  const int64_t *n_points_dev, int64_t row_bound, int32_t *status_dev, const float *cam_tgt,
                                  float radius, int K, int H, int W, float *rgb, int rgb_planar, float *mask, void *workspace,
                                  int64_t workspace_bytes, pgdvs_stream_t stream, bool counters_cleared) {
  PGDVS_REQUIRE(row_bound >= 0 && status_dev && n_points >= 0, "points_raster_bounded_cleared: bad arguments");
  if (n_points_dev == nullptr && n_points > row_bound) {
    set_error("pgdvs_points_raster_bounded: %lld rows given (host count) but the row bound is %lld", (long long)n_points,
              (long long)row_bound);
    return PGDVS_ERR_INVALID;
  }
  return points_raster_impl(pts, pts_stride, feat, feat_stride, n_points < row_bound ? n_points : row_bound, n_points_dev,
                            status_dev, cam_tgt, radius, K, H, W, nullptr, nullptr, nullptr, rgb, rgb_planar, mask, workspace,
                            workspace_bytes, stream, counters_cleared);
}
}  // namespace pgdvs

PGDVS_API int64_t pgdvs_points_raster_workspace_bytes(int64_t n_points, int H, int W, float radius) {
  if (n_points < 0 || H <= 0 || W <= 0) return -1;
  const RasterWs w = raster_ws_layout(nullptr, n_points, H, W, radius);
  if (w.list_capacity >= (1ll << 31)) {
    set_error("pgdvs_points_raster: %lld points x this radius exceeds the 2^31 tile-list entries supported", (long long)n_points);
    return PGDVS_ERR_UNSUPPORTED;
  }
  return w.total_bytes;
}

template <int K>
static void launch_tile(dim3 grid, hipStream_t st, const RasterWs &ws, const float *feat,
                        int64_t feat_stride, float radius, int H, int W, int ntx, int nty,
                        int tiles_per_xcd, int64_t *idx, float *zbuf, float *dist2, float *rgb,
                        int rgb_planar, float *mask, bool long_lists, int seg) {
  const float4 *sl = (const float4 *)ws.seg_lists;
  const int32_t *sc = (const int32_t *)ws.seg_cursor, *so = (const int32_t *)(ws.stats + 1);
  if (zbuf != nullptr) {
    PGDVS_LAUNCH("raster_tile", (raster_tile_kernel<K, kSortCap, true>), grid, dim3(256), 0, st, (const float4 *)ws.lists,
                 (const int32_t *)ws.offsets, ws.list_capacity, sl, sc, seg, so, long_lists ? 1 : 0, feat, feat_stride, radius, H, W, ntx, nty,
                 tiles_per_xcd, idx, zbuf, dist2, rgb, rgb_planar, mask, ws.stats);
  } else {
    PGDVS_LAUNCH("raster_tile", (raster_tile_kernel<K, kSortCap, false>), grid, dim3(256), 0, st, (const float4 *)ws.lists,
                 (const int32_t *)ws.offsets, ws.list_capacity, sl, sc, seg, so, long_lists ? 1 : 0, feat, feat_stride, radius, H, W, ntx, nty,
                 tiles_per_xcd, idx, zbuf, dist2, rgb, rgb_planar, mask, ws.stats);
  }
  if (long_lists) {  // (the long-list launch keeps the depths in LDS either way: two workgroups per CU with or without)
    PGDVS_LAUNCH("raster_tile_long", (raster_tile_kernel<K, kSortCapLong, true>), grid, dim3(256), 0, st, (const float4 *)ws.lists,
                 (const int32_t *)ws.offsets, ws.list_capacity, sl, sc, seg, so, 2, feat, feat_stride, radius, H, W, ntx, nty, tiles_per_xcd, idx, zbuf,
                 dist2, rgb, rgb_planar, mask, ws.stats);
  }
}

static int points_raster_impl(const float *pts, int64_t pts_stride, const float *feat, int64_t feat_stride, int64_t n_points,
                              const int64_t *n_points_dev, int32_t *status_dev, const float *cam_tgt, float radius, int K,
                              int H, int W, int64_t *idx, float *zbuf, float *dist2, float *rgb, int rgb_planar, float *mask,
                              void *workspace, int64_t workspace_bytes, pgdvs_stream_t stream, bool counters_cleared = false);

PGDVS_API int pgdvs_points_raster(const float *pts, int64_t pts_stride, const float *feat,
                                  int64_t feat_stride, int64_t n_points,
                                  const int64_t *n_points_dev, const float *cam_tgt, float radius,
                                  int K, int H, int W, int64_t *idx, float *zbuf, float *dist2,
                                  float *rgb, int rgb_planar, float *mask, void *workspace,
                                  int64_t workspace_bytes, pgdvs_stream_t stream) {
  return points_raster_impl(pts, pts_stride, feat, feat_stride, n_points, n_points_dev, nullptr, cam_tgt, radius, K, H, W, idx,
                            zbuf, dist2, rgb, rgb_planar, mask, workspace, workspace_bytes, stream);
}

// The same with the workspace sized for `row_bound` rows instead of the arrays' capacity (a cloud buffer is
// capacity-sized -- S*H*W rows -- while a few per cent of the rows exist: 3.2 GB of tile lists per view in flight at
// 1080p x 24 frames against 0.25 GB for the rows actually there).  The device count is clamped to the bound and
// status_dev reports whether that cut rows off.
PGDVS_API int pgdvs_points_raster_bounded(const float *pts, int64_t pts_stride, const float *feat, int64_t feat_stride,
                                          int64_t n_points, const int64_t *n_points_dev, int64_t row_bound,
                                          int32_t *status_dev, const float *cam_tgt, float radius, int K, int H, int W,
                                          int64_t *idx, float *zbuf, float *dist2, float *rgb, int rgb_planar, float *mask,
                                          void *workspace, int64_t workspace_bytes, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(row_bound >= 0 && status_dev, "pgdvs_points_raster_bounded: bad row_bound / null status_dev");
  PGDVS_REQUIRE(n_points >= 0, "pgdvs_points_raster_bounded: bad n_points");
  if (n_points_dev == nullptr && n_points > row_bound) {
    set_error("pgdvs_points_raster_bounded: %lld rows given (host count) but the row bound is %lld", (long long)n_points,
              (long long)row_bound);
    return PGDVS_ERR_INVALID;
  }
  return points_raster_impl(pts, pts_stride, feat, feat_stride, n_points < row_bound ? n_points : row_bound, n_points_dev,
                            status_dev, cam_tgt, radius, K, H, W, idx, zbuf, dist2, rgb, rgb_planar, mask, workspace,
                            workspace_bytes, stream);
}

static int points_raster_impl(const float *pts, int64_t pts_stride, const float *feat, int64_t feat_stride, int64_t n_points,
                              const int64_t *n_points_dev, int32_t *status_dev, const float *cam_tgt, float radius, int K,
                              int H, int W, int64_t *idx, float *zbuf, float *dist2, float *rgb, int rgb_planar, float *mask,
                              void *workspace, int64_t workspace_bytes, pgdvs_stream_t stream, bool counters_cleared) {
  PGDVS_REQUIRE(H > 0 && W > 0 && (int64_t)H * W < (1ll << 31), "pgdvs_points_raster: bad H/W");
  PGDVS_REQUIRE(n_points >= 0 && n_points < (1ll << 31), "pgdvs_points_raster: bad n_points");
  PGDVS_REQUIRE(cam_tgt && (n_points == 0 || pts) && pts_stride >= 3, "pgdvs_points_raster: bad points");
  PGDVS_REQUIRE(!rgb || n_points == 0 || (feat && feat_stride >= 3), "pgdvs_points_raster: rgb output needs features");
  PGDVS_REQUIRE(radius > 0.0f, "pgdvs_points_raster: radius must be > 0");
  if (K < 1 || K > kRasterMaxK) {
    set_error("pgdvs_points_raster: points_per_pixel must be in [1, %d]", kRasterMaxK);
    return PGDVS_ERR_UNSUPPORTED;
  }
  RasterWs ws = raster_ws_layout(workspace, n_points, H, W, radius);
  if (ws.list_capacity >= (1ll << 31)) {  // int32 list offsets
    set_error("pgdvs_points_raster: %lld points x %lld tiles per point exceeds the 2^31 tile-list entries supported",
              (long long)n_points, (long long)(ws.list_capacity / (n_points > 0 ? n_points : 1)));
    return PGDVS_ERR_UNSUPPORTED;
  }
  if (!workspace || workspace_bytes < ws.total_bytes) {
    set_error("pgdvs_points_raster: workspace too small (%lld < %lld)", (long long)workspace_bytes,
              (long long)ws.total_bytes);
    return PGDVS_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const int ntx = (int)cdiv(W, kTile), nty = (int)cdiv(H, kTile), ntiles = ntx * nty;
  if (!counters_cleared) {
    hipError_t e = hipMemsetAsync(ws.tile_count, 0, (size_t)((char *)ws.offsets - (char *)ws.tile_count), st);
    if (e != hipSuccess) {
      set_error("points_raster memset: %s", hipGetErrorString(e));
      return PGDVS_ERR_LAUNCH;
    }
  }
  const bool small_table = ntiles <= kBinSlots / 2;  // the binning kernels' table of tile counters: 32 KB or 64 KB
  // The depth bound (raster_zmin / raster_bound): a b x b block's points must cover the block's pixels -- radius in pixels
  // beyond (b - 1/2) sqrt(2) by a margin that dwarfs the rounding of the pixel <-> NDC mapping (1e-3 px at 4k) -- and a block
  // must be able to hold K points.  b = 4 from 5.05 px (the reference's radius 0.01 at 1080p: 5.4 px), b = 2 from 2.25 px.
  const float *tile_bound = nullptr;
  int64_t gate_rows = 0;
  {
    // option raster_bound_density: rows per pixel from which the bound is computed (default 2.2; 0 = always, a large value = never)
    const double density = (double)option_float(options().raster_bound_density);
    const float px_per_ndc = (float)(W < H ? W : H) / 2.0f;  // (both axes: PixToNonSquareNdc keeps pixels square)
    const float rpx = radius * px_per_ndc;
    const int b = rpx >= 5.05f ? 4 : (rpx >= 2.25f ? 2 : 0);
    gate_rows = (int64_t)(density * (double)H * (double)W);
    // (a host count, or a capacity, below the gate: no launch at all)
    if (n_points > 0 && n_points >= gate_rows && b != 0 && K <= b * b) {
      const int64_t n16 = cdiv((int64_t)H * W * 4, 16);  // (the region is padded to 256 bytes)
      PGDVS_LAUNCH("raster_zmin_init", raster_zmin_init_kernel, dim3((unsigned)(cdiv(n16, 256) < 2048 ? cdiv(n16, 256) : 2048)), dim3(256),
                   0, st, n_points, n_points_dev, gate_rows, reinterpret_cast<uint4 *>(ws.zmin), n16);
      const unsigned gz = (unsigned)(cdiv(n_points, 256) < 4096 ? cdiv(n_points, 256) : 4096);
      PGDVS_LAUNCH("raster_zmin", raster_zmin_kernel, dim3(gz), dim3(256), 0, st, pts, pts_stride, n_points, n_points_dev, gate_rows,
                   cam_tgt, H, W, ws.zmin);
      if (b == 4) {
        PGDVS_LAUNCH("raster_bound", raster_bound_kernel<4>, dim3((unsigned)ntiles), dim3(256), 0, st, (const unsigned *)ws.zmin,
                     n_points, n_points_dev, gate_rows, H, W, ntx, K, ws.tile_bound);
      } else {
        PGDVS_LAUNCH("raster_bound", raster_bound_kernel<2>, dim3((unsigned)ntiles), dim3(256), 0, st, (const unsigned *)ws.zmin,
                     n_points, n_points_dev, gate_rows, H, W, ntx, K, ws.tile_bound);
      }
      tile_bound = ws.tile_bound;
    }
  }
  if (n_points == 0 && status_dev != nullptr) {  // (row bound 0: nothing runs that could look at the device count)
    PGDVS_LAUNCH("raster_status", raster_status_kernel, dim3(1), dim3(64), 0, st, n_points_dev, status_dev);
  }
  // Direct binning first (round 5): sparse clouds only -- below the density gate, i.e. where the lists are short and the
  // second tile launch is not needed; dense clouds fill 4096-entry segments and go straight to the exact passes.  The exact
  // passes follow either way and return at once unless a segment overflowed (stats[1]).
  const int seg = (n_points > 0 && n_points < gate_rows && ws.seg_lists != nullptr) ? ws.seg_entries : 0;
  const int32_t *run_flag = seg > 0 ? ws.stats + 1 : nullptr;
  const int64_t fill_chunks = cdiv(n_points, (int64_t)kFillThreads * kFillPer);
  const unsigned g_fill = (unsigned)(fill_chunks < 4096 ? fill_chunks : 4096);
  if (seg > 0) {
    if (small_table) {
      PGDVS_LAUNCH("raster_fill", raster_fill_kernel<kBinSlots / 2>, dim3(g_fill), dim3(kFillThreads), 0, st, pts, pts_stride, n_points, n_points_dev,
                   cam_tgt, radius, H, W, ntx, nty, (const int32_t *)nullptr, ws.seg_cursor, ws.seg_lists, ws.list_capacity, tile_bound, gate_rows,
                   seg, ws.stats + 1, status_dev, (const int32_t *)nullptr);
    } else {
      PGDVS_LAUNCH("raster_fill", raster_fill_kernel<kBinSlots>, dim3(g_fill), dim3(kFillThreads), 0, st, pts, pts_stride, n_points, n_points_dev,
                   cam_tgt, radius, H, W, ntx, nty, (const int32_t *)nullptr, ws.seg_cursor, ws.seg_lists, ws.list_capacity, tile_bound, gate_rows,
                   seg, ws.stats + 1, status_dev, (const int32_t *)nullptr);
    }
  }
  // (behind a direct pass the exact passes are launched on one workgroup per CU: a launch of thousands of workgroups that
  // leave at once still occupies its queue for ~5 us, three of them 15 us per view; the rare overflow pays with slower
  // exact passes)
  if (n_points > 0) {
    unsigned g = (unsigned)(cdiv(n_points, kBinThreads) < 512 ? cdiv(n_points, kBinThreads) : 512);
    if (seg > 0 && g > 256) g = 256;
    if (small_table) {
      PGDVS_LAUNCH(seg > 0 ? "raster_exact_count" : "raster_project_count", raster_project_count_kernel<kBinSlots / 2>, dim3(g), dim3(kBinThreads), 0, st, pts,
                   pts_stride, n_points, n_points_dev, cam_tgt, radius, H, W, ntx, nty, ws.tile_count, status_dev, tile_bound, gate_rows, run_flag);
    } else {
      PGDVS_LAUNCH(seg > 0 ? "raster_exact_count" : "raster_project_count", raster_project_count_kernel<kBinSlots>, dim3(g), dim3(kBinThreads), 0, st, pts,
                   pts_stride, n_points, n_points_dev, cam_tgt, radius, H, W, ntx, nty, ws.tile_count, status_dev, tile_bound, gate_rows, run_flag);
    }
  }
  PGDVS_LAUNCH(seg > 0 ? "raster_exact_scan" : "raster_scan", raster_scan_kernel, dim3(1), dim3(1024), 0, st, ws.tile_count, ntiles, ws.offsets, run_flag);
  if (n_points > 0) {
    const unsigned g_exact_fill = seg > 0 && g_fill > 256 ? 256u : g_fill;
    if (small_table) {
      PGDVS_LAUNCH(seg > 0 ? "raster_exact_fill" : "raster_fill", raster_fill_kernel<kBinSlots / 2>, dim3(g_exact_fill), dim3(kFillThreads), 0, st, pts, pts_stride, n_points, n_points_dev, cam_tgt,
                   radius, H, W, ntx, nty, (const int32_t *)ws.offsets, ws.cursor, ws.lists, ws.list_capacity, tile_bound, gate_rows,
                   0, (int32_t *)nullptr, (int32_t *)nullptr, run_flag);
    } else {
      PGDVS_LAUNCH(seg > 0 ? "raster_exact_fill" : "raster_fill", raster_fill_kernel<kBinSlots>, dim3(g_exact_fill), dim3(kFillThreads), 0, st, pts, pts_stride, n_points, n_points_dev, cam_tgt,
                   radius, H, W, ntx, nty, (const int32_t *)ws.offsets, ws.cursor, ws.lists, ws.list_capacity, tile_bound, gate_rows,
                   0, (int32_t *)nullptr, (int32_t *)nullptr, run_flag);
    }
  }
  const int tiles_per_xcd = (int)cdiv(ntiles, 8);  // (unused by the kernel since round 5: see raster_tile_kernel)
  dim3 grid((unsigned)ntiles);
  // dense clouds (the same density from which the depth bound is computed): a second tile launch with a 4096-entry sorted
  // path takes the lists the first one cannot hold
  const bool long_lists = n_points > 0 && n_points >= gate_rows;
#define PGDVS_TILE_CASE(KK)                                                                      \
  case KK:                                                                                       \
    launch_tile<KK>(grid, st, ws, feat, feat_stride, radius, H, W, ntx, nty, tiles_per_xcd, idx, \
                    zbuf, dist2, rgb, rgb_planar, mask, long_lists, seg);                        \
    break;
  switch (K) {
    PGDVS_TILE_CASE(1)
    PGDVS_TILE_CASE(2)
    PGDVS_TILE_CASE(3)
    PGDVS_TILE_CASE(4)
    PGDVS_TILE_CASE(5)
    PGDVS_TILE_CASE(6)
    PGDVS_TILE_CASE(7)
    PGDVS_TILE_CASE(8)
  }
#undef PGDVS_TILE_CASE
  return check_launch("points_raster");
}

