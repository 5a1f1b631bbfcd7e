// Ordered stream compaction (the device-side replacement of boolean-mask indexing
// and torch.nonzero in pgdvs_renderer_dyn.py:309-320,477 and of the static-mask
// selection in nvidia_eval_pure_geo.py:247-248).  Three short launches, no host
// sync: per-block popcounts -> single-block scan of the block totals -> ordered
// scatter using wave ballots.  Output order is ascending input position, i.e.
// the row-major pixel order the reference gets from boolean indexing.
#include "scan.h"

#include "fused.h"

namespace pgdvs {

constexpr int kCompactBlock = 1024;             // threads per block
constexpr int kCompactItems = 4;                // flags per thread
constexpr int kCompactTile = kCompactBlock * kCompactItems;

__global__ void __launch_bounds__(kCompactBlock)
compact_count_kernel(const uint8_t *__restrict__ flags, int64_t n, int32_t *__restrict__ block_counts) {
  __shared__ int wave_sums[kCompactBlock / kWave];
  int64_t base = (int64_t)blockIdx.x * kCompactTile + (int64_t)threadIdx.x * kCompactItems;
  int c = 0;
  const bool aligned = (reinterpret_cast<uintptr_t>(flags) & 3) == 0;
  if (aligned && base + kCompactItems <= n) {
    uint32_t w = *reinterpret_cast<const uint32_t *>(flags + base);  // 4-aligned: base % 4 == 0
    c = ((w & 0xffu) != 0) + ((w & 0xff00u) != 0) + ((w & 0xff0000u) != 0) + ((w & 0xff000000u) != 0);
  } else {
    for (int k = 0; k < kCompactItems; ++k)
      if (base + k < n) c += flags[base + k] != 0;
  }
  // wave reduce
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) wave_sums[wave] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int i = 0; i < kCompactBlock / kWave; ++i) s += wave_sums[i];
    block_counts[blockIdx.x] = s;
  }
}

// exclusive scan of block_counts[nb] in place -> block_offsets; total -> count_out
__global__ void __launch_bounds__(1024)
compact_scan_kernel(int32_t *__restrict__ block_counts, int nb, int32_t *__restrict__ count_out) {
  __shared__ int wave_sums[1024 / kWave];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int start = 0; start < nb; start += 1024) {
    int i = start + threadIdx.x;
    int v = i < nb ? block_counts[i] : 0;
    int x = v;
    for (int off = 1; off < 64; off <<= 1) {
      int y = __shfl_up(x, off, 64);
      if ((threadIdx.x & 63) >= off) x += y;
    }
    int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wave_sums[wave] = x;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += wave_sums[w];
    int incl = carry + wave_off + x;
    if (i < nb) block_counts[i] = incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *count_out = carry;
}

__global__ void __launch_bounds__(kCompactBlock)
compact_scatter_kernel(const uint8_t *__restrict__ flags, int64_t n,
                       const int32_t *__restrict__ block_offsets, int32_t *__restrict__ idx_out) {
  __shared__ int wave_sums[kCompactBlock / kWave];
  int64_t base = (int64_t)blockIdx.x * kCompactTile + (int64_t)threadIdx.x * kCompactItems;
  bool f[kCompactItems];
  int c = 0;
#pragma unroll
  for (int k = 0; k < kCompactItems; ++k) {
    f[k] = (base + k < n) && flags[base + k] != 0;
    c += f[k];
  }
  int x = c;
  for (int off = 1; off < 64; off <<= 1) {
    int y = __shfl_up(x, off, 64);
    if ((threadIdx.x & 63) >= off) x += y;
  }
  int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 63) wave_sums[wave] = x;
  __syncthreads();
  int wave_off = 0;
  for (int w = 0; w < wave; ++w) wave_off += wave_sums[w];
  int pos = block_offsets[blockIdx.x] + wave_off + x - c;
#pragma unroll
  for (int k = 0; k < kCompactItems; ++k)
    if (f[k]) idx_out[pos++] = (int32_t)(base + k);
}

// The same ordered compaction in ONE launch when somebody has already counted the flags per 256-element chunk (the per-view
// call: dyn_warp_kernel counts the valid pixels of its own workgroup): a tile of kCompactTile = 4096 flags spans 16 chunks,
// every workgroup adds up the chunk counts before its tile itself (at most n / 256 values of an L2-resident array: 8 per
// thread at 1080p), scatters the indices, writes the gathered rows rows_out[i] = rows_in[idx[i]] (3 floats: what
// pgdvs_gather_rows did in a launch of its own) and folds their bounding box into bbox[6] (what grid_bbox_kernel did: ~min /
// max in the order-preserving encoding, finite values only; six atomics per workgroup that holds a point).  The last tile
// leaves the count.
__global__ void __launch_bounds__(kCompactBlock)
compact_gather_bbox_kernel(const uint8_t *__restrict__ flags, int64_t n, const int32_t *__restrict__ chunk_cnt,
                           int32_t *__restrict__ idx_out, int32_t *__restrict__ count_out, const float *__restrict__ rows_in,
                           float *__restrict__ rows_out, unsigned *__restrict__ bbox, int32_t *__restrict__ zero_per_row,
                           int zero_mult) {
  constexpr int kWaves = kCompactBlock / kWave;
  __shared__ int wave_sums[kWaves];
  __shared__ int pre_sums[kWaves];
  __shared__ float s_mn[kWaves][3], s_mx[kWaves][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  static_assert(kCompactTile % 256 == 0, "whole chunks per tile");
  const int chunks_before = (int)blockIdx.x * (kCompactTile / 256);
  int pre = 0;
  for (int c = tid; c < chunks_before; c += kCompactBlock) pre += chunk_cnt[c];
  for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off, 64);
  if (lane == 0) pre_sums[wave] = pre;
  const int64_t base = (int64_t)blockIdx.x * kCompactTile + (int64_t)tid * kCompactItems;
  bool f[kCompactItems];
  int c = 0;
#pragma unroll
  for (int k = 0; k < kCompactItems; ++k) {
    f[k] = (base + k < n) && flags[base + k] != 0;
    c += f[k];
  }
  int x = c;
  for (int off = 1; off < 64; off <<= 1) {
    int y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) wave_sums[wave] = x;
  __syncthreads();
  int tile_off = 0, wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    tile_off += pre_sums[w];
    wave_off += w < wave ? wave_sums[w] : 0;
    total += wave_sums[w];
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) {
    *count_out = tile_off + total;
    if (zero_per_row) zero_per_row[(int64_t)(tile_off + total) * zero_mult] = 0;
  }
  if (total == 0) return;  // (uniform: no barrier follows for this workgroup)
  if (zero_per_row)
    for (int i = tid; i < total * zero_mult; i += kCompactBlock) zero_per_row[(int64_t)tile_off * zero_mult + i] = 0;
  int pos = tile_off + wave_off + x - c;
  float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  float mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
  for (int k = 0; k < kCompactItems; ++k)
    if (f[k]) {
      idx_out[pos] = (int32_t)(base + k);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float v = rows_in[(size_t)(base + k) * 3 + a];
        rows_out[(size_t)pos * 3 + a] = v;
        if (isfinite(v)) {
          mn[a] = fminf(mn[a], v);
          mx[a] = fmaxf(mx[a], v);
        }
      }
      ++pos;
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    for (int off = 32; off > 0; off >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
    }
    if (lane == 0) {
      s_mn[wave][a] = mn[a];
      s_mx[wave][a] = mx[a];
    }
  }
  __syncthreads();
  if (tid < 3) {
    float lo = __builtin_inff(), hi = -__builtin_inff();
    for (int w = 0; w < kWaves; ++w) {
      lo = fminf(lo, s_mn[w][tid]);
      hi = fmaxf(hi, s_mx[w][tid]);
    }
    // (an all-infinite pair -- no finite coordinate in the tile -- encodes below every finite value: no effect)
    if (lo <= hi) {
      atomicMax(&bbox[tid], ~f2ord(lo));
      atomicMax(&bbox[3 + tid], f2ord(hi));
    }
  }
}

int compact_gather_bbox(const uint8_t *flags, int64_t n, const int32_t *chunk_cnt, int32_t *idx_out, int32_t *count_out,
                        const float *rows_in, float *rows_out, unsigned *bbox, int32_t *zero_per_row, int zero_mult, hipStream_t st) {
  if (n <= 0 || n >= (1ll << 31)) {
    set_error("compact_gather_bbox: n out of range");
    return PGDVS_ERR_INVALID;
  }
  const int nb = (int)cdiv(n, kCompactTile);
  PGDVS_LAUNCH("compact_gather_bbox", compact_gather_bbox_kernel, dim3(nb), dim3(kCompactBlock), 0, st, flags, n, chunk_cnt, idx_out,
               count_out, rows_in, rows_out, bbox, zero_per_row, zero_mult);
  return check_launch("compact_gather_bbox");
}

int64_t compact_workspace_bytes(int64_t n) {
  int64_t nb = cdiv(n > 0 ? n : 1, kCompactTile);
  return align_up(nb * (int64_t)sizeof(int32_t), 256);
}

int compact_u8(const uint8_t *flags, int64_t n, int32_t *idx_out, int32_t *count_out,
               void *workspace, int64_t workspace_bytes, hipStream_t stream) {
  if (n < 0 || n >= (1ll << 31)) {
    set_error("compact_u8: n out of range");
    return PGDVS_ERR_INVALID;
  }
  if (workspace_bytes < compact_workspace_bytes(n)) {
    set_error("compact_u8: workspace too small");
    return PGDVS_ERR_WORKSPACE;
  }
  if (n == 0) {
    (void)hipMemsetAsync(count_out, 0, sizeof(int32_t), stream);
    return check_launch("compact memset");
  }
  int nb = (int)cdiv(n, kCompactTile);
  int32_t *block_counts = reinterpret_cast<int32_t *>(workspace);
  PGDVS_LAUNCH("compact_count", compact_count_kernel, dim3(nb), dim3(kCompactBlock), 0, stream, flags, n,
                     block_counts);
  PGDVS_LAUNCH("compact_scan", compact_scan_kernel, dim3(1), dim3(1024), 0, stream, block_counts, nb,
                     count_out);
  PGDVS_LAUNCH("compact_scatter", compact_scatter_kernel, dim3(nb), dim3(kCompactBlock), 0, stream, flags, n,
                     block_counts, idx_out);
  return check_launch("compact_u8");
}

}  // namespace pgdvs

using namespace pgdvs;

PGDVS_API int64_t pgdvs_compact_workspace_bytes(int64_t n) { return compact_workspace_bytes(n); }

PGDVS_API int pgdvs_compact_u8(const uint8_t *flags, int64_t n, int32_t *idx_out,
                               int32_t *count_out, void *workspace, int64_t workspace_bytes,
                               pgdvs_stream_t stream) {
  PGDVS_REQUIRE(flags && idx_out && count_out && workspace, "pgdvs_compact_u8: null pointer");
  return compact_u8(flags, n, idx_out, count_out, workspace, workspace_bytes, as_stream(stream));
}
