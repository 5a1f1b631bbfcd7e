// Ordered stream compaction (the device-side replacement of boolean-mask indexing
// and torch.nonzero in pgdvs_renderer_dyn.py:309-320,477 and of the static-mask
// selection in nvidia_eval_pure_geo.py:247-248).  Three short launches, no host
// sync: per-block popcounts -> single-block scan of the block totals -> ordered
// scatter using wave ballots.  Output order is ascending input position, i.e.
// the row-major pixel order the reference gets from boolean indexing.
#include "scan.h"

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
