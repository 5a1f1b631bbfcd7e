// A4: statistical outlier filter.
//   knn_mean_dist : exact brute-force K+1 nearest neighbours by squared L2 (the
//                   contract of pytorch3d.ops.knn_points, pgdvs_renderer_dyn.py:405-419),
//                   reduced on the fly to the mean of the K non-self distances.
//   outlier_flags : lower median (torch.median), unbiased std (torch.std), threshold
//                   and `avg < thres` flags (pgdvs_renderer_dyn.py:419-427).
// The mean of the K smallest distances does not depend on how equal distances are
// ordered, so no index bookkeeping is needed; values are summed in ascending order
// exactly like the CPU oracle so that results agree bit-for-bit.
//
// VALU-bound (N^2 distance evaluations), not HBM-bound: candidates are staged through
// LDS once per block and broadcast to all lanes; each lane keeps its K+1 best in an
// LDS column (conflict-free: column index = thread id) with the current maximum in
// registers, so the common case per candidate is sub/mul/add/compare only.
#include "common.h"

namespace pgdvs {

constexpr int kKnnBlock = 256;
constexpr int kKnnMaxKK = 128;

__global__ void __launch_bounds__(kKnnBlock)
knn_mean_dist_kernel(const float *__restrict__ pts, const int32_t *__restrict__ count, int K,
                     float *__restrict__ avg_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int KK = K + 1;
  float4 *cand = reinterpret_cast<float4 *>(smem);            // [kKnnBlock]
  float *best = smem + kKnnBlock * 4;                          // [KK][kKnnBlock]
  const int n = *count;
  const int q0 = blockIdx.x * kKnnBlock;
  if (q0 >= n) return;
  const int tid = threadIdx.x;
  const int q = q0 + tid;
  const bool active = q < n;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (active) {
    qx = pts[(size_t)q * 3];
    qy = pts[(size_t)q * 3 + 1];
    qz = pts[(size_t)q * 3 + 2];
  }
  const int ntiles = (n + kKnnBlock - 1) / kKnnBlock;
  int cnt = 0;       // number of stored values (uniform across active lanes)
  float mx = 0.0f;   // current maximum among the stored values once cnt == KK
  int mxk = 0;
  for (int s = 0; s < ntiles; ++s) {
    // start with the block's own tile: neighbours in raster order are near in space,
    // so the running threshold tightens immediately
    int t = blockIdx.x + s;
    if (t >= ntiles) t -= ntiles;
    int j = t * kKnnBlock + tid;
    __syncthreads();
    if (j < n) {
      cand[tid] = make_float4(pts[(size_t)j * 3], pts[(size_t)j * 3 + 1], pts[(size_t)j * 3 + 2], 0.f);
    }
    __syncthreads();
    int m = n - t * kKnnBlock;
    if (m > kKnnBlock) m = kKnnBlock;
    int c = 0;
    // warm-up: fill the first KK slots (same trip count for every lane)
    for (; c < m && cnt < KK; ++c) {
      float4 p = cand[c];
      float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
      float d = dx * dx;
      d = d + dy * dy;
      d = d + dz * dz;
      best[cnt * kKnnBlock + tid] = d;
      if (cnt == 0 || d > mx) {
        mx = d;
        mxk = cnt;
      }
      ++cnt;
    }
#pragma unroll 4
    for (; c < m; ++c) {
      float4 p = cand[c];
      float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
      float d = dx * dx;
      d = d + dy * dy;
      d = d + dz * dz;
      if (d < mx) {
        best[mxk * kKnnBlock + tid] = d;
        float nm = best[tid];
        int nk = 0;
        for (int k = 1; k < KK; ++k) {
          float v = best[k * kKnnBlock + tid];
          if (v > nm) {
            nm = v;
            nk = k;
          }
        }
        mx = nm;
        mxk = nk;
      }
    }
  }
  if (!active) return;
  // ascending insertion sort of the cnt stored values (own LDS column)
  for (int i = 1; i < cnt; ++i) {
    float key = best[i * kKnnBlock + tid];
    int k = i - 1;
    while (k >= 0 && best[k * kKnnBlock + tid] > key) {
      best[(k + 1) * kKnnBlock + tid] = best[k * kKnnBlock + tid];
      --k;
    }
    best[(k + 1) * kKnnBlock + tid] = key;
  }
  // drop column 0 (the point itself), mean over K columns; missing columns are 0.  Same fixed
  // butterfly summation order as the grid kernel / CPU oracle (64 slots, offsets 32..1).
  // The column is reused as scratch: slots [0,64) when KK <= 64, else a plain ascending sum.
  float ssum = 0.0f;
  if (KK <= 64) {
    float sl[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) sl[k] = 0.0f;
    for (int k = 1; k < KK; ++k) sl[k] = k < cnt ? best[k * kKnnBlock + tid] : 0.0f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int i = 0; i < off; ++i) sl[i] = sl[i] + sl[i + off];
    ssum = sl[0];
  } else {
    for (int k = 1; k < KK; ++k) ssum = ssum + (k < cnt ? best[k * kKnnBlock + tid] : 0.0f);
  }
  avg_out[q] = ssum / (float)K;
}

// ---------------------------------------------------------------------------
// single-block statistics: mean / unbiased std in fp64, lower median by radix select
// ---------------------------------------------------------------------------
constexpr int kStatBlock = 1024;

__device__ double block_sum_f64(double v, double *scratch) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[wave] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < kStatBlock / kWave; ++i) s += scratch[i];
  return s;
}

__global__ void __launch_bounds__(kStatBlock)
outlier_threshold_kernel(const float *__restrict__ avg, const int32_t *__restrict__ count,
                         float std_thres, float *__restrict__ thres_out) {
  __shared__ double scratch[kStatBlock / kWave];
  __shared__ unsigned int hist[256];
  __shared__ unsigned int sel_prefix;
  __shared__ unsigned int sel_rank;
  const int n = *count;
  const int tid = threadIdx.x;
  if (n <= 0) {
    if (tid == 0) *thres_out = __builtin_nanf("");
    return;
  }
  double s = 0.0;
  for (int i = tid; i < n; i += kStatBlock) s += (double)avg[i];
  double mean = block_sum_f64(s, scratch) / (double)n;
  double m2 = 0.0;
  for (int i = tid; i < n; i += kStatBlock) {
    double d = (double)avg[i] - mean;
    m2 += d * d;
  }
  m2 = block_sum_f64(m2, scratch);
  // lower median = element of rank (n-1)/2 in ascending order.  Values are >= 0 (or
  // NaN); map float bits to an order-preserving unsigned key.
  if (tid == 0) {
    sel_prefix = 0;
    sel_rank = (unsigned)((n - 1) / 2);
  }
  __syncthreads();
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += kStatBlock) hist[i] = 0;
    __syncthreads();
    unsigned prefix = sel_prefix;
    unsigned himask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
    const int n_round = (n + kStatBlock - 1) / kStatBlock * kStatBlock;  // whole waves stay in the loop
    for (int i = tid; i < n_round; i += kStatBlock) {
      unsigned u = i < n ? __float_as_uint(avg[i]) : 0u;
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      bool act = i < n && (u & himask) == (prefix & himask);
      unsigned bin = (u >> shift) & 0xffu;
      // neighbouring values share their leading bits: one LDS atomic per distinct bin per wave
      unsigned long long todo = __ballot(act);
      while (todo) {
        int l = __builtin_ctzll(todo);
        unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)bin, l);
        unsigned long long same = __ballot(act && bin == b0);
        if ((tid & 63) == l) atomicAdd(&hist[b0], (unsigned)__popcll(same));
        todo &= ~same;
      }
    }
    __syncthreads();
    if (tid == 0) {
      unsigned r = sel_rank, acc = 0;
      int b = 0;
      for (; b < 256; ++b) {
        if (acc + hist[b] > r) break;
        acc += hist[b];
      }
      if (b > 255) b = 255;
      sel_rank = r - acc;
      sel_prefix = prefix | ((unsigned)b << shift);
    }
    __syncthreads();
  }
  if (tid == 0) {
    unsigned u = sel_prefix;
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float med = __uint_as_float(u);
    float sd = n > 1 ? (float)sqrt(m2 / (double)(n - 1)) : __builtin_nanf("");
    *thres_out = med + sd * std_thres;
  }
}

__global__ void outlier_flag_kernel(const float *__restrict__ avg, const int32_t *__restrict__ count,
                                    const float *__restrict__ thres, int remove_outlier,
                                    uint8_t *__restrict__ flag) {
  int64_t n = *count;
  float t = *thres;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    flag[i] = remove_outlier ? (uint8_t)(avg[i] < t) : (uint8_t)1;
}

int64_t knn_grid_workspace_bytes(int64_t capacity);
int knn_grid_mean_dist(const float *pts, const int32_t *count, int64_t capacity, int K, float *avg_out,
                       void *workspace, int64_t workspace_bytes, hipStream_t st);

}  // namespace pgdvs

using namespace pgdvs;

PGDVS_API int64_t pgdvs_knn_workspace_bytes(int64_t capacity) {
  return knn_grid_workspace_bytes(capacity);
}

PGDVS_API int pgdvs_knn_mean_dist(const float *pts, const int32_t *count, int64_t capacity, int K,
                                  float *avg_out, int algo, void *workspace,
                                  int64_t workspace_bytes, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(pts && count && avg_out, "pgdvs_knn_mean_dist: null pointer");
  PGDVS_REQUIRE(algo >= 0 && algo <= 2, "pgdvs_knn_mean_dist: algo must be 0, 1 or 2");
  PGDVS_REQUIRE(algo != 2 || K + 1 <= 64, "pgdvs_knn_mean_dist: grid search needs K+1 <= 64");
  if (capacity > 0 && K >= 1 && K + 1 <= 64 && algo != 1 && capacity < (1ll << 31))
    return knn_grid_mean_dist(pts, count, capacity, K, avg_out, workspace, workspace_bytes,
                              as_stream(stream));
  PGDVS_REQUIRE(K >= 1 && K + 1 <= kKnnMaxKK, "pgdvs_knn_mean_dist: K must be in [1, %d]",
                kKnnMaxKK - 1);
  PGDVS_REQUIRE(capacity >= 0 && capacity < (1ll << 31), "pgdvs_knn_mean_dist: bad capacity");
  if (capacity == 0) return PGDVS_OK;
  size_t lds = (size_t)kKnnBlock * 4 * sizeof(float) + (size_t)(K + 1) * kKnnBlock * sizeof(float);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(knn_mean_dist_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("knn_mean_dist: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
      return PGDVS_ERR_LAUNCH;
    }
  }
  PGDVS_LAUNCH("knn_mean_dist", knn_mean_dist_kernel, dim3((unsigned)cdiv(capacity, kKnnBlock)),
                     dim3(kKnnBlock), lds, as_stream(stream), pts, count, K, avg_out);
  return check_launch("knn_mean_dist");
}

PGDVS_API int64_t pgdvs_outlier_workspace_bytes(int64_t capacity) {
  (void)capacity;
  return 256;
}

PGDVS_API int pgdvs_outlier_flags(const float *avg, const int32_t *count, int64_t capacity,
                                  float std_thres, int remove_outlier, float *thres_out,
                                  uint8_t *flag_out, void *workspace, int64_t workspace_bytes,
                                  pgdvs_stream_t stream) {
  (void)workspace;
  (void)workspace_bytes;
  PGDVS_REQUIRE(avg && count && thres_out && flag_out && capacity >= 0,
                "pgdvs_outlier_flags: bad arguments");
  PGDVS_LAUNCH("outlier_threshold", outlier_threshold_kernel, dim3(1), dim3(kStatBlock), 0, as_stream(stream), avg,
                     count, std_thres, thres_out);
  if (capacity > 0) {
    unsigned grid = (unsigned)(cdiv(capacity, 256) < 1024 ? cdiv(capacity, 256) : 1024);
    PGDVS_LAUNCH("outlier_flag", outlier_flag_kernel, dim3(grid), dim3(256), 0, as_stream(stream), avg, count,
                       thres_out, remove_outlier, flag_out);
  }
  return check_launch("outlier_flags");
}
