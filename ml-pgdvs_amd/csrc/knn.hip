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
#include "fused.h"

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
// statistics for the threshold: mean / unbiased std accumulated in fp64 with a fixed
// reduction order (deterministic), lower median by a 3-pass radix select (11+11+10 bits of
// the order-preserving key).  Multi-block passes with block-local LDS histograms, tiny
// single-block kernels in between to pick the bin.
// ---------------------------------------------------------------------------
constexpr int kStatBlocks = 120;
constexpr int kStatThreads = 256;
constexpr int kStatBins = 2048;

struct StatState {
  double mean, m2;
  unsigned prefix, rank;
  int n;
};

__device__ __forceinline__ unsigned stat_key(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// after pass p: fold the partial sums (fixed tree order), pick the bin holding the wanted rank.  Runs on ONE 256-thread
// workgroup (all of its threads call it): as a kernel of its own between the passes (the per-op entry point), or -- the
// per-view call, round 6 -- at the head of EVERY workgroup of the launch that needs its result (three launches less on the
// dynamic branch's chain: every workgroup reads the same 120 partial sums and 2048 bins and computes the same state).
// `in`: the state the previous selection left (unused for pass 0); the result goes to *out (shared or global memory) and, for
// pass 2, the threshold to *thres_out.  Ends with a barrier.
__device__ __forceinline__ void stat_select_block(const int n, const int pass, const int nblocks,
                                                  const double *__restrict__ partials, const unsigned *__restrict__ ghist,
                                                  const StatState in, StatState *out, const float std_thres, float *thres_out) {
  __shared__ unsigned wtot[4];
  __shared__ double dsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned rank_in = pass == 0 ? (unsigned)(n > 0 ? (n - 1) / 2 : 0) : in.rank;  // torch.median: lower median
  const unsigned prefix_in = pass == 0 ? 0u : in.prefix;
  const double m2_in = in.m2;
  // sum of the per-block partials (nblocks <= 256), fixed shuffle tree
  double t = (pass < 2 && tid < nblocks) ? partials[tid] : 0.0;
  for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
  if (lane == 0) dsum[wave] = t;
  // 2048 bins, 8 per thread: find the bin where the running count passes rank
  const unsigned *h = ghist + pass * kStatBins;
  const int nb = pass == 2 ? 1024 : 2048;
  unsigned loc[8], s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    loc[k] = tid * 8 + k < nb ? h[tid * 8 + k] : 0;
    s += loc[k];
  }
  unsigned x = s;
  for (int off = 1; off < 64; off <<= 1) {
    unsigned y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) wtot[wave] = x;
  __syncthreads();
  unsigned before = x - s;  // counts in the bins of the threads before this one
  for (int w = 0; w < wave; ++w) before += wtot[w];
  const double total = (dsum[0] + dsum[1]) + (dsum[2] + dsum[3]);
  const unsigned all = wtot[0] + wtot[1] + wtot[2] + wtot[3];
  // the thread whose 8 bins contain the rank (the last thread takes it if the histogram is short)
  const bool owner = (before <= rank_in && rank_in < before + s) || (tid == 255 && rank_in >= all);
  if (owner) {
    unsigned acc = before;
    int k = 0;
    for (; k < 7; ++k) {
      if (acc + loc[k] > rank_in) break;
      acc += loc[k];
    }
    int bsel = tid * 8 + k;
    if (bsel >= nb) bsel = nb - 1;
    const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
    const unsigned prefix = prefix_in | ((unsigned)bsel << shift);
    StatState o = in;
    o.rank = rank_in - acc;
    o.prefix = prefix;
    if (pass == 0) {
      o.mean = n > 0 ? total / (double)n : 0.0;
      o.n = n;
    } else if (pass == 1) {
      o.m2 = total;
    } else {
      unsigned u = (prefix & 0x80000000u) ? (prefix & 0x7fffffffu) : ~prefix;
      float med = __uint_as_float(u);
      float sd = n > 1 ? (float)sqrt(m2_in / (double)(n - 1)) : __builtin_nanf("");
      *thres_out = n > 0 ? med + sd * std_thres : __builtin_nanf("");
    }
    *out = o;
  }
  __syncthreads();
}

// pass: 0 -> accumulate sum(x) and histogram key bits [31:21]
//       1 -> accumulate sum((x-mean)^2) and histogram bits [20:10] of keys matching prefix[31:21]
//       2 -> histogram bits [9:0] of keys matching prefix[31:10]
// kFused: the selection after pass - 1 runs at the head of this launch (stat_select_block; `partials` then holds one array of
// kStatBlocks sums per pass, so that a workgroup that is already writing its sum cannot disturb one that still selects);
// workgroup 0 leaves the state in *st_out for the launch behind this one.
template <bool kFused>
__global__ void __launch_bounds__(kStatThreads)
stat_pass_kernel(const float *__restrict__ avg, const int32_t *__restrict__ count, int pass,
                 const StatState *__restrict__ st, double *__restrict__ partials,
                 unsigned *__restrict__ ghist, StatState *__restrict__ st_out) {
  __shared__ unsigned hist[kStatBins];
  __shared__ double wsum[kStatThreads / kWave];
  __shared__ StatState s_st;
  const int n = *count;
  const int tid = threadIdx.x;
  if (kFused && pass > 0) {
    stat_select_block(n, pass - 1, kStatBlocks, partials + (pass - 1) * kStatBlocks, ghist, *st, &s_st, 0.0f, nullptr);
    if (blockIdx.x == 0 && tid == 0) *st_out = s_st;
    partials += pass * kStatBlocks;
  }
  for (int i = tid; i < kStatBins; i += kStatThreads) hist[i] = 0;
  __syncthreads();
  const double mean = pass == 1 ? (kFused ? s_st.mean : st->mean) : 0.0;
  const unsigned prefix = pass > 0 ? (kFused ? s_st.prefix : st->prefix) : 0u;
  const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
  const unsigned himask = pass == 0 ? 0u : (pass == 1 ? 0xffe00000u : 0xfffffc00u);
  const unsigned binmask = pass == 2 ? 0x3ffu : 0x7ffu;
  double acc = 0.0;
  const int stride = gridDim.x * kStatThreads;
  const int n_round = (n + stride - 1) / stride * stride;  // whole waves stay in the loop
  for (int i = blockIdx.x * kStatThreads + tid; i < n_round; i += stride) {
    bool in = i < n;
    float x = in ? avg[i] : 0.0f;
    if (pass == 0) acc += in ? (double)x : 0.0;
    if (pass == 1) {
      double d = (double)x - mean;
      acc += in ? d * d : 0.0;
    }
    unsigned u = stat_key(x);
    bool act = in && (u & himask) == (prefix & himask);
    unsigned bin = (u >> shift) & binmask;
    // neighbouring values often share the bin: one LDS atomic for the whole wave then
    unsigned long long am = __ballot(act);
    if (am) {
      unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)bin, __builtin_ctzll(am));
      unsigned long long same = __ballot(act && bin == b0);
      if (same == am) {
        if ((tid & 63) == (int)__builtin_ctzll(am)) atomicAdd(&hist[b0], (unsigned)__popcll(am));
      } else if (act) {
        atomicAdd(&hist[bin], 1u);
      }
    }
  }
  if (pass < 2) {
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((tid & 63) == 0) wsum[tid >> 6] = acc;
  }
  __syncthreads();
  if (pass < 2 && tid == 0) {
    double t = 0.0;
    for (int w = 0; w < kStatThreads / kWave; ++w) t += wsum[w];
    partials[blockIdx.x] = t;
  }
  for (int i = tid; i < kStatBins; i += kStatThreads)
    if (hist[i]) atomicAdd(&ghist[pass * kStatBins + i], hist[i]);
}

// One 256-thread block, everything in parallel: the kernel sits on the critical chain of the
// dynamic branch three times per view (per-op entry point; the per-view call runs stat_select_block inside its passes).
__global__ void __launch_bounds__(256)
stat_select_kernel(const int32_t *__restrict__ count, int pass, int nblocks,
                   const double *__restrict__ partials, const unsigned *__restrict__ ghist,
                   StatState *__restrict__ st, float std_thres, float *__restrict__ thres_out) {
  const StatState in = *st;
  __syncthreads();  // (every thread holds the old state before its owner overwrites it)
  stat_select_block(*count, pass, nblocks, partials, ghist, in, st, std_thres, thres_out);
}

__global__ void outlier_flag_kernel(const float *__restrict__ avg, const int32_t *__restrict__ count,
                                    int64_t capacity, const float *__restrict__ thres, int remove_outlier,
                                    uint8_t *__restrict__ flag) {
  int64_t n = *count;
  float t = *thres;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < capacity;
       i += (int64_t)gridDim.x * blockDim.x)
    flag[i] = i < n ? (remove_outlier ? (uint8_t)(avg[i] < t) : (uint8_t)1) : (uint8_t)0;
}

// The per-view call's last launch of the filter: the selection after pass 2 (the threshold) at the head of every workgroup,
// then keep[idx[i]] = 1 for every point below it -- what outlier_flag_kernel + a fill + scatter_keep_kernel did in three
// launches (keep is cleared by the caller: dyn_warp_kernel's extras).
__global__ void __launch_bounds__(256)
outlier_keep_kernel(const float *__restrict__ avg, const int32_t *__restrict__ count, const StatState *__restrict__ st,
                    const double *__restrict__ partials, const unsigned *__restrict__ ghist, float std_thres,
                    float *__restrict__ thres_out, StatState *__restrict__ st_out, const int32_t *__restrict__ idx,
                    uint8_t *__restrict__ keep) {
  __shared__ StatState s_st;
  __shared__ float s_thres;
  const int n = *count;
  stat_select_block(n, 2, kStatBlocks, partials, ghist, *st, &s_st, std_thres, &s_thres);
  const float t = s_thres;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *thres_out = t;
    *st_out = s_st;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if (avg[i] < t) keep[idx[i]] = 1;
}

int64_t knn_grid_workspace_bytes(int64_t capacity, int64_t qcapacity);
int knn_grid_mean_dist(const float *pts, const int32_t *count, int64_t capacity, int K, float *avg_out,
                       void *workspace, int64_t workspace_bytes, hipStream_t st, const float *qpts,
                       const int32_t *qcount, int64_t qcapacity);

}  // namespace pgdvs

using namespace pgdvs;

PGDVS_API int64_t pgdvs_knn_workspace_bytes(int64_t capacity) {
  return knn_grid_workspace_bytes(capacity, 0);
}

PGDVS_API int pgdvs_knn_mean_dist(const float *pts, const int32_t *count, int64_t capacity, int K,
                                  float *avg_out, int algo, void *workspace,
                                  int64_t workspace_bytes, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(pts && count && avg_out, "pgdvs_knn_mean_dist: null pointer");
  PGDVS_REQUIRE(algo >= 0 && algo <= 2, "pgdvs_knn_mean_dist: algo must be 0, 1 or 2");
  PGDVS_REQUIRE(algo != 2 || K + 1 <= 64, "pgdvs_knn_mean_dist: grid search needs K+1 <= 64");
  if (capacity > 0 && K >= 1 && K + 1 <= 64 && algo != 1 && capacity < (1ll << 31))
    return knn_grid_mean_dist(pts, count, capacity, K, avg_out, workspace, workspace_bytes,
                              as_stream(stream), nullptr, nullptr, 0);
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

PGDVS_API int64_t pgdvs_knn_cross_workspace_bytes(int64_t capacity, int64_t query_capacity) {
  return knn_grid_workspace_bytes(capacity, query_capacity);
}

PGDVS_API int pgdvs_knn_cross_mean_dist(const float *queries, const int32_t *query_count,
                                        int64_t query_capacity, const float *pts, const int32_t *count,
                                        int64_t capacity, int KK, float *avg_out, void *workspace,
                                        int64_t workspace_bytes, pgdvs_stream_t stream) {
  PGDVS_REQUIRE(queries && query_count && pts && count && avg_out, "pgdvs_knn_cross_mean_dist: null pointer");
  PGDVS_REQUIRE(KK >= 1 && KK <= 64, "pgdvs_knn_cross_mean_dist: KK must be in [1, 64]");
  PGDVS_REQUIRE(capacity > 0 && capacity < (1ll << 31) && query_capacity >= 0 && query_capacity < (1ll << 31),
                "pgdvs_knn_cross_mean_dist: bad capacity");
  if (query_capacity == 0) return PGDVS_OK;
  return knn_grid_mean_dist(pts, count, capacity, KK - 1, avg_out, workspace, workspace_bytes,
                            as_stream(stream), queries, query_count, query_capacity);
}

// workspace: [0, 256) two StatState slots (the fused passes ping-pong between them) | partial sums, one array of kStatBlocks per
// pass (the per-op path uses the first) | the three histograms | spare
PGDVS_API int64_t pgdvs_outlier_workspace_bytes(int64_t capacity) {
  (void)capacity;
  return 256 + 2 * kStatBlocks * 8 + 3 * kStatBins * 4 + 256;
}

namespace pgdvs {
void outlier_hist_block(void *workspace, void **block, int64_t *bytes) {
  *block = reinterpret_cast<char *>(workspace) + 256 + 2 * kStatBlocks * 8;
  *bytes = 3 * kStatBins * 4;
}

int outlier_keep_fused(const float *avg, const int32_t *count, int64_t capacity, float std_thres, float *thres_out,
                       const int32_t *idx, uint8_t *keep, void *workspace, int64_t workspace_bytes, hipStream_t st) {
  if (!workspace || workspace_bytes < pgdvs_outlier_workspace_bytes(capacity)) {
    set_error("outlier_keep_fused: workspace too small");
    return PGDVS_ERR_WORKSPACE;
  }
  char *p = reinterpret_cast<char *>(workspace);
  static_assert(sizeof(StatState) <= 128, "two state slots in the first 256 bytes");
  StatState *s0 = reinterpret_cast<StatState *>(p), *s1 = reinterpret_cast<StatState *>(p + 128);
  double *partials = reinterpret_cast<double *>(p + 256);
  unsigned *ghist = reinterpret_cast<unsigned *>(p + 256 + 2 * kStatBlocks * 8);
  // (a launch reads the state its predecessor's workgroup 0 wrote and writes the other slot: no workgroup of a launch can
  // overwrite what a slower one still reads)
  PGDVS_LAUNCH("stat_pass", stat_pass_kernel<true>, dim3(kStatBlocks), dim3(kStatThreads), 0, st, avg, count, 0,
               (const StatState *)s0, partials, ghist, s1);
  PGDVS_LAUNCH("stat_pass", stat_pass_kernel<true>, dim3(kStatBlocks), dim3(kStatThreads), 0, st, avg, count, 1,
               (const StatState *)s0, partials, ghist, s1);
  PGDVS_LAUNCH("stat_pass", stat_pass_kernel<true>, dim3(kStatBlocks), dim3(kStatThreads), 0, st, avg, count, 2,
               (const StatState *)s1, partials, ghist, s0);
  if (capacity > 0) {
    const unsigned grid = (unsigned)(cdiv(capacity, 256) < 1024 ? cdiv(capacity, 256) : 1024);
    PGDVS_LAUNCH("outlier_keep", outlier_keep_kernel, dim3(grid), dim3(256), 0, st, avg, count, (const StatState *)s0,
                 (const double *)partials, (const unsigned *)ghist, std_thres, thres_out, s1, idx, keep);
  }
  return check_launch("outlier_keep_fused");
}
}  // namespace pgdvs

PGDVS_API int pgdvs_outlier_flags(const float *avg, const int32_t *count, int64_t capacity,
                                  float std_thres, int remove_outlier, float *thres_out,
                                  uint8_t *flag_out, void *workspace, int64_t workspace_bytes,
                                  pgdvs_stream_t stream) {
  PGDVS_REQUIRE(avg && count && thres_out && flag_out && capacity >= 0,
                "pgdvs_outlier_flags: bad arguments");
  if (!workspace || workspace_bytes < pgdvs_outlier_workspace_bytes(capacity)) {
    set_error("pgdvs_outlier_flags: workspace too small");
    return PGDVS_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  char *p = reinterpret_cast<char *>(workspace);
  StatState *state = reinterpret_cast<StatState *>(p);
  double *partials = reinterpret_cast<double *>(p + 256);
  unsigned *ghist = reinterpret_cast<unsigned *>(p + 256 + 2 * kStatBlocks * 8);
  hipError_t e = hipMemsetAsync(ghist, 0, 3 * kStatBins * 4, st);
  if (e != hipSuccess) {
    set_error("outlier_flags memset: %s", hipGetErrorString(e));
    return PGDVS_ERR_LAUNCH;
  }
  for (int pass = 0; pass < 3; ++pass) {
    PGDVS_LAUNCH("stat_pass", stat_pass_kernel<false>, dim3(kStatBlocks), dim3(kStatThreads), 0, st, avg,
                 count, pass, (const StatState *)state, partials, ghist, (StatState *)nullptr);
    PGDVS_LAUNCH("stat_select", stat_select_kernel, dim3(1), dim3(256), 0, st, count, pass,
                 kStatBlocks, partials, ghist, state, std_thres, thres_out);
  }
  if (capacity > 0) {
    unsigned grid = (unsigned)(cdiv(capacity, 256) < 1024 ? cdiv(capacity, 256) : 1024);
    PGDVS_LAUNCH("outlier_flag", outlier_flag_kernel, dim3(grid), dim3(256), 0, st, avg, count,
                 capacity, thres_out, remove_outlier, flag_out);
  }
  return check_launch("outlier_flags");
}
