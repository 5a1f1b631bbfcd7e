// Shared helpers for the gfx950 kernels of libpgdvs_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pgdvs_hip.h"

#define PGDVS_API extern "C" __attribute__((visibility("default")))

namespace pgdvs {

void set_error(const char *fmt, ...);

// Process-wide options (include/pgdvs_hip.h, "Options"): initialised from the environment ONCE, when the library is loaded,
// and changed afterwards only through pgdvs_option_set.  Plain words behind relaxed atomic accesses: an entry point reads
// the options it needs once, at its start, so a concurrent pgdvs_option_set takes effect on calls that start after it.
struct Options {
  int agg_ordered;             // PGDVS_AGG_ORDERED=1: A12 as round 2's ordered chain (second implementation for the tests)
  int agg_stage;               // PGDVS_AGG_STAGE=0: A12's links leave no (depth, colour) rows for agg_rows (the long-video path)
  int gnt_fp32;                // PGDVS_GNT_FP32=1: every GNT product on the fp32 matrix instruction (default: bf16x3 products)
  int knn_no_tpq;              // PGDVS_KNN_NO_TPQ=1: diagnostics, the wavefront-per-query search for every query
  int knn_stats;               // PGDVS_KNN_STATS=1: diagnostics, ring histogram to stderr (synchronises)
  float raster_bound_density;  // PGDVS_RASTER_BOUND_DENSITY: rows per pixel from which the rasteriser computes its depth bound
  int side_thread;             // PGDVS_SIDE_THREAD=0: the per-view call enqueues the dynamic branch itself instead of handing it to its worker thread
};
const char *last_error_text();  // this thread's error string (error.cpp)
int option_int(const int &field);
float option_float(const float &field);
const Options &options();

inline hipStream_t as_stream(pgdvs_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return PGDVS_ERR_LAUNCH;
  }
  return PGDVS_OK;
}

// Optional per-kernel timing with HIP events on the launch stream (off by default;
// pgdvs_prof_enable).  A ProfScope brackets one or more launches under a name.
bool prof_enabled();
void prof_begin(const char *name, hipStream_t s);
void prof_end(hipStream_t s);
struct ProfScope {
  hipStream_t s;
  bool on;
  ProfScope(const char *name, hipStream_t st) : s(st), on(prof_enabled()) {
    if (on) prof_begin(name, s);
  }
  ~ProfScope() {
    if (on) prof_end(s);
  }
};

#define PGDVS_LAUNCH(name, kernel, grid, block, lds, stream, ...)       \
  do {                                                                   \
    pgdvs::ProfScope _ps(name, stream);                                  \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);   \
  } while (0)

#define PGDVS_REQUIRE(cond, ...)      \
  do {                                \
    if (!(cond)) {                    \
      pgdvs::set_error(__VA_ARGS__);  \
      return PGDVS_ERR_INVALID;       \
    }                                 \
  } while (0)

constexpr int kWave = 64;

// Byte fill on the stream.  hipMemsetAsync of a few tens of MB runs at ~0.3 TB/s here (the
// 43 MB of splat accumulators took 150 us); 16-byte stores from a grid-stride kernel reach the
// HBM write rate.  Falls back to hipMemsetAsync for small or unaligned regions.
__global__ static void __launch_bounds__(256) fill_bytes_kernel(uint4 *__restrict__ p, int64_t n16, uint32_t v32) {
  const uint4 v = make_uint4(v32, v32, v32, v32);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

inline hipError_t fill_async(void *ptr, unsigned char byte, size_t nbytes, hipStream_t st) {
  if (nbytes < (1u << 20) || (reinterpret_cast<uintptr_t>(ptr) & 15) != 0) return hipMemsetAsync(ptr, byte, nbytes, st);
  const int64_t n16 = (int64_t)(nbytes / 16);
  const uint32_t v32 = 0x01010101u * byte;
  const unsigned grid = (unsigned)((n16 + 255) / 256 < 4096 ? (n16 + 255) / 256 : 4096);
  {
    ProfScope ps("fill_bytes", st);
    hipLaunchKernelGGL(fill_bytes_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<uint4 *>(ptr), n16, v32);
  }
  const size_t tail = nbytes - (size_t)n16 * 16;
  if (tail) return hipMemsetAsync(static_cast<char *>(ptr) + (size_t)n16 * 16, byte, tail, st);
  return hipGetLastError();
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t align_up(int64_t a, int64_t b) { return cdiv(a, b) * b; }

// Camera block kept in registers/SGPRs: loaded through a uniform pointer so the
// compiler can use scalar loads.
struct CamBlock {
  float v[PGDVS_CAM_BLOCK];
};

// Gauss-Jordan inverse with partial pivoting, fp64, n <= 4.  Same operation order
// as the CPU oracle so that derived camera constants agree bit-for-bit.
__host__ __device__ inline int inv_f64(const double *a, double *out, int n) {
  double m[4][8];
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) {
      m[i][j] = a[i * n + j];
      m[i][n + j] = (i == j) ? 1.0 : 0.0;
    }
  }
  for (int c = 0; c < n; ++c) {
    int piv = c;
    double best = fabs(m[c][c]);
    for (int r = c + 1; r < n; ++r) {
      double v = fabs(m[r][c]);
      if (v > best) {
        best = v;
        piv = r;
      }
    }
    if (best == 0.0) return -1;
    if (piv != c) {
      for (int j = 0; j < 2 * n; ++j) {
        double t = m[c][j];
        m[c][j] = m[piv][j];
        m[piv][j] = t;
      }
    }
    double d = m[c][c];
    for (int j = 0; j < 2 * n; ++j) m[c][j] = m[c][j] / d;
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      double f = m[r][c];
      if (f == 0.0) continue;
      for (int j = 0; j < 2 * n; ++j) m[r][j] = m[r][j] - f * m[c][j];
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) out[i * n + j] = m[i][n + j];
  return 0;
}

// flat_cam[34] -> camera block.  fp64 inverses rounded once to fp32, products in
// fp32 with a fixed left-to-right order (no FMA contraction in this library).
__host__ __device__ inline int cam_block_from_flat(const float *flat_cam, float *blk) {
  const float *K = flat_cam + 2;
  const float *c2w = flat_cam + 18;
  double k3[9], k3i[9], c4[16], c4i[16];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) k3[i * 3 + j] = (double)K[i * 4 + j];
  for (int i = 0; i < 16; ++i) c4[i] = (double)c2w[i];
  int bad = 0;
  if (inv_f64(k3, k3i, 3) != 0) bad = 1;
  if (inv_f64(c4, c4i, 4) != 0) bad = 1;
  float kinv[9], w2c[16];
  for (int i = 0; i < 9; ++i) kinv[i] = bad ? __builtin_nanf("") : (float)k3i[i];
  for (int i = 0; i < 16; ++i) w2c[i] = bad ? __builtin_nanf("") : (float)c4i[i];
  for (int i = 0; i < 9; ++i) blk[PGDVS_CAM_KINV + i] = kinv[i];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      float s = c2w[i * 4 + 0] * kinv[0 * 3 + j];
      s = s + c2w[i * 4 + 1] * kinv[1 * 3 + j];
      s = s + c2w[i * 4 + 2] * kinv[2 * 3 + j];
      blk[PGDVS_CAM_M + i * 3 + j] = s;
      blk[PGDVS_CAM_R + i * 3 + j] = c2w[i * 4 + j];
    }
    blk[PGDVS_CAM_O + i] = c2w[i * 4 + 3];
  }
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 4; ++j) {
      float s = K[i * 4 + 0] * w2c[0 * 4 + j];
      s = s + K[i * 4 + 1] * w2c[1 * 4 + j];
      s = s + K[i * 4 + 2] * w2c[2 * 4 + j];
      s = s + K[i * 4 + 3] * w2c[3 * 4 + j];
      blk[PGDVS_CAM_P + i * 4 + j] = s;
    }
  }
  for (int i = 0; i < 16; ++i) blk[PGDVS_CAM_W2C + i] = w2c[i];
  blk[PGDVS_CAM_HW + 0] = flat_cam[0];
  blk[PGDVS_CAM_HW + 1] = flat_cam[1];
  for (int i = 0; i < 16; ++i) blk[PGDVS_CAM_K + i] = K[i];
  return bad ? -1 : 0;
}

// order-preserving map float -> unsigned (and back): bounding boxes through integer atomicMax (knn_grid.hip, scan.hip)
__device__ __forceinline__ unsigned f2ord(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) {
  return x < lo ? lo : (x > hi ? hi : x);
}

// 3-vector  M(3x3,row-major) * (a,b,c) with ((m0*a + m1*b) + m2*c) ordering
__device__ __forceinline__ float dot3(const float *m, float a, float b, float c) {
  float s = m[0] * a;
  s = s + m[1] * b;
  s = s + m[2] * c;
  return s;
}

// p = P(4x4) @ [X,1]; uv = p[:2] / clamp(p[2], min=1e-8); clamp +-1e6
// (pgdvs/models/gnt/projector.py:58-67)
__device__ __forceinline__ void project_point(const float *P, float x, float y, float z,
                                              float &u, float &v) {
  float p[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float s = P[i * 4 + 0] * x;
    s = s + P[i * 4 + 1] * y;
    s = s + P[i * 4 + 2] * z;
    s = s + P[i * 4 + 3];
    p[i] = s;
  }
  float zz = p[2] < 1e-8f ? 1e-8f : p[2];
  u = clampf(p[0] / zz, -1e6f, 1e6f);
  v = clampf(p[1] / zz, -1e6f, 1e6f);
}

// Wave-aggregated tile counter update.  Clouds arrive in the raster order of their source
// frames, so consecutive lanes of a wavefront mostly hit the same tile: lanes form RUNS of
// equal tile id, the first lane of each run adds the run length for all of them, and every
// run leader is active in ONE atomic wave-instruction (the atomic units are paced per
// instruction, not per lane).  Any order is correct -- less coherent input only means
// shorter runs.  Returns the slot reserved for this lane (fill) or nothing (count).
struct RunInfo {
  int leader;   // lane index of this lane's run leader
  int length;   // run length (valid on the leader)
  bool is_leader;
};

// PRECONDITION: all 64 lanes of the wavefront call it together (callers pad their loops to whole wavefronts and pass
// t < 0 for lanes without work).  An inactive left neighbour leaves the DPP read at the lane's own value; the ballot of
// active lanes below makes such a lane a leader anyway, so a partial wavefront still gets correct (shorter) runs.
__device__ __forceinline__ RunInfo wave_runs(int t) {
  const int lane = threadIdx.x & 63;
  // wave_shr:1 -- lane i reads lane i-1 through the DPP path (no LDS round trip); lane 0 is a leader anyway
  int prev = __builtin_amdgcn_update_dpp(t, t, 0x138, 0xf, 0xf, false);
  const unsigned long long act = __ballot(1);
  bool lead = lane == 0 || prev != t || !((act >> (lane ? lane - 1 : 0)) & 1ull);
  unsigned long long L = __ballot(lead);
  RunInfo r;
  r.is_leader = lead;
  unsigned long long below = L & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));  // leaders at or below this lane
  r.leader = 63 - __builtin_clzll(below);
  unsigned long long above = lane == 63 ? 0ull : (L >> (lane + 1));  // leaders after this lane
  r.length = above ? (int)__builtin_ctzll(above) + 1 : 64 - lane;
  return r;
}

__device__ __forceinline__ void wave_tile_count(int32_t *__restrict__ counter, int t) {
  RunInfo r = wave_runs(t);
  if (r.is_leader && t >= 0)
    __hip_atomic_fetch_add(&counter[t], r.length, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int wave_tile_reserve(int32_t *__restrict__ counter, int t) {
  const int lane = threadIdx.x & 63;
  RunInfo r = wave_runs(t);
  int base = 0;
  if (r.is_leader && t >= 0) base = atomicAdd(&counter[t], r.length);
  base = __shfl(base, r.leader, 64);
  return base + (lane - r.leader);
}


// wave-level inclusive/exclusive helpers (64-wide wavefront)
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

}  // namespace pgdvs
